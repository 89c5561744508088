"""Build a stand-alone attention prototype (a .hip file exporting tdc_attention_proto(const tdc_attn_desc*, stream)) with optional
defines, time it on the DINOv2 tower shape (24 heads x 64, S = 730) beside the library kernel and check it against torch SDPA.
GPU box:  python tools/attn_proto_bench.py tools/attention_swp_proto.hip [batch=512] [DEF=VAL[,DEF=VAL..]] ..."""
import ctypes as C
import math
import os
import subprocess
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tdc_video_amd  # noqa: F401,E402
from tdc_video_amd import lib as L, ops  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

src = os.path.join(ROOT, sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
variants = sys.argv[3:] or [""]
H, d, S = 24, 64, 730
D = H * d
ld = ops.pad64(3 * D)
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * S, ld, device="cuda", generator=g).to(torch.bfloat16)
out = torch.zeros(B * S, ops.pad64(D), device="cuda", dtype=torch.bfloat16)
q_, k_, v_ = (qkv[: 2 * S, i * D:(i + 1) * D].float().view(2, S, H, d).transpose(1, 2) for i in range(3))
ref = F.scaled_dot_product_attention(q_, k_, v_).transpose(1, 2).reshape(2 * S, D)
a = L.AttnDesc()
a.q, a.k, a.v, a.o = qkv.data_ptr(), qkv.data_ptr() + 2 * D, qkv.data_ptr() + 4 * D, out.data_ptr()
a.q_bs = a.k_bs = a.v_bs = S * ld
a.o_bs = S * out.stride(0)
a.q_rs = a.k_rs = a.v_rs = ld
a.o_rs = out.stride(0)
a.batch, a.heads, a.head_dim, a.sq, a.sk = B, H, d, S, S
a.scale, a.dtype = 1 / math.sqrt(d), L.BF16
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
flop = 4.0 * B * H * S * S * d
libfn = L.load().tdc_attention
ms = timeit(lambda: libfn(C.byref(a), st))
print("library kernel            %8.3f ms %7.1f TFLOP/s | max err %.2e" % (ms, flop / ms / 1e9, (out[: 2 * S, :D].float() - ref).abs().max().item()),
      flush=True)
scratch = os.path.join(ROOT, "gpurun_out")
os.makedirs(scratch, exist_ok=True)
for vi, v in enumerate(variants):
    so = os.path.join(scratch, "attn_proto_%d.so" % vi)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w",
                           "-mllvm", "--amdgpu-mfma-vgpr-form", "-fno-honor-nans", "-fno-slp-vectorize",
                           *["-D" + x for x in v.split(",") if x], src, "-o", so])
    fn = C.CDLL(so).tdc_attention_proto
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(L.AttnDesc), C.c_void_p]
    out.zero_()
    assert fn(C.byref(a), st) == 0
    torch.cuda.synchronize()
    err = (out[: 2 * S, :D].float() - ref).abs().max().item()
    ms = timeit(lambda: fn(C.byref(a), st))
    print("prototype %-15s %8.3f ms %7.1f TFLOP/s | max err %.2e" % (v or "(defaults)", ms, flop / ms / 1e9, err), flush=True)
    os.remove(so)
