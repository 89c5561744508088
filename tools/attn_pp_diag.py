"""Where does the phase-alternating attention prototype (tools/attention_pp_proto.hip) spend its time?  Builds it
with -DPP_DIAG=<mask> (pieces left out: 1 = softmax arithmetic, 2 = MFMAs, 4 = LDS fragment reads, 8 = K / V staging) into scratch
libraries and times the DINOv2 shape (24 heads x 64, S = 730) through each.
GPU box:  python tools/attn_pp_diag.py [batch=512] [masks...] > gpurun_out/attn_pp_diag.log"""
import ctypes as C
import math
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tdc_video_amd  # noqa: F401,E402
from tdc_video_amd import lib as L, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
# a mask may carry defines: "0:PP_YPRIO=1:PP_YIELD=4"
masks = sys.argv[2:] or ["0", "1", "2", "3", "4", "8", "6", "14", "15"]
H, d, S = 24, 64, 730
D = H * d
ld = ops.pad64(3 * D)
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * S, ld, device="cuda", generator=g).to(torch.bfloat16)
out = torch.empty(B * S, ops.pad64(D), device="cuda", dtype=torch.bfloat16)
scratch = os.path.join(ROOT, "gpurun_out")
os.makedirs(scratch, exist_ok=True)
csrc = os.path.join(ROOT, "tdc-video_amd", "csrc")
for m in masks:
    m, *defs = m.split(":")
    m = int(m)
    so = os.path.join(scratch, "attn_pp_diag_%d_%s.so" % (m, "_".join(d.replace("=", "") for d in defs)))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
                           "-mllvm", "--amdgpu-mfma-vgpr-form", "-fno-honor-nans", "-fno-slp-vectorize", "-DPP_DIAG=%d" % m, *["-D" + d for d in defs],
                           os.path.join(ROOT, "tools", "attention_pp_proto.hip"), "-o", so])
    lib = C.CDLL(so)
    fn = lib.tdc_attention_pp_proto
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(L.AttnDesc), C.c_void_p]
    a = L.AttnDesc()
    a.q, a.k, a.v, a.o = qkv.data_ptr(), qkv.data_ptr() + 2 * D, qkv.data_ptr() + 4 * D, out.data_ptr()
    a.q_bs = a.k_bs = a.v_bs = S * ld
    a.o_bs = S * out.stride(0)
    a.q_rs = a.k_rs = a.v_rs = ld
    a.o_rs = out.stride(0)
    a.batch, a.heads, a.head_dim, a.sq, a.sk = B, H, d, S, S
    a.scale, a.dtype = 1 / math.sqrt(d), L.BF16
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        assert fn(C.byref(a), st) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn(C.byref(a), st)
    e1.record()
    torch.cuda.synchronize()
    print("PP_DIAG=%2d %s: %.3f ms per launch" % (m, " ".join(defs), e0.elapsed_time(e1) / 10), flush=True)
