"""Same-box A/B of builds of the tower attention kernel (csrc/attention32.hip): the committed source (git show HEAD:...) against the
working tree with different flags / defines.  Every variant is built with csrc/attention.hip into a scratch library and timed on
the two tower shapes through tdc_attention.
GPU box:  python tools/attn32_variants.py [batch=512] > gpurun_out/attn32_variants.log"""
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
scratch = os.path.join(ROOT, "gpurun_out")
os.makedirs(scratch, exist_ok=True)
csrc = os.path.join(ROOT, "tdc-video_amd", "csrc")
base_src = os.path.join(csrc, "attention32_head.hip")          # beside the others: relative includes
ref = os.path.join(ROOT, "tools", "attention32_head.txt")      # committed source, saved by the caller before gpurun (no .git on
                                                                # the GPU box): git show HEAD:tdc-video_amd/csrc/attention32.hip > tools/attention32_head.txt
if os.path.exists(ref):
    open(base_src, "w").write(open(ref).read())
# (name, source, extra hipcc arguments): edit to the experiment at hand
tree = os.path.join(csrc, "attention32.hip")
variants = [("committed source", base_src, []),
            ("working tree", tree, []),
            ("working tree, packed softmax arithmetic (rounds 2-4)", tree, ["-DATTN32_PACKED"]),
            ("working tree (again)", tree, [])]
g = torch.Generator(device="cuda").manual_seed(0)
shapes = [(16, 72, 729), (24, 64, 730)]
data = {}
for (H, d, S) in shapes:
    D = H * d
    ld = ops.pad64(3 * D)
    data[d] = (torch.randn(B * S, ld, device="cuda", generator=g).to(torch.bfloat16),
               torch.empty(B * S, ops.pad64(D), device="cuda", dtype=torch.bfloat16))
for vi, (name, src, extra) in enumerate(variants):
    if not os.path.exists(src):
        continue
    so = os.path.join(scratch, "attn32_variant_%d.so" % vi)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w",
                           "-mllvm", "--amdgpu-mfma-vgpr-form", "-fno-honor-nans", *extra, src, os.path.join(csrc, "attention.hip"), "-x", "hip", os.path.join(csrc, "profile.cpp"),
                           "-o", so])
    lib = C.CDLL(so)
    lib.tdc_attention.restype = C.c_int
    lib.tdc_attention.argtypes = [C.POINTER(L.AttnDesc), C.c_void_p]
    line = "%-44s" % name
    for (H, d, S) in shapes:
        qkv, out = data[d]
        D = H * d
        ld = qkv.stride(0)
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
            assert lib.tdc_attention(C.byref(a), st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.tdc_attention(C.byref(a), st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line += " | d = %d %.3f ms %6.1f TFLOP/s" % (d, ms, 4.0 * B * H * S * S * d / ms / 1e9)
    print(line, flush=True)
    os.remove(so)
if os.path.exists(ref):
    os.remove(base_src)
