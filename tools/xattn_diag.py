"""Where does tdc_qformer_xattn spend its time?  Builds csrc/xattn.hip with -DXATTN_DIAG=<mask> (pieces left out: 1 = q-proj
MFMA loop, 2 = attention phase, 4 = out-proj MFMA loop, 8 = final stores, 128 = no staging loads) into scratch libraries and times the same launch
through each.  GPU box:  python tools/xattn_diag.py [--out-only] [masks...] > gpurun_out/xattn_diag.log"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tdc_video_amd  # noqa: F401,E402
from tdc_video_amd import lib as L, ops  # noqa: E402

OUT = "--out-only" in sys.argv      # time the ctx form (output projection + residual + LayerNorm alone)
# a mask may carry defines: "0:XATTN_OUT_OLD" = the ctx form on the 8-wave / 64-row kernel instead of the 4-wave / 32-row one
masks = [a for a in sys.argv[1:] if a != "--out-only"] or ["0", "1", "2", "4", "7", "15"]
F, K, N, D, heads, Lt = 439, 144, 156, 768, 12, 12
S = K + Lt
dt, dev = torch.float16, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: torch.randn(*s, device=dev, generator=g) * sc
h32 = rnd(F * S, D); h16 = h32.to(dt)
wq, wo = ops.xattn_tile_weight(rnd(D, D, sc=0.03).to(dt)), ops.xattn_tile_weight(rnd(D, D, sc=0.03).to(dt))
bq, bo, bv = rnd(D, sc=0.02), rnd(D, sc=0.02), rnd(D, sc=0.02)
ln_g, ln_b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
k = rnd(F * N, D).to(dt)
vt = rnd(D, ops.pad64(F * N)).to(dt)
ctx = rnd(F * K, D).to(dt)
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
for m in masks:
    m, *defs = m.split(":")
    m = int(m)
    so = os.path.join(out, "xattn_diag_%d_%s.so" % (m, "_".join(defs)))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
                           "-DXATTN_DIAG=%d" % m, *["-D" + d for d in defs], os.path.join(ROOT, "tdc-video_amd", "csrc", "xattn.hip"), "-o", so])
    lib = C.CDLL(so)
    lib.tdc_qformer_xattn.restype = C.c_int
    lib.tdc_qformer_xattn.argtypes = [C.POINTER(L.XattnDesc), C.c_void_p]
    d = L.XattnDesc()
    d.h16, d.h32, d.ldh = h16.data_ptr(), h32.data_ptr(), h16.stride(0)
    d.F, d.K, d.S = F, K, S
    d.wq, d.bq, d.wo, d.bo = wq.data_ptr(), bq.data_ptr(), wo.data_ptr(), bo.data_ptr()
    d.k, d.ldk, d.vt, d.ldvt, d.bv = k.data_ptr(), k.stride(0), vt.data_ptr(), vt.stride(0), bv.data_ptr()
    d.Nenc, d.ln_g, d.ln_b, d.eps = N, ln_g.data_ptr(), ln_b.data_ptr(), 1e-12
    d.dim, d.heads, d.scale, d.dtype = D, heads, 0.125, L.F16
    if OUT:
        d.ctx, d.ldctx = ctx.data_ptr(), ctx.stride(0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        assert lib.tdc_qformer_xattn(C.byref(d), st) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.tdc_qformer_xattn(C.byref(d), st)
    e1.record()
    torch.cuda.synchronize()
    print("XATTN_DIAG=%2d %s: %.3f ms per launch" % (m, " ".join(defs), e0.elapsed_time(e1) / 10), flush=True)
