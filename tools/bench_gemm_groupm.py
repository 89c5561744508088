"""Time of the wide GEMMs of a step against the group height of the tile order (TDC_GEMM_GROUP_M, diagnostics build of csrc/gemm.hip
only): the Q-Former K/V projection [F N, 9216, 3584] and the towers' qkv / fc1 shapes.  One process per group height (the library
caches the setting).  GPU box:  python tools/bench_gemm_groupm.py > gpurun_out/gemm_groupm.log"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "gpurun_out", "gemm_diag.so")
SHAPES = [(68484, 9216, 3584, "fp16"), (373760, 8192, 1536, "bf16"), (373760, 4608, 1536, "bf16"), (373248, 4352, 1152, "bf16"),
          (373248, 3456, 1152, "bf16")]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ROOT)
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd import lib as L
    lib = C.CDLL(SO)
    lib.tdc_gemm.restype = C.c_int
    lib.tdc_gemm.argtypes = [C.POINTER(L.GemmDesc), C.c_void_p]
    out = []
    for M, N, K, dt in SHAPES:
        t = torch.float16 if dt == "fp16" else torch.bfloat16
        a = torch.randn(M, K, device="cuda").to(t)
        w = (torch.randn(N, K, device="cuda") * 0.02).to(t)
        c = torch.empty(M, N, device="cuda", dtype=t)
        d = L.GemmDesc()
        d.A, d.lda, d.W, d.ldw, d.C, d.ldc = a.data_ptr(), K, w.data_ptr(), K, c.data_ptr(), N
        d.M, d.N, d.K, d.dtype = M, N, K, (L.F16 if dt == "fp16" else L.BF16)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(3):
            assert lib.tdc_gemm(C.byref(d), st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.tdc_gemm(C.byref(d), st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        out.append("%d x %d x %d %.3f ms %.0f TF/s" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
        del a, w, c
    print("TDC_GEMM_GROUP_M=%s | " % os.environ.get("TDC_GEMM_GROUP_M", "default") + " | ".join(out), flush=True)
    sys.exit(0)

os.makedirs(os.path.dirname(SO), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w", "-DTDC_GEMM_DIAG",
                       os.path.join(ROOT, "tdc-video_amd", "csrc", "gemm.hip"), "-o", SO])
for gm in [None, 2, 4, 8, 16]:
    env = dict(os.environ)
    if gm is not None:
        env["TDC_GEMM_GROUP_M"] = str(gm)
    subprocess.call([sys.executable, os.path.abspath(__file__), "child"], env=env)
os.remove(SO)
