"""Same-process A/B of a tdc_gemm_set_debug()-switched kernel experiment on tower GEMM shapes (alternating runs, 3 rounds):
python tools/bench_gemm_ab.py <debug value> rmw|plain"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tdc_video_amd  # noqa
from tdc_video_amd import ops, lib as L
from tools.bench_ops import timeit
frames, flag, kind = 256, sys.argv[1], sys.argv[2]
g = torch.Generator(device="cuda").manual_seed(0)
shapes = {"rmw": (("siglip out", 729, 1152, 1152), ("siglip fc2", 729, 1152, 4352), ("dino out", 730, 1536, 1536), ("dino fc2", 730, 1536, 4096)),
          "plain": (("siglip qkv", 729, 3456, 1152), ("dino qkv", 730, 4608, 1536))}[kind]
for name, S, N, K in shapes:
    M = frames * S
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g)
    if kind == "rmw":
        x = torch.randn(M, N, device="cuda", generator=g)
        fn = lambda: ops.gemm(a, w, bias=b, res=x, out=x, out_f32=True)
    else:
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        fn = lambda: ops.gemm(a, w, bias=b, out=out)
    res = []
    for rep in range(3):
        for dbg in (None, flag):
            L.load().tdc_gemm_set_debug(int(dbg) if dbg else 0)
            res.append(timeit(fn, iters=10))
    L.load().tdc_gemm_set_debug(0)
    print(name, " base: %s  switched: %s" % (["%.3f" % t for t in res[0::2]], ["%.3f" % t for t in res[1::2]]), flush=True)
