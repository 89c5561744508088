"""What the a5 selection refinement (VideoEncoder.selection_refine) costs: one T = 512 step of the bench type on (a) the bench's own
synthetic video, whose decisive similarity ranks are 0.43 apart (nothing is re-encoded), and (b) the slow-drift video of
tests/test_hip_selection_risk.py, whose ranks are ~3e-4 apart everywhere (the worst case: a band of pairs is re-encoded by the
fp16-operand DINOv2 tower in every call) - each with the refinement on and off, interleaved.
    python tools/bench_selection_refine.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import test_hip_selection_risk as tsr  # noqa: E402


def main():
    enc, _, drift = tsr.build_world()
    dev = drift.device
    T = drift.shape[0]
    vs = bench.synth_video(0, T, 384, dev, torch.bfloat16)
    vd = bench.synth_video(0, T, 378, dev, torch.bfloat16, seed=4321)
    prompt = [101] + list(range(2000, 2010)) + [102]

    def run(px_d, eps, n=4):
        enc.selection_eps = eps
        info = {}
        for _ in range(2):
            enc.encode_video(vs, px_d, (384, 384), budget_text_len=64, n_text_tokens=64, prompt_ids=prompt, frame_cap=T, info=info)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            enc.encode_video(vs, px_d, (384, 384), budget_text_len=64, n_text_tokens=64, prompt_ids=prompt, frame_cap=T)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, len(info.get("refined_pairs", []))
    for name, px_d in (("bench video (margin 0.43)", vd), ("slow-drift video (near-tied everywhere)", drift)):
        rows = []
        for rep in range(2):
            for eps in (None, 1e-3):
                ms, pairs = run(px_d, eps)
                rows.append((eps, ms, pairs))
        off = sum(r[1] for r in rows if r[0] is None) / 2
        on = sum(r[1] for r in rows if r[0] is not None) / 2
        print("%-42s refinement off %8.2f ms | on %8.2f ms (%+.2f %%), %d pairs re-ranked per call   [%s]"
              % (name, off, on, (on / off - 1) * 100, rows[-1][2], ", ".join("%.1f" % r[1] for r in rows)), flush=True)
    enc.selection_eps = 1e-3


if __name__ == "__main__":
    main()
