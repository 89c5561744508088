"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of tools/gemm_pmc into per-shape and per-launch HBM traffic.
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads ->
doubled; both counters are in KiB."""
import csv
import glob
import json
import sys

out, shapes_file = sys.argv[1], sys.argv[2]
shapes = [tuple(int(x) for x in l.split()) for l in open(shapes_file) if l.strip()]


def per_dispatch(counter):
    f = glob.glob("%s/*pmc_%s*counter_collection.csv" % (out, counter))
    rows = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == counter and "gemm" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows]


REPS = int(__import__("os").environ.get("PMC_REPS", "2"))   # launches per shape in the replay (tools/run_gemm_pmc.sh)


def fold(values):
    """per-dispatch values -> one value per shape: the LAST of the shape's REPS launches (the first one pays for the code
    object, cold TLBs and freshly allocated pages: 10-15 % longer)"""
    assert len(values) == REPS * len(shapes), (len(values), REPS, len(shapes))
    return [values[REPS * i + REPS - 1] for i in range(len(shapes))]


import os


def kernel_ms():
    """per-shape kernel duration (ms) from the kernel trace of the GRBM_GUI_ACTIVE pass (any pass would do: one counter
    per pass costs the kernel nothing measurable): lets bench.py print live / replayed GEMM time, so staleness shows"""
    f = glob.glob("%s/*pmc_GRBM_GUI_ACTIVE*kernel_trace.csv" % out)
    rows = [r for r in csv.DictReader(open(f[0])) if "gemm" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return fold([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in rows])


fetch, write = fold(per_dispatch("FETCH_SIZE")), fold(per_dispatch("WRITE_SIZE"))
assert len(fetch) == len(shapes) == len(write), (len(fetch), len(write), len(shapes))
try:    # MFMA pipe occupancy: busy cycles summed over the 1024 SIMDs / (1024 x kernel cycles); GRBM_GUI_ACTIVE is summed
    # over the 8 XCDs (MI355X_MICROARCH.md, DVFS).  16 busy cycles per v_mfma_f32_16x16x32 (its pipe time).
    mfma, gui = fold(per_dispatch("SQ_VALU_MFMA_BUSY_CYCLES")), fold(per_dispatch("GRBM_GUI_ACTIVE"))
    assert len(mfma) == len(gui) == len(shapes)
except Exception as e:  # noqa: BLE001
    print("no MFMA-busy pass:", e)
    mfma = gui = None
tot_traffic = tot_alg = tot_n = 0.0
res = []
for (M, N, K, act, has_res, of32, count), fk, wk in zip(shapes, fetch, write):
    traffic = (2.0 * fk + wk) * 1024.0
    n_out = N // 2 if act == 3 else N
    # has_res: 0 none, 1 fp32 residual (with an fp32 output), 2 16-bit residual (with a 16-bit output): read + write of C's type
    alg = 2.0 * (M * K + N * K) + M * n_out * (4 if of32 else 2) * (2 if has_res else 1)
    row = dict(M=M, N=N, K=K, act=act, res=has_res, out_f32=of32, launches_per_step=count,
               hbm_bytes=traffic, algorithmic_bytes=alg, ratio=round(traffic / alg, 3))
    if mfma is not None:
        i = len(res)
        cyc = gui[i] / 8.0
        row.update(mfma_busy_cycles=mfma[i], kernel_cycles=cyc, mfma_busy_frac=round(mfma[i] / (1024.0 * cyc), 4),
                   mfma_busy_expected=16.0 * (2.0 * M * N * K / 16384.0))
        mf_b = locals().get("mf_b", 0.0) + mfma[i] * count
        mf_c = locals().get("mf_c", 0.0) + 1024.0 * cyc * count
    res.append(row)
    tot_traffic += traffic * count
    tot_alg += alg * count
    tot_n += count
summary = dict(per_launch_hbm_bytes=tot_traffic / tot_n, per_launch_algorithmic_bytes=tot_alg / tot_n,
               ratio=round(tot_traffic / tot_alg, 3), launches_per_step=int(tot_n), shapes=res)
if mfma is not None:
    summary["mfma_busy_frac"] = round(mf_b / mf_c, 4)
# provenance: the bench command line these launches belong to (written by bench.py --dump-gemm-shapes) and the commit of the
# library the counters were collected on (TDC_COMMIT: the GPU box has no .git)
import os
if os.path.exists(shapes_file + ".args.json"):
    summary["bench_args"] = json.load(open(shapes_file + ".args.json"))
try:
    ms = kernel_ms()
    for r, m in zip(res, ms):
        r["kernel_ms"] = round(m, 4)
    summary["gemm_ms_per_step_at_collection"] = round(sum(m * r["launches_per_step"] for r, m in zip(res, ms)), 2)
except Exception as e:  # noqa: BLE001
    print("no kernel trace:", e)
summary["collected_at_commit"] = os.environ.get("TDC_COMMIT", "unknown")
summary["collected_by"] = "tools/run_gemm_pmc.sh (torch-free replay tools/gemm_pmc.cpp, one counter per rocprofv3 pass)"
json.dump(summary, open(out + "/gemm_pmc_summary.json", "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "shapes"}))
for r in res[:12]:
    print(r)
