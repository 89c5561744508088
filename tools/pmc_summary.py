"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE passes of tools/gemm_pmc into
per-shape and per-launch HBM-side traffic and MFMA-pipe occupancy.
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads ->
doubled; both counters are in KiB.  One counter per pass (FETCH_SIZE and WRITE_SIZE do not fit one pass).

  python tools/pmc_summary.py <dir with the pmc_<COUNTER>_*.csv files> <shapes file>      (tools/run_gemm_pmc.sh)
bench.py imports collect() / summarise() for its in-run counter leg."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

COUNTERS = ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")


def read_shapes(path):
    return [tuple(int(x) for x in l.split()) for l in open(path) if l.strip()]


def _per_dispatch(out, counter):
    f = glob.glob("%s/*pmc_%s*counter_collection.csv" % (out, counter))
    rows = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == counter and "gemm" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows]


def _fold(values, n_shapes, reps):
    """per-dispatch values -> one value per shape: the LAST of the shape's `reps` launches (the first one pays for the code
    object, cold TLBs and freshly allocated pages: 10-15 % longer)"""
    assert len(values) == reps * n_shapes, (len(values), reps, n_shapes)
    return [values[reps * i + reps - 1] for i in range(n_shapes)]


def _kernel_ms(out, n_shapes, reps):
    """per-shape kernel duration (ms) from the kernel trace of the GRBM_GUI_ACTIVE pass (any pass would do: one counter
    per pass costs the kernel nothing measurable): lets bench.py print live / replayed GEMM time, so a slow replay shows"""
    f = glob.glob("%s/*pmc_GRBM_GUI_ACTIVE*kernel_trace.csv" % out)
    rows = [r for r in csv.DictReader(open(f[0])) if "gemm" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return _fold([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in rows], n_shapes, reps)


def summarise(out, shapes, reps=2):
    """shapes: (M, N, K, act, res, out_f32, launches per step) in the replay's order -> summary dict"""
    n = len(shapes)
    fetch, write = _fold(_per_dispatch(out, "FETCH_SIZE"), n, reps), _fold(_per_dispatch(out, "WRITE_SIZE"), n, reps)
    try:    # MFMA pipe occupancy: busy cycles summed over the 1024 SIMDs / (1024 x kernel cycles); GRBM_GUI_ACTIVE is summed
        # over the 8 XCDs (MI355X_MICROARCH.md, DVFS).  16 busy cycles per v_mfma_f32_16x16x32 (its pipe time).
        mfma = _fold(_per_dispatch(out, "SQ_VALU_MFMA_BUSY_CYCLES"), n, reps)
        gui = _fold(_per_dispatch(out, "GRBM_GUI_ACTIVE"), n, reps)
    except Exception as e:  # noqa: BLE001
        print("no MFMA-busy pass:", e, file=sys.stderr)
        mfma = gui = None
    tot_traffic = tot_alg = tot_n = mf_b = mf_c = 0.0
    res = []
    for i, ((M, N, K, act, has_res, of32, count), fk, wk) in enumerate(zip(shapes, fetch, write)):
        traffic = (2.0 * fk + wk) * 1024.0
        n_out = N // 2 if act == 3 else N
        # has_res: 0 none, 1 fp32 residual (with an fp32 output), 2 16-bit residual (with a 16-bit output): read + write of C's type
        alg = 2.0 * (M * K + N * K) + M * n_out * (4 if of32 else 2) * (2 if has_res else 1)
        row = dict(M=M, N=N, K=K, act=act, res=has_res, out_f32=of32, launches_per_step=count,
                   hbm_bytes=traffic, algorithmic_bytes=alg, ratio=round(traffic / alg, 3))
        if mfma is not None:
            cyc = gui[i] / 8.0
            row.update(mfma_busy_cycles=mfma[i], kernel_cycles=cyc, mfma_busy_frac=round(mfma[i] / (1024.0 * cyc), 4),
                       mfma_busy_expected=16.0 * (2.0 * M * N * K / 16384.0))
            mf_b += mfma[i] * count
            mf_c += 1024.0 * cyc * count
        res.append(row)
        tot_traffic += traffic * count
        tot_alg += alg * count
        tot_n += count
    summary = dict(per_launch_hbm_bytes=tot_traffic / tot_n, per_launch_algorithmic_bytes=tot_alg / tot_n,
                   ratio=round(tot_traffic / tot_alg, 3), launches_per_step=int(tot_n), shapes=res)
    if mfma is not None:
        summary["mfma_busy_frac"] = round(mf_b / mf_c, 4)
    try:
        ms = _kernel_ms(out, n, reps)
        for r, m in zip(res, ms):
            r["kernel_ms"] = round(m, 4)
        summary["gemm_ms_per_step_at_collection"] = round(sum(m * r["launches_per_step"] for r, m in zip(res, ms)), 2)
    except Exception as e:  # noqa: BLE001
        print("no kernel trace:", e, file=sys.stderr)
    return summary


def collect(binary, shapes_file, out, reps=2, timeout_s=60.0, log=None):
    """Run the torch-free replay `binary` once per counter under rocprofv3 (the program itself behind `--`; each pass is a FRESH
    child process - the profiler's counter collection cannot run inside a torch process on this image).  Returns the list of
    counters whose pass completed; stops early when the time budget is spent."""
    import time
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof) or not os.path.exists(binary):
        return []
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    done = []
    t_end = time.monotonic() + timeout_s
    for c in COUNTERS:
        left = t_end - time.monotonic()
        if left < 5:
            break
        cmd = [rocprof, "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "pmc_" + c, "--",
               binary, shapes_file, str(reps)]
        try:
            r = subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=left, text=True)
        except subprocess.TimeoutExpired:
            break
        if log is not None:
            log.append((c, r.returncode, r.stdout[-400:]))
        if r.returncode != 0:
            break
        done.append(c)
    return done


if __name__ == "__main__":
    out_dir, shapes_file = sys.argv[1], sys.argv[2]
    summary = summarise(out_dir, read_shapes(shapes_file), int(os.environ.get("PMC_REPS", "2")))
    # provenance: the bench command line these launches belong to (written by bench.py --dump-gemm-shapes) and the commit of the
    # library the counters were collected on (TDC_COMMIT: the GPU box has no .git)
    if os.path.exists(shapes_file + ".args.json"):
        summary["bench_args"] = json.load(open(shapes_file + ".args.json"))
    summary["collected_at_commit"] = os.environ.get("TDC_COMMIT", "unknown")
    summary["collected_by"] = "tools/run_gemm_pmc.sh (torch-free replay tools/gemm_pmc.cpp, one counter per rocprofv3 pass)"
    json.dump(summary, open(out_dir + "/gemm_pmc_summary.json", "w"), indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k != "shapes"}))
    for r in summary["shapes"][:12]:
        print(r)
