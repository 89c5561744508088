#!/bin/bash
# Fabric traffic (FETCH_SIZE x 2 + WRITE_SIZE, gfx950 correction) and kernel time of the eight tower GEMMs against the group
# height of the tile order (TDC_GEMM_GROUP_M; a -DTDC_GEMM_DIAG build of csrc/gemm.hip linked into the torch-free replay - the
# product library reads no environment).  "auto" = the library's own choice (choose_group_m).  Three launches per shape, the
# last one counts.  Run on the MI355X box from the repo root:  bash tools/run_gemm_groupm_traffic.sh
set -e
OUT=gpurun_out/groupm
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DTDC_GEMM_DIAG -o $OUT/gemm_pmc_diag tools/gemm_pmc.cpp tdc-video_amd/csrc/gemm.hip
cat > $OUT/shapes.txt <<S
373760 8192 1536 3 0 0 40
373760 4608 1536 0 0 0 40
373760 1536 4096 0 2 0 40
373760 1536 1536 0 2 0 40
373248 4352 1152 2 0 0 27
373248 3456 1152 0 0 0 27
373248 1152 4352 0 2 0 27
373248 1152 1152 0 2 0 27
68484 9216 3584 0 0 0 1
S
export TMPDIR=/tmp
export PMC_REPS=3
for g in auto 1 2 4 8; do
  if [ $g = auto ]; then unset TDC_GEMM_GROUP_M; else export TDC_GEMM_GROUP_M=$g; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/g$g -o pmc_$c -- $OUT/gemm_pmc_diag $OUT/shapes.txt $PMC_REPS > $OUT/g${g}_$c.log 2>&1 || tail -5 $OUT/g${g}_$c.log
  done
done
python3 - <<'PY'
import csv, glob
out = "gpurun_out/groupm"
shapes = [tuple(int(x) for x in l.split()) for l in open(out + "/shapes.txt") if l.strip()]
REPS = 3
def last(vals): return [vals[REPS * i + REPS - 1] for i in range(len(shapes))]
rows = []
for g in ("auto", "1", "2", "4", "8"):
    def col(counter):
        f = glob.glob("%s/g%s/**/*pmc_%s*counter_collection.csv" % (out, g, counter), recursive=True)[0]
        r = [x for x in csv.DictReader(open(f)) if x["Counter_Name"] == counter and "gemm" in x["Kernel_Name"]]
        r.sort(key=lambda x: int(x["Dispatch_Id"]))
        return last([float(x["Counter_Value"]) for x in r])
    def ms():
        f = glob.glob("%s/g%s/**/*pmc_FETCH_SIZE*kernel_trace.csv" % (out, g), recursive=True)[0]
        r = [x for x in csv.DictReader(open(f)) if "gemm" in x["Kernel_Name"]]
        r.sort(key=lambda x: int(x["Dispatch_Id"]))
        return last([(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) * 1e-6 for x in r])
    rows.append((g, col("FETCH_SIZE"), col("WRITE_SIZE"), ms()))
with open(out + "/summary.txt", "w") as fo:
    for i, (M, N, K, act, res, of32, cnt) in enumerate(shapes):
        nout = N // 2 if act == 3 else N
        alg = 2.0 * (M * K + N * K) + M * nout * (4 if of32 else 2) * (2 if res else 1)
        line = "%6d x %4d x %4d res %d |" % (M, N, K, res)
        for g, f, w, t in rows:
            line += "  G=%-4s %.2fx %.3f ms |" % (g, (2.0 * f[i] + w[i]) * 1024.0 / alg, t[i])
        print(line); fo.write(line + "\n")
    tot = "per step (launch counts of the bench):"
    for g, f, w, t in rows:
        tot += "  G=%-4s %.2f TB %.1f ms |" % (g, sum((2.0 * f[i] + w[i]) * 1024.0 * s[6] for i, s in enumerate(shapes)) / 1e12,
                                             sum(t[i] * s[6] for i, s in enumerate(shapes)))
    print(tot); fo.write(tot + "\n")
PY
