#!/bin/bash
# SQ issue / wait counters of the persistent GEMM on two tower shapes, with and without its epilogue (TDC_GEMM_DEBUG=1)
# - what the waves of the main loop spend their cycles on.  One counter per pass; run on the MI355X box from the repo root.
OUT=gpurun_out/pmc_sq
mkdir -p $OUT
/opt/rocm/bin/hipcc -O2 -o $OUT/gemm_pmc tools/gemm_pmc.cpp -Ltdc-video_amd -ltdc_hip -Wl,-rpath,$PWD/tdc-video_amd || exit 1
printf "373248 3456 1152 0 0 0 27\n373760 8192 1536 3 0 0 40\n" > $OUT/shapes.txt
export TMPDIR=/tmp
for dbg in 0 1; do
  if [ $dbg = 1 ]; then export TDC_GEMM_DEBUG=1; fi
  for c in GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
           SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS \
           SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT -o d${dbg}_$c -- $OUT/gemm_pmc $OUT/shapes.txt 1 > $OUT/d${dbg}_$c.log 2>&1 || echo "$c failed"
  done
done
python3 - <<'PY'
import csv, glob, collections, re
out = "gpurun_out/pmc_sq"
res = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    dbg = re.search(r"/d(\d)_", f).group(1)
    launches = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm256p" not in r.get("Kernel_Name", ""): continue
        launches[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in launches.items():
        # launches alternate shape 0 / shape 1 (reps = 1 -> one launch per shape, plus a warm-up each)
        res[(dbg, c)] = v
with open(out + "/summary.txt", "w") as fo:
    for (dbg, c), v in sorted(res.items()):
        line = "debug=%s %-30s %s" % (dbg, c, " ".join("%.4g" % x for x in v))
        print(line); fo.write(line + "\n")
PY
