#!/bin/bash
# SQ counters of the tower attention kernels (run on the MI355X box from the repo root): one counter per pass
OUT=gpurun_out/pmc_attn
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 -L > $OUT/avail.txt 2>&1 || true
for c in GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
         SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT \
         SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_WAVES; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT -o pmc_$c -- python3 tools/bench_attn.py 64 > $OUT/pmc_$c.log 2>&1 || echo "$c failed"
  echo "$c done"
done
python3 - <<'PY'
import csv, glob, os, collections
out = "gpurun_out/pmc_attn"
res = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "attn_kernel" not in k and "attn32_kernel" not in k: continue
        key = ("32x32 " if "attn32" in k else "16x16 ") + ("d72" if ("Li96E" in k or "Li80E" in k) else "d64")
        res[key].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for key, d in res.items():
        for c, v in sorted(d.items()):
            line = "%s %-28s avg/launch %.4g (n=%d)" % (key, c, sum(v) / len(v), len(v))
            print(line); fo.write(line + "\n")
PY
