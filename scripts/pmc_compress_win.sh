#!/bin/bash
# PMC passes over the window compressor (LDS tables: win, global tables: wing) at NB fragments.   gpurun -- 'bash scripts/pmc_compress_win.sh'
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_win
mkdir -p $OUT
PASSES=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
        "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAVES SQ_INSTS_SMEM"
        "GRBM_GUI_ACTIVE TA_TA_BUSY_sum")
for m in ${MODES:-win wing}; do
  i=0
  for p in "${PASSES[@]}"; do
    d=$OUT/${m}_$i
    rm -rf $d
    (cd /tmp && SNAPPIER_HIP_COMPRESS=$m timeout 300 rocprofv3 --pmc $p -d $d -o pmc --output-format csv -- python $OLDPWD/scripts/time_compress.py ${NB:-16383} > /dev/null 2>&1)
    f=$(find $d -name "*counter_collection.csv" | head -1)
    python - "$f" "$m" <<'PY'
import csv, sys, collections
f, m = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.defaultdict(int)
try:
    for r in csv.DictReader(open(f)):
        if "k_compress_win" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
except Exception as e:
    print("pmc pass failed:", e)
for c, v in sorted(acc.items()):
    print(f"{m:5s} {c:26s} {v / max(n[c],1):.6g}  (per launch, {n[c]} launches)")
PY
    i=$((i+1))
  done
done
