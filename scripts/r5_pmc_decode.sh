#!/bin/bash
# Round 5: PMC passes over ONE decode launch per mode (rocprofv3 --pmc, separate passes; no trace domains).
#   gpurun -- 'MODES="chains chains_r04" bash scripts/r5_pmc_decode.sh'   -> one line per (mode, counter) on stdout and in gpurun_out/r5_pmc_decode.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_decode
mkdir -p $OUT
PASSES=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
        "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
        "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE"
        ${EXTRA_PASSES})
for m in ${MODES:-chains chains_r04}; do
  i=0
  for p in "${PASSES[@]}"; do
    d=$OUT/${m}_$i
    rm -rf $d
    (cd /tmp && DATA=${DATA:-html} SNAPPIER_HIP_DECODE=$m SNAPPIER_HIP_TABLE_TRIES=1 REPS=1 timeout 300 rocprofv3 --pmc $p -d $d -o pmc --output-format csv -- python $OLDPWD/scripts/time_decompress.py ${BLOCKS:-163840} > /dev/null 2>&1)
    f=$(find $d -name "*counter_collection.csv" | head -1)
    python - "$f" "$m" <<'PY' | tee -a gpurun_out/r5_pmc_decode.txt
import csv, sys, collections
f, m = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float)
try:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if ("k_decode_chains" in k or "k_decompress_chains" in k) :
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
except Exception as e:
    print("pmc pass failed:", e)
for c, v in sorted(acc.items()):
    print(f"{m:11s} {c:26s} {v:.6g}")
PY
    i=$((i+1))
  done
done
