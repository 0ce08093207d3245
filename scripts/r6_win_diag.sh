#!/bin/bash
# Round 6, VERDICT r5 item 2: what is k_compress_win bound by?  Phase clock (s_memtime per phase: a -DSNP_W_PROF=2 variant, built BEFORE the
# GPU visit: LAB=1 SRC=compress_win.hip scripts/build_variant.sh wprof2 -DSNP_W_PROF=2), PMC passes for both table forms (LDS: win, global
# slot: wing), and the kernel's time at three batch sizes.
#   gpurun -- 'bash scripts/r6_win_diag.sh'      -> gpurun_out/r06_compress_win_{phase_clock.jsonl,pmc.txt,by_batch.jsonl}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
NB=${NB:-8192}
PROF=snappier_amd/variants/libsnappier_hip_wprof2.so
: > gpurun_out/r06_compress_win_phase_clock.jsonl
if [ -f $PROF ]; then
  for m in win wing; do
    for d in html mixed; do
      SNAPPIER_HIP_LIB=$PWD/$PROF SNAPPIER_HIP_COMPRESS_FORM=$m NP=1 DATA=$d BLOCKS=$NB timeout 300 python scripts/prof_compress_win.py 2>&1 | tail -1 >> gpurun_out/r06_compress_win_phase_clock.jsonl
    done
  done
fi
cat gpurun_out/r06_compress_win_phase_clock.jsonl | cut -c1-600
: > gpurun_out/r06_compress_win_by_batch.jsonl
for m in win wing; do
  for n in 1024 4096 8192 16383; do
    SNAPPIER_HIP_LAB=1 SNAPPIER_HIP_COMPRESS=$m timeout 300 python scripts/time_compress.py $n 2>&1 | tail -1 | sed "s/^{/{\"form\": \"$m\", /" >> gpurun_out/r06_compress_win_by_batch.jsonl
  done
done
cat gpurun_out/r06_compress_win_by_batch.jsonl
OUT=$PWD/gpurun_out/pmc_win6
mkdir -p $OUT
PASSES=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
        "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAVES SQ_WAIT_INST_LDS"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
        "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr"
        "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
        "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
        "FETCH_SIZE" "WRITE_SIZE")
: > gpurun_out/r06_compress_win_pmc.txt
for m in win wing; do
  i=0
  for p in "${PASSES[@]}"; do
    d=$OUT/${m}_$i
    rm -rf $d
    (cd /tmp && SNAPPIER_HIP_LAB=1 SNAPPIER_HIP_COMPRESS=$m timeout 300 rocprofv3 --pmc $p -d $d -o pmc --output-format csv -- python $OLDPWD/scripts/time_compress.py $NB > /dev/null 2>&1)
    f=$(find $d -name "*counter_collection.csv" | head -1)
    python - "$f" "$m" "$NB" <<'PY' | tee -a gpurun_out/r06_compress_win_pmc.txt
import csv, sys, collections
f, m, nb = sys.argv[1], sys.argv[2], int(sys.argv[3])
acc = collections.defaultdict(float); n = collections.defaultdict(int)
try:
    for r in csv.DictReader(open(f)):
        if "k_compress_win" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
except Exception as e:
    print("pmc pass failed:", m, e)
for c, v in sorted(acc.items()):
    per = v / max(n[c], 1)
    print(f"{m:5s} {c:34s} {per:.6g} per launch ({n[c]} launches)  {per / nb:.6g} per fragment")
PY
    i=$((i+1))
  done
done
