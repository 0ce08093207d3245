#!/bin/bash
# PMC passes over the lane compressor on 1 GiB of 256-byte blocks (scripts/small_blocks.py 256): where does a small-block launch spend its time?
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TA_BUSY_avr TD_TD_BUSY_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  d=gpurun_out/pmcs_$i
  rm -rf $d
  timeout 200 rocprofv3 --pmc $set -d $d -o pmc --output-format csv -- python scripts/small_blocks.py ${BS:-256} > gpurun_out/pmcs_$i.log 2>&1
  echo "== pass $i: $set (rc=$?)"
  python - "$d" <<'PY'
import csv, glob, re, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_compress_lanes)", r["Kernel_Name"])
        if m:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for c, v in sorted(acc.items()):
    print(f"   k_compress_lanes {c:36s} {v:.6g}   over {n[c]} launches")
PY
done
