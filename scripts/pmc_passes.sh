#!/bin/bash
# PMC passes over one bench step (full configs[1] workload), a few counters per pass (large sets crash the profiler).
#   gpurun -- 'bash scripts/pmc_passes.sh "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TA_TA_BUSY_sum" ...'
# Prints, per pass, the per-kernel counter totals of the codec kernels; raw CSVs stay under gpurun_out/pmc_<n>/.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for set in "$@"; do
  i=$((i+1))
  d=gpurun_out/pmc_$i
  rm -rf $d
  BENCH_NO_PLAIN=1 timeout ${PMC_TIMEOUT:-120} rocprofv3 --pmc $set -d $d -o pmc --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-live-traffic ${BENCH_ARGS:-} > gpurun_out/pmc_$i.log 2>&1
  echo "== pass $i: $set (rc=$?)"
  python - "$d" <<'PY'
import csv, glob, re, sys, collections
acc = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(r"(k_(?:compress|decompress|crc|frame|tag)[a-z_0-9]*)", k)
        if m:
            acc[(m.group(1), r["Counter_Name"])] += float(r["Counter_Value"])
for (k, c), v in sorted(acc.items()):
    print(f"   {k:20s} {c:32s} {v:.6g}")
PY
done
