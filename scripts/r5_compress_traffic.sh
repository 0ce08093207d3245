#!/bin/bash
# what does zeroing the hash tables cost?  Ablation 8192 of the lab lane compressor (tables zeroed TWICE: the difference is the ceiling of
# what an epoch tag could save), interleaved with the unchanged kernel on one workspace; then FETCH_SIZE / WRITE_SIZE of both (rocprofv3 --pmc, one
# counter per pass).   gpurun -- 'bash scripts/r5_compress_traffic.sh'
cd "$(dirname "$0")/.."
export TMPDIR=/tmp SNAPPIER_HIP_LIB=$PWD/snappier_amd/variants/libsnappier_hip_clablate.so SNAPPIER_HIP_TABLE_TRIES=2
mkdir -p gpurun_out
REPS=6 timeout 300 python scripts/ab_compress_ablate.py 0 8192 | tail -1 > gpurun_out/r5_compress_zero_ablation.json; cat gpurun_out/r5_compress_zero_ablation.json
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmc_cl_$c; rm -rf $d
  (cd /tmp && REPS=2 timeout 300 rocprofv3 --pmc $c -d $d -o pmc --output-format csv -- python $OLDPWD/scripts/ab_compress_ablate.py 0 8192 > /dev/null 2>&1)
  python - "$d" "$c" <<'PY' | tee -a gpurun_out/r5_compress_zero_pmc.txt
import csv, glob, sys
d, c = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_compress_lanes" in r["Kernel_Name"] and r["Counter_Name"] == c:
            rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]) * 1024.0))
rows.sort()
# dispatch order of ab_compress_ablate.py 0 8192 with REPS=2: warm 0, warm 0, rep0: 0, 8192, rep1 (reversed): 8192, 0
names = ["0 (warm)", "0 (warm)", "0", "8192", "8192", "0"]
for (i, v), n in zip(rows, names):
    print(f"{c:11s} mask {n:9s} {v/1e9:8.2f} GB")
PY
done
