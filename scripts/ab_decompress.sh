#!/bin/bash
# A/B of decompressor variants on ONE box, interleaved: every snappier_amd/variants/libsnappier_hip_*.so runs bench.py
# (3 steps) REPS times in alternating order; prints decompress GB/s.   gpurun -- 'bash scripts/ab_decompress.sh'
cd "$(dirname "$0")/.."
for rep in ${REPS:-1 2 3}; do
  for lib in $( [ $((rep % 2)) = 1 ] && ls snappier_amd/variants/libsnappier_hip_*.so || ls -r snappier_amd/variants/libsnappier_hip_*.so ); do
    echo "$(basename $lib) $(SNAPPIER_HIP_LIB=$PWD/$lib SNAPPIER_HIP_TABLE_TRIES=1 timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["decompress_GBps"], d["roofline_decompress"]["avg_launch_ms"])')"
  done
done
