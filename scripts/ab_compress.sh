#!/bin/bash
# A/B of kernel variants on ONE box: for each snappier_amd/variants/libsnappier_hip_*.so time configs[1] compress and
# (optionally, CONFIGS="5 3") the other configs.   gpurun -- 'bash scripts/ab_compress.sh'
cd "$(dirname "$0")/.."
for rep in ${REPS:-1 2}; do
  for lib in $( [ $((rep % 2)) = 1 ] && ls snappier_amd/variants/libsnappier_hip_*.so || ls -r snappier_amd/variants/libsnappier_hip_*.so ); do
    if [ "$lib" = default ]; then unset SNAPPIER_HIP_LIB; else export SNAPPIER_HIP_LIB=$PWD/$lib; fi
    echo "$lib $(timeout 120 python scripts/time_compress.py 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["compress_ms"][1:], d["GBps"])')"
    for c in ${CONFIGS:-}; do
      echo "   cfg$c $(timeout 200 python scripts/bench_configs.py $c 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d.get("compress_GBps"), d.get("decompress_GBps"))')"
    done
  done
done
