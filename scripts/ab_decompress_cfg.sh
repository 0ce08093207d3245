#!/bin/bash
# like ab_decompress.sh but on the other configs (3 = low entropy, 5 = mixed corpus): prints compress/decompress GB/s
cd "$(dirname "$0")/.."
for lib in snappier_amd/variants/libsnappier_hip_*.so; do
  for c in ${CONFIGS:-3 5}; do
    echo "$(basename $lib) cfg$c $(SNAPPIER_HIP_LIB=$PWD/$lib timeout 200 python scripts/bench_configs.py $c 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d.get("decompress_GBps"))')"
  done
done
