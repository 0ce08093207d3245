#!/bin/bash
# Interleaved same-box A/B of decoder front ends over the full 10 GiB: MODES="chains ring" DATA kinds html low mixed, REPS rounds.
#   gpurun -- 'bash scripts/ab_decode_modes.sh'
cd "$(dirname "$0")/.."
for rep in ${REPS:-1 2}; do
  for data in ${KINDS:-html low mixed}; do
    for m in ${MODES:-chains ring}; do
      DATA=$data SNAPPIER_HIP_DECODE=$m SNAPPIER_HIP_TABLE_TRIES=1 REPS=3 timeout 300 python scripts/time_decompress.py ${BLOCKS:-163840} 2>/dev/null | tail -1
    done
  done
done
