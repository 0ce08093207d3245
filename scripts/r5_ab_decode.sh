#!/bin/bash
# Round 5: the new decoder against the round-4 one in one process order, interleaved.   gpurun -- 'bash scripts/r5_ab_decode.sh'
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=${OUT:-gpurun_out/r5_ab_decode.jsonl}
for rep in ${AB_REPS:-1 2}; do for d in ${DATASETS:-html low mixed}; do for m in ${MODES:-chains chains_r04}; do
DATA=$d SNAPPIER_HIP_DECODE=$m timeout 300 python scripts/time_decompress.py ${BLOCKS:-163840} 2>&1 | tail -1 | tee -a $OUT
done; done; done
