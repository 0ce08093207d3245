#!/bin/bash
# Interleaved A/B of every snappier_amd/variants/libsnappier_hip_*.so on one box: scripts/time_decompress.py per variant and data set.
#   gpurun -- 'OUT=gpurun_out/x.jsonl DATASETS="html mixed" bash scripts/ab_variants.sh'
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in ${AB_REPS:-1 2}; do for lib in snappier_amd/variants/libsnappier_hip_*.so; do for d in ${DATASETS:-html low mixed}; do
v=$(basename $lib .so); v=${v#libsnappier_hip_}
DATA=$d SNAPPIER_HIP_LIB=$PWD/$lib timeout 300 python scripts/time_decompress.py ${BLOCKS:-163840} 2>&1 | tail -1 | sed "s/}$/, \"variant\": \"$v\"}/" | tee -a ${OUT:-gpurun_out/ab_variants.jsonl}
done; done; done
