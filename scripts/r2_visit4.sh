#!/bin/bash
mkdir -p gpurun_out
for np in 1 2; do
SNAPPIER_HIP_LIB=$PWD/snappier_amd/variants/libsnappier_hip_wprof2.so NP=$np DATA=html BLOCKS=${BLOCKS:-4096} timeout 120 python scripts/prof_compress_win.py 2>&1 | tail -1 | tee -a gpurun_out/r2v4_prof.jsonl
done
SNAPPIER_HIP_LIB=$PWD/snappier_amd/variants/libsnappier_hip_wprof2.so NP=1 DATA=html BLOCKS=256 timeout 120 python scripts/prof_compress_win.py 2>&1 | tail -1 | tee -a gpurun_out/r2v4_prof.jsonl
