#!/bin/bash
mkdir -p gpurun_out
for np in 1 2; do for data in html; do
SNAPPIER_HIP_LIB=$PWD/snappier_amd/variants/libsnappier_hip_wprof2.so NP=$np DATA=$data BLOCKS=4096 timeout 120 python scripts/prof_compress_win.py 2>&1 | tail -1 | tee -a gpurun_out/r2v2_prof.jsonl
done; done
