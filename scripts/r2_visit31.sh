#!/bin/bash
mkdir -p gpurun_out
for rep in 1 2; do
for cfg in "" "SNAPPIER_HIP_FENCED=0" "SNAPPIER_HIP_DEC_LDS=1024" "SNAPPIER_HIP_DEC_LDS=2048" "SNAPPIER_HIP_DEC_LDS=3584" "SNAPPIER_HIP_DEC_LDS=6144"; do
env $cfg DATA=html timeout 300 python scripts/time_decompress.py 163840 2>&1 | tail -1 | sed "s/}$/, \"env\": \"$cfg\"}/" | tee -a gpurun_out/r2v31_times.jsonl
done; done
