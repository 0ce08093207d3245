#!/bin/bash
mkdir -p gpurun_out
for rep in 1 2; do for v in nopf pf7 pf8; do for d in html mixed; do
DATA=$d SNAPPIER_HIP_DECODE=chains SNAPPIER_HIP_LIB=$PWD/snappier_amd/variants/libsnappier_hip_$v.so timeout 300 python scripts/time_decompress.py 65536 2>&1 | tail -1 | sed "s/}$/, \"variant\": \"$v\"}/" | tee -a gpurun_out/r2v20_times.jsonl
done; done; done
DATA=html SNAPPIER_HIP_DECODE=queued timeout 300 python scripts/time_decompress.py 65536 2>&1 | tail -1 | tee -a gpurun_out/r2v20_times.jsonl
