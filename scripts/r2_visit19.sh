#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "variants_agree" 2>&1 | tail -15 > gpurun_out/r2v19_tests.txt
for d in html low mixed; do for m in queued chains; do
DATA=$d SNAPPIER_HIP_DECODE=$m timeout 300 python scripts/time_decompress.py 32768 2>&1 | tail -1 | tee -a gpurun_out/r2v19_times.jsonl
done; done
SNAPPIER_HIP_DECODE=chains SNAPPIER_HIP_LIB=scripts/_bin/libsnappier_hip_dprof.so BLOCKS=8192 timeout 300 python scripts/prof_decompress.py > gpurun_out/r2v19_dprof.txt 2>&1
