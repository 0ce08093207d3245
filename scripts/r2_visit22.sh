#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "variants_agree or limited or big" 2>&1 | tail -5 > gpurun_out/r2v22_tests.txt
for d in html low mixed; do for m in chains; do
DATA=$d SNAPPIER_HIP_DECODE=$m timeout 300 python scripts/time_decompress.py 163840 2>&1 | tail -1 | tee -a gpurun_out/r2v22_times.jsonl
done; done
SNAPPIER_HIP_LIB=scripts/_bin/libsnappier_hip_dprof.so BLOCKS=8192 timeout 300 python scripts/prof_decompress.py > gpurun_out/r2v22_dprof.txt 2>&1
