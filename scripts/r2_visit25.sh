#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "variants_agree or limited or big or corrupted" 2>&1 | tail -3 > gpurun_out/r2v25_tests.txt
for d in html low mixed; do
DATA=$d timeout 300 python scripts/time_decompress.py 163840 2>&1 | tail -1 | tee -a gpurun_out/r2v25_times.jsonl
done
SNAPPIER_HIP_LIB=scripts/_bin/libsnappier_hip_dprof.so BLOCKS=8192 timeout 300 python scripts/prof_decompress.py > gpurun_out/r2v25_dprof.txt 2>&1
bash scripts/pmc_passes.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" > gpurun_out/r2v25_pmc.txt 2>&1
