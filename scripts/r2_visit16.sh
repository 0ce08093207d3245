#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x -k "(layouts and win) or (fuzz_compress and win) or small_and_edge or corpus" > gpurun_out/r2v16_pytest.log 2>&1; tail -3 gpurun_out/r2v16_pytest.log
SNAPPIER_HIP_LIB=$PWD/snappier_amd/variants/libsnappier_hip_wprof2.so NP=1 DATA=html BLOCKS=4096 timeout 120 python scripts/prof_compress_win.py 2>&1 | tail -1 | tee -a gpurun_out/r2v16_prof.jsonl
SNAPPIER_HIP_COMPRESS=win timeout 300 python scripts/time_compress.py 8192 2>&1 | tail -1 | tee -a gpurun_out/r2v16_time.jsonl
