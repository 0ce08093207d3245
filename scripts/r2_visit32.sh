#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "variants_agree or limited or big or corrupted" 2>&1 | tail -3 > gpurun_out/r2v32_tests.txt
OUT=gpurun_out/r2v32.jsonl bash scripts/ab_variants.sh
