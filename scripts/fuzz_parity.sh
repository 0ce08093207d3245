#!/bin/bash
# Long differential-fuzz session on the GPU box (tests/test_gpu_fuzz.py scaled up):  gpurun -- 'bash scripts/fuzz_parity.sh'
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/fuzz_log.jsonl
FUZZ_SEED=${FUZZ_SEED:-0} FUZZ_ROUNDS=${FUZZ_ROUNDS:-40} FUZZ_BLOCKS=${FUZZ_BLOCKS:-2048} timeout ${FUZZ_TIMEOUT:-1500} python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/fuzz_parity.log
