#!/bin/bash
# round 2, first GPU visit of the window compressor: parity (layout matrix + fuzz) then timing against the lane kernel
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x -k "(layouts and win) or (fuzz_compress and win)" > gpurun_out/r2v1_pytest.log 2>&1
tail -15 gpurun_out/r2v1_pytest.log
for cfg in "win 2" "win 1" "lanes 2"; do
  set -- $cfg
  SNAPPIER_HIP_COMPRESS=$1 SNAPPIER_HIP_WIN_NP=$2 timeout 300 python scripts/time_compress.py ${NB:-163840} 2>&1 | tail -1 | tee -a gpurun_out/r2v1_time.jsonl
done
