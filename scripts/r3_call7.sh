#!/bin/bash
cd "$(dirname "$0")/.."
bash scripts/r3_final.sh > gpurun_out/r03p_final.log 2>&1
timeout 900 python scripts/small_compress_sweep.py 256 1024 > gpurun_out/r03p_small_compress_sweep.jsonl 2> gpurun_out/r03p_small_compress_sweep.err
tail -40 gpurun_out/r03p_final.log | cut -c1-600; cat gpurun_out/r03p_small_compress_sweep.jsonl | cut -c1-300
