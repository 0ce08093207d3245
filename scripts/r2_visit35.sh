#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r2v35_tests.txt
for bs in 256 384 448 512; do timeout 300 python scripts/small_blocks.py $bs 2>&1 | grep block_bytes | tee -a gpurun_out/r2v35_small.jsonl; done
