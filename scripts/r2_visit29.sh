#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r2v29_tests.txt
for d in html low mixed; do
DATA=$d timeout 300 python scripts/time_decompress.py 163840 2>&1 | tail -1 | tee -a gpurun_out/r2v29_times.jsonl
done
