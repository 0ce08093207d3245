#!/bin/bash
mkdir -p gpurun_out
for bs in 320 384 448; do for sm in 512 0; do SNAPPIER_HIP_SMALL_MAX=$sm timeout 300 python scripts/small_blocks.py $bs 2>&1 | grep block_bytes | sed "s/}$/, \"small_max\": $sm}/" | tee -a gpurun_out/r2v34_small.jsonl; done; done
