#!/bin/bash
mkdir -p gpurun_out
for m in chains queued; do SNAPPIER_HIP_DECODE=$m timeout 600 python scripts/adversarial_streams.py 16384 2>&1 | grep stream | tee -a gpurun_out/r2v33_adversarial.jsonl; done
for bs in 256 512 1024; do for sm in 512 0; do SNAPPIER_HIP_SMALL_MAX=$sm timeout 300 python scripts/small_blocks.py $bs 2>&1 | grep block_bytes | sed "s/}$/, \"small_max\": $sm}/" | tee -a gpurun_out/r2v33_small.jsonl; done; done
