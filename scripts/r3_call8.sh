#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 300 python scripts/small_blocks.py 64 128 256 512 1024 4096 2>&1 | grep block_bytes > gpurun_out/r03r_small_blocks.jsonl; cat gpurun_out/r03r_small_blocks.jsonl
SNAPPIER_HIP_SLICE=65536 timeout 300 python scripts/small_blocks.py 64 128 256 2>&1 | grep block_bytes > gpurun_out/r03r_small_blocks_slice65536.jsonl; cat gpurun_out/r03r_small_blocks_slice65536.jsonl
