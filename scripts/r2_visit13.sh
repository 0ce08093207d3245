#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2v13_pytest.log 2>&1; tail -6 gpurun_out/r2v13_pytest.log
timeout 300 python scripts/small_blocks.py 256 1024 4096 2>&1 | grep block_bytes | tee gpurun_out/r2v13_small.jsonl
SNAPPIER_HIP_SMALL_MAX=16384 timeout 300 python scripts/small_blocks.py 16384 2>&1 | grep block_bytes | tee -a gpurun_out/r2v13_small.jsonl
timeout 600 python scripts/bench_configs.py 4 2>&1 | grep "configs\[" | tee gpurun_out/r2v13_cfg4.json
SNAPPIER_HIP_OVERLAP_CRC=0 timeout 600 python scripts/bench_configs.py 4 2>&1 | grep "configs\[" | tee -a gpurun_out/r2v13_cfg4.json
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['compress_GBps'], d['decompress_GBps'], d['roofline_decompress']['avg_launch_ms'])" | tee gpurun_out/r2v13_bench.txt
