#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2v11_pytest.log 2>&1; tail -3 gpurun_out/r2v11_pytest.log
timeout 300 python scripts/small_blocks.py 256 1024 4096 16384 65536 2>&1 | grep block_bytes | tee gpurun_out/r2v11_small.jsonl
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['compress_GBps'], d['decompress_GBps'], d['roofline_compress']['avg_launch_ms'])" | tee gpurun_out/r2v11_bench.txt
