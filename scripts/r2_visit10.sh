#!/bin/bash
mkdir -p gpurun_out
for dec in queued lanes; do
SNAPPIER_HIP_DECODE=$dec timeout 300 python scripts/small_blocks.py 256 1024 4096 2>&1 | grep block_bytes | tee -a gpurun_out/r2v10_small.jsonl
done
SNAPPIER_HIP_COMPRESS=win timeout 300 python scripts/small_blocks.py 256 1024 4096 16384 2>&1 | grep block_bytes | sed 's/}$/, "compress_layout": "win"}/' | tee -a gpurun_out/r2v10_small.jsonl
# FENCED decision: full-size bench with and without the vmcnt drain
for f in 0 1; do SNAPPIER_HIP_FENCED=$f timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'fenced': $f, 'decompress_ms': d['roofline_decompress']['avg_launch_ms'], 'decompress_GBps': d['decompress_GBps']}))" | tee -a gpurun_out/r2v10_fenced.jsonl; done
for f in 0 1; do SNAPPIER_HIP_FENCED=$f timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'fenced': $f, 'decompress_ms': d['roofline_decompress']['avg_launch_ms'], 'decompress_GBps': d['decompress_GBps']}))" | tee -a gpurun_out/r2v10_fenced.jsonl; done
