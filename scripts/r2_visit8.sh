#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "frame or stream" > gpurun_out/r2v8_pytest.log 2>&1; tail -3 gpurun_out/r2v8_pytest.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2v8_prof -o cfg4 -- python scripts/bench_configs.py 4 > gpurun_out/r2v8.log 2>&1
tail -1 gpurun_out/r2v8.log
f=$(find gpurun_out/r2v8_prof -name "*kernel_stats.csv" | head -1); grep -E "k_span|k_frame|k_crc|k_decompress" "$f" | cut -c1-60,150-400 | cut -d, -f1-4
