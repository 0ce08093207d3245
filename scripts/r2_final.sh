#!/bin/bash
# the round's final measurement set (one GPU visit): smoke, -m gpu tests, bench line with and without rocprofv3, other configs,
# small blocks, host API, layout sweep
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 200 python __graft_entry__.py smoke > gpurun_out/r2f_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r2f_smoke.log); tail -2 gpurun_out/r2f_smoke.log
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2f_pytest.log 2>&1; tail -2 gpurun_out/r2f_pytest.log
timeout 400 python bench.py > gpurun_out/r2f_bench_line.json 2> gpurun_out/r2f_bench.err; tail -c 300 gpurun_out/r2f_bench_line.json
rm -rf gpurun_out/r2f_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2f_prof -o bench -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r2f_bench_line_under_rocprof.json 2> gpurun_out/r2f_rocprof.err
f=$(find gpurun_out/r2f_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2f_bench_kernel_stats.csv; head -6 gpurun_out/r2f_bench_kernel_stats.csv | cut -c1-200
timeout 600 python scripts/bench_configs.py 2>&1 | grep "config" > gpurun_out/r2f_other_configs.jsonl; cat gpurun_out/r2f_other_configs.jsonl | cut -c1-400
timeout 400 python bench.py --config 5 --no-cpu-baseline --steps 3 > gpurun_out/r2f_bench_config5.json 2>/dev/null; tail -c 400 gpurun_out/r2f_bench_config5.json
timeout 300 python scripts/small_blocks.py 256 512 1024 4096 16384 65536 2>&1 | grep block_bytes > gpurun_out/r2f_small_blocks.jsonl; cat gpurun_out/r2f_small_blocks.jsonl
timeout 600 python scripts/host_api_rates.py 65536 1048576 16777216 268435456 1073741824 2>&1 | grep bytes > gpurun_out/r2f_host_api.jsonl; cat gpurun_out/r2f_host_api.jsonl
for nbk in 1024 4096 8192 16383 16384 32768 65536; do timeout 200 python scripts/time_compress.py $nbk 2>&1 | tail -1; done > gpurun_out/r2f_compress_by_batch.jsonl; cat gpurun_out/r2f_compress_by_batch.jsonl
