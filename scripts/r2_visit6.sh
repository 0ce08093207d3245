#!/bin/bash
# full -m gpu suite + default bench + the bench's other entry points
mkdir -p gpurun_out
(timeout 200 python __graft_entry__.py smoke > gpurun_out/r2v6_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r2v6_smoke.log); tail -2 gpurun_out/r2v6_smoke.log
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/r2v6_pytest.log 2>&1; tail -25 gpurun_out/r2v6_pytest.log
timeout 400 python bench.py > gpurun_out/r2v6_bench.json 2> gpurun_out/r2v6_bench.err; tail -c 1500 gpurun_out/r2v6_bench.json; tail -3 gpurun_out/r2v6_bench.err
timeout 400 python bench.py --config 5 --no-cpu-baseline --steps 3 > gpurun_out/r2v6_bench5.json 2> gpurun_out/r2v6_bench5.err; tail -c 600 gpurun_out/r2v6_bench5.json; tail -3 gpurun_out/r2v6_bench5.err
timeout 200 python bench.py --gpus 2 --steps 1 --blocks 1024 > gpurun_out/r2v6_bench_g2.log 2>&1; tail -5 gpurun_out/r2v6_bench_g2.log
