#!/bin/bash
# One GPU visit: smoke, -m gpu tests, a short bench.  Logs land in gpurun_out/.
mkdir -p gpurun_out
(timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log)
tail -3 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
tail -${TAIL:-40} gpurun_out/pytest_gpu.log
timeout 600 python bench.py ${BENCH_ARGS:---blocks 16384 --steps 3 --warmup 1} > gpurun_out/bench.log 2>&1
tail -2 gpurun_out/bench.log
