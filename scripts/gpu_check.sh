#!/bin/bash
# One GPU visit: smoke, -m gpu tests, a short bench.  Logs land in gpurun_out/.  Tight timeouts: a hung kernel must
# not burn the GPU budget.
mkdir -p gpurun_out
(timeout ${SMOKE_TIMEOUT:-150} python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log)
tail -3 gpurun_out/smoke.log
if ! grep -q "smoke rc=0" gpurun_out/smoke.log; then echo "SMOKE FAILED - skipping the rest"; exit 1; fi
timeout ${PYTEST_TIMEOUT:-600} python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
tail -${TAIL:-30} gpurun_out/pytest_gpu.log
timeout ${BENCH_TIMEOUT:-300} python bench.py ${BENCH_ARGS:---blocks 16384 --steps 3 --warmup 1} > gpurun_out/bench.log 2>&1
tail -2 gpurun_out/bench.log
