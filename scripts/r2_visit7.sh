#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_abi_conformance.py -m gpu -q -x -k "frame or abi or stream" > gpurun_out/r2v7_pytest.log 2>&1; tail -15 gpurun_out/r2v7_pytest.log
timeout 600 python scripts/bench_configs.py > gpurun_out/r2v7_configs.log 2>&1; tail -12 gpurun_out/r2v7_configs.log
