#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_big_blocks.py tests/test_abi_conformance.py -m gpu -q -x > gpurun_out/r2v9_pytest.log 2>&1; tail -3 gpurun_out/r2v9_pytest.log
timeout 600 python scripts/host_api_rates.py 65536 16777216 268435456 1073741824 2>&1 | grep bytes | tee gpurun_out/r2v9_host_api.jsonl
SNAPPIER_HIP_PINNED=0 timeout 600 python scripts/host_api_rates.py 268435456 1073741824 2>&1 | grep bytes | tee -a gpurun_out/r2v9_host_api.jsonl
SNAPPIER_HIP_COPY_THREADS=15 timeout 600 python scripts/host_api_rates.py 1073741824 2>&1 | grep bytes | tee -a gpurun_out/r2v9_host_api.jsonl
