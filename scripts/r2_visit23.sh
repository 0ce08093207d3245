#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "variants_agree or limited or big" 2>&1 | tail -3 > gpurun_out/r2v23_tests.txt
for d in html low mixed; do
DATA=$d timeout 300 python scripts/time_decompress.py 163840 2>&1 | tail -1 | tee -a gpurun_out/r2v23_times.jsonl
done
bash scripts/pmc_passes.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "TA_TA_BUSY_sum" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" > gpurun_out/r2v23_pmc.txt 2>&1
