#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2v14_pytest.log 2>&1; tail -3 gpurun_out/r2v14_pytest.log
PMC_TIMEOUT=200 bash scripts/pmc_passes.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE" 2>&1 | tee gpurun_out/r2v14_pmc.txt
