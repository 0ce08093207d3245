#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r2v21_tests.txt
timeout 600 python bench.py 2>gpurun_out/r2v21_bench.err | tail -1 > gpurun_out/r2v21_bench.json
for d in html low mixed; do for m in queued chains; do
DATA=$d SNAPPIER_HIP_DECODE=$m timeout 300 python scripts/time_decompress.py 163840 2>&1 | tail -1 | tee -a gpurun_out/r2v21_times.jsonl
done; done
bash scripts/pmc_passes.sh "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" > gpurun_out/r2v21_pmc.txt 2>&1
