#!/bin/bash
# Round 3, GPU call 1: texture-path price list, parity of the V4 decoder (the default build), interleaved A/B of the decoder
# variants under snappier_amd/variants/, same-process A/B of the lane compressor's new option bits.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=${TAG:-r03a}
scripts/_bin/microbench_vmem_scatter > gpurun_out/${T}_vmem_scatter.jsonl 2>&1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/${T}_pytest.txt
OUT=gpurun_out/${T}_ab_decode.jsonl DATASETS="${DATASETS:-html mixed}" AB_REPS="${AB_REPS:-1 2}" bash scripts/ab_variants.sh > /dev/null 2>&1
timeout 600 python scripts/ab_compress_opts.py 23 55 87 151 119 > gpurun_out/${T}_ab_compress.json 2> gpurun_out/${T}_ab_compress.err
tail -3 gpurun_out/${T}_pytest.txt; cat gpurun_out/${T}_ab_compress.json; tail -2 gpurun_out/${T}_ab_compress.err
