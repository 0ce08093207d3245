#!/bin/bash
# Round 3, GPU call 2: full GPU suite on the rebuilt library (options API, small-block fixes), far-source ablation of the decoder,
# lane-compressor option combinations on html and the mixed corpus, per-file compress rates.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=${TAG:-r03b}
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/${T}_pytest.txt
OUT=gpurun_out/${T}_ab_decode.jsonl DATASETS="html" AB_REPS="1 2" bash scripts/ab_variants.sh > /dev/null 2>&1
timeout 600 python scripts/ab_compress_opts.py 23 87 215 > gpurun_out/${T}_ab_compress_html.json 2> gpurun_out/${T}_ab_compress.err
DATA=mixed timeout 600 python scripts/ab_compress_opts.py 23 87 215 > gpurun_out/${T}_ab_compress_mixed.json 2>> gpurun_out/${T}_ab_compress.err
timeout 900 python scripts/compress_by_file.py > gpurun_out/${T}_by_file_23.jsonl 2>> gpurun_out/${T}_ab_compress.err
SNAPPIER_HIP_CL_OPTS=87 timeout 900 python scripts/compress_by_file.py > gpurun_out/${T}_by_file_87.jsonl 2>> gpurun_out/${T}_ab_compress.err
tail -3 gpurun_out/${T}_pytest.txt; cat gpurun_out/${T}_ab_compress_html.json gpurun_out/${T}_ab_compress_mixed.json; tail -2 gpurun_out/${T}_ab_compress.err
