#!/bin/bash
# Round 3, GPU call 3: decoder variants (code placement, priority, overlap), new bench.py end to end.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=${TAG:-r03c}
OUT=gpurun_out/${T}_ab_decode.jsonl DATASETS="html" AB_REPS="1 2 3" REPS=6 bash scripts/ab_variants.sh > /dev/null 2>&1
timeout 900 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 3000 gpurun_out/${T}_bench.json; tail -3 gpurun_out/${T}_bench.err
