#!/bin/bash
mkdir -p gpurun_out
for h in 0 80 120 160 200 250; do
SNAPPIER_HIP_HYBRID=$h timeout 200 python scripts/time_compress.py 163840 2>&1 | tail -1 | sed "s/}$/, \"hybrid_permille\": $h}/" | tee -a gpurun_out/r2v15_hybrid.jsonl
done
