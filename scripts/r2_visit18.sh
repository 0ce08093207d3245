#!/bin/bash
mkdir -p gpurun_out
for nb in 20000 32768 65536; do for h in 0 200 350 500; do
SNAPPIER_HIP_HYBRID=$h timeout 200 python scripts/time_compress.py $nb 2>&1 | tail -1 | sed "s/}$/, \"hybrid_permille\": $h}/" | tee -a gpurun_out/r2v18_hybrid_mid.jsonl
done; done
SNAPPIER_HIP_LIB=scripts/_bin/libsnappier_hip_dprof.so BLOCKS=16384 timeout 300 python scripts/prof_decompress.py > gpurun_out/r2v18_dprof.txt 2>&1
