for rep in 1 2; do for l in 0 1024 2048 3328; do
DATA=html SNAPPIER_HIP_DEC_LDS=$l timeout 300 python scripts/time_decompress.py 163840 2>&1 | tail -1 | sed "s/}$/, \"dec_lds\": $l}/" | tee -a gpurun_out/occ_sweep.jsonl
done; done
