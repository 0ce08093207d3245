#!/bin/bash
# the round's final measurement set (one GPU visit): smoke, -m gpu tests, bench line with and without rocprofv3, other configs,
# small blocks, host API, layout sweep
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 200 python __graft_entry__.py smoke > gpurun_out/r2p_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r2p_smoke.log); tail -2 gpurun_out/r2p_smoke.log
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2p_pytest.log 2>&1; tail -2 gpurun_out/r2p_pytest.log
timeout 400 python bench.py > gpurun_out/r2p_bench_line.json 2> gpurun_out/r2p_bench.err; tail -c 300 gpurun_out/r2p_bench_line.json
rm -rf gpurun_out/r2p_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2p_prof -o bench -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r2p_bench_line_under_rocprof.json 2> gpurun_out/r2p_rocprof.err
f=$(find gpurun_out/r2p_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2p_bench_kernel_stats.csv; head -6 gpurun_out/r2p_bench_kernel_stats.csv | cut -c1-200
timeout 600 python scripts/bench_configs.py 2>&1 | grep "config" > gpurun_out/r2p_other_configs.jsonl; cat gpurun_out/r2p_other_configs.jsonl | cut -c1-400
timeout 400 python bench.py --config 5 --no-cpu-baseline --steps 3 > gpurun_out/r2p_bench_config5.json 2>/dev/null; tail -c 400 gpurun_out/r2p_bench_config5.json
timeout 300 python scripts/small_blocks.py 256 512 1024 4096 16384 65536 2>&1 | grep block_bytes > gpurun_out/r2p_small_blocks.jsonl; cat gpurun_out/r2p_small_blocks.jsonl
timeout 600 python scripts/host_api_rates.py 65536 1048576 16777216 268435456 1073741824 2>&1 | grep bytes > gpurun_out/r2p_host_api.jsonl; cat gpurun_out/r2p_host_api.jsonl
for nbk in 1024 4096 8192 16383 16384 32768 65536; do timeout 200 python scripts/time_compress.py $nbk 2>&1 | tail -1; done > gpurun_out/r2p_compress_by_batch.jsonl; cat gpurun_out/r2p_compress_by_batch.jsonl
# decompressor front ends side by side, and the PMC passes the traffic / instruction-mix JSON is made from
for d in html low mixed; do for m in queued chains; do DATA=$d SNAPPIER_HIP_DECODE=$m timeout 300 python scripts/time_decompress.py 163840 2>&1 | tail -1; done; done > gpurun_out/r2p_decode_front_ends.jsonl; cat gpurun_out/r2p_decode_front_ends.jsonl
bash scripts/pmc_passes.sh "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS" "TA_BUSY_avr TD_TD_BUSY_sum" > gpurun_out/r2p_pmc_passes.txt 2>&1; grep -c "k_" gpurun_out/r2p_pmc_passes.txt
