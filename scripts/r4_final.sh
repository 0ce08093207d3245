#!/bin/bash
# round 4's final measurement set (one GPU visit): smoke, -m gpu tests, bench line with and without rocprofv3, other configs,
# small blocks, host API, compress by batch size, decoder front ends, PMC passes (traffic + instruction mix)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${TAG:-r04z}
(timeout 200 python __graft_entry__.py smoke > gpurun_out/${T}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${T}_smoke.log); tail -2 gpurun_out/${T}_smoke.log
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/${T}_pytest.log 2>&1; tail -2 gpurun_out/${T}_pytest.log
timeout 600 python bench.py > gpurun_out/${T}_bench_line.json 2> gpurun_out/${T}_bench.err; tail -c 300 gpurun_out/${T}_bench_line.json
rm -rf gpurun_out/${T}_prof
BENCH_NO_PLAIN=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof -o bench -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-live-traffic > gpurun_out/${T}_bench_line_under_rocprof.json 2> gpurun_out/${T}_rocprof.err
f=$(find gpurun_out/${T}_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${T}_bench_kernel_stats.csv; head -6 gpurun_out/${T}_bench_kernel_stats.csv | cut -c1-200
timeout 600 python scripts/bench_configs.py 2>&1 | grep "config" > gpurun_out/${T}_other_configs.jsonl; cat gpurun_out/${T}_other_configs.jsonl | cut -c1-400
timeout 600 python bench.py --config5-lines --no-cpu-baseline --no-live-traffic --steps 3 > gpurun_out/${T}_bench_config5_lines.json 2>/dev/null; tail -c 900 gpurun_out/${T}_bench_config5_lines.json
timeout 300 python scripts/small_blocks.py 64 96 128 192 256 384 512 768 1024 4096 16384 65536 2>&1 | grep block_bytes > gpurun_out/${T}_small_blocks.jsonl; cat gpurun_out/${T}_small_blocks.jsonl
timeout 600 python scripts/host_api_rates.py 65536 1048576 4194304 16777216 268435456 1073741824 2>&1 | grep bytes > gpurun_out/${T}_host_api.jsonl; cat gpurun_out/${T}_host_api.jsonl
timeout 900 python scripts/compress_by_batch.py 1024 2048 4096 8192 16383 16384 32768 65536 163840 2>&1 | grep blocks > gpurun_out/${T}_compress_by_batch.jsonl; cat gpurun_out/${T}_compress_by_batch.jsonl
for d in html low mixed; do for m in queued chains ring; do DATA=$d SNAPPIER_HIP_DECODE=$m timeout 300 python scripts/time_decompress.py 163840 2>&1 | tail -1; done; done > gpurun_out/${T}_decode_front_ends.jsonl; cat gpurun_out/${T}_decode_front_ends.jsonl
bash scripts/pmc_passes.sh "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS" "TA_BUSY_avr TD_TD_BUSY_sum" "TCC_HIT_sum TCC_MISS_sum" > gpurun_out/${T}_pmc_passes.txt 2>&1; grep -c "k_" gpurun_out/${T}_pmc_passes.txt
