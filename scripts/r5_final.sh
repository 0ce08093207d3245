#!/bin/bash
# round 5's final measurement set (one GPU visit): smoke, -m gpu tests, the bench line (full: cpu baseline, live traffic, other configs), the same under
# rocprofv3 --kernel-trace --stats, decoder per corpus file and on the streams built against it (round-5 kernel vs the round-4 one from the lab
# library), compress by batch size, host API, small blocks, PMC passes of the decoder.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${TAG:-r05zz}
(timeout 200 python __graft_entry__.py smoke > gpurun_out/${T}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${T}_smoke.log); tail -2 gpurun_out/${T}_smoke.log
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/${T}_pytest.log 2>&1; tail -2 gpurun_out/${T}_pytest.log
timeout 600 python bench.py > gpurun_out/${T}_bench_line.json 2> gpurun_out/${T}_bench.err; tail -c 300 gpurun_out/${T}_bench_line.json; echo
rm -rf gpurun_out/${T}_prof
(cd /tmp && BENCH_NO_PLAIN=1 BENCH_NO_DEFAULT_SEARCH=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/${T}_prof -o bench -- python $OLDPWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-live-traffic --no-other-configs > $OLDPWD/gpurun_out/${T}_bench_line_under_rocprof.json 2> $OLDPWD/gpurun_out/${T}_rocprof.err)
f=$(find gpurun_out/${T}_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${T}_bench_kernel_stats.csv; head -5 gpurun_out/${T}_bench_kernel_stats.csv | cut -c1-160
MODES="chains chains_r04" timeout 600 python scripts/decode_by_file.py 32768 2>&1 | grep '"file"' > gpurun_out/${T}_decode_by_file.jsonl; cat gpurun_out/${T}_decode_by_file.jsonl | cut -c1-220
for d in html low mixed; do for m in chains chains_r04; do DATA=$d SNAPPIER_HIP_DECODE=$m REPS=5 timeout 300 python scripts/time_decompress.py 163840 2>&1 | tail -1; done; done > gpurun_out/${T}_decode_new_vs_r04.jsonl; cat gpurun_out/${T}_decode_new_vs_r04.jsonl
for m in chains chains_r04; do SNAPPIER_HIP_DECODE=$m timeout 300 python scripts/adversarial_streams.py 2>&1 | grep -E "^\{"; done > gpurun_out/${T}_adversarial_streams.jsonl; cat gpurun_out/${T}_adversarial_streams.jsonl | cut -c1-200
timeout 900 python scripts/compress_by_batch.py 1024 4096 16383 16384 32768 65536 163840 2>&1 | grep blocks > gpurun_out/${T}_compress_by_batch.jsonl; cat gpurun_out/${T}_compress_by_batch.jsonl
timeout 600 python scripts/host_api_rates.py 65536 4194304 268435456 1073741824 2>&1 | grep bytes > gpurun_out/${T}_host_api.jsonl; cat gpurun_out/${T}_host_api.jsonl
timeout 300 python scripts/small_blocks.py 64 256 1024 4096 2>&1 | grep block_bytes > gpurun_out/${T}_small_blocks.jsonl; cat gpurun_out/${T}_small_blocks.jsonl
EXTRA_PASSES="" MODES="chains chains_r04" timeout 300 bash scripts/r5_pmc_decode.sh > gpurun_out/${T}_pmc_decode.txt 2>&1; grep -c SQ_ gpurun_out/${T}_pmc_decode.txt
