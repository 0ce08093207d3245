#!/bin/bash
# round 6's final measurement set (one GPU visit, PRODUCT library only): smoke, -m gpu tests, the bench line (cpu baseline, live traffic, other configs,
# decode-alone + clocks), the same under rocprofv3 --kernel-trace --stats, decoder per data set and per corpus file, streams built against the decoder,
# compress by batch size and layout, host API, small blocks, PMC passes of the decoder and of the dual compressor, two long fuzz seeds.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${TAG:-r06z}
(timeout 200 python __graft_entry__.py smoke > gpurun_out/${T}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${T}_smoke.log); tail -2 gpurun_out/${T}_smoke.log
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/${T}_pytest.log 2>&1; tail -2 gpurun_out/${T}_pytest.log; tail -4 gpurun_out/${T}_pytest.log > gpurun_out/${T}_pytest_tail.txt
timeout 600 python bench.py > gpurun_out/${T}_bench_line.json 2> gpurun_out/${T}_bench.err; tail -c 300 gpurun_out/${T}_bench_line.json; echo
rm -rf gpurun_out/${T}_prof
(cd /tmp && BENCH_NO_PLAIN=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/${T}_prof -o bench -- python $OLDPWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-live-traffic --no-other-configs > $OLDPWD/gpurun_out/${T}_bench_line_under_rocprof.json 2> $OLDPWD/gpurun_out/${T}_rocprof.err)
f=$(find gpurun_out/${T}_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${T}_bench_kernel_stats.csv; head -5 gpurun_out/${T}_bench_kernel_stats.csv | cut -c1-160
timeout 600 python bench.py --thorough-search --no-cpu-baseline --no-live-traffic --no-other-configs > gpurun_out/${T}_bench_line_thorough_search.json 2>> gpurun_out/${T}_bench.err; tail -c 200 gpurun_out/${T}_bench_line_thorough_search.json; echo
for d in html low mixed; do DATA=$d REPS=6 timeout 300 python scripts/time_decompress.py 163840 2>&1 | tail -1; done > gpurun_out/${T}_decode_by_data.jsonl; cat gpurun_out/${T}_decode_by_data.jsonl
MODES="default" timeout 600 python scripts/decode_by_file.py 32768 2>&1 | grep '"file"' > gpurun_out/${T}_decode_by_file.jsonl; cat gpurun_out/${T}_decode_by_file.jsonl | cut -c1-200
timeout 300 python scripts/adversarial_streams.py 2>&1 | grep -E "^\{" > gpurun_out/${T}_adversarial_streams.jsonl; cat gpurun_out/${T}_adversarial_streams.jsonl | cut -c1-200
timeout 900 python scripts/compress_by_batch.py 1024 2048 4096 6144 8192 12288 16383 24576 32768 65536 163840 2>&1 | grep blocks > gpurun_out/${T}_compress_by_batch.jsonl; cat gpurun_out/${T}_compress_by_batch.jsonl | cut -c1-420
timeout 600 python scripts/host_api_rates.py 65536 4194304 268435456 1073741824 2>&1 | grep bytes > gpurun_out/${T}_host_api.jsonl; cat gpurun_out/${T}_host_api.jsonl
timeout 300 python scripts/small_blocks.py 64 256 1024 4096 2>&1 | grep block_bytes > gpurun_out/${T}_small_blocks.jsonl; cat gpurun_out/${T}_small_blocks.jsonl
# PMC: the decoder (one launch per pass), instructions per block
OUT=$PWD/gpurun_out/pmc6
mkdir -p $OUT; : > gpurun_out/${T}_pmc_decode.txt
PASSES=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
        "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
        "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE")
i=0
for p in "${PASSES[@]}"; do
  d=$OUT/dec_$i; rm -rf $d
  (cd /tmp && DATA=html REPS=1 timeout 300 rocprofv3 --pmc $p -d $d -o pmc --output-format csv -- python $OLDPWD/scripts/time_decompress.py 163840 > /dev/null 2>&1)
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY' | tee -a gpurun_out/${T}_pmc_decode.txt
import csv, sys, collections
acc = collections.defaultdict(float)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if "k_decode_chains" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
except Exception as e:
    print("pmc pass failed:", e)
for c, v in sorted(acc.items()):
    print(f"k_decode_chains {c:26s} {v:.6g} per launch   {v / 163840:.6g} per block")
PY
  i=$((i+1))
done
for s in 12 13; do FUZZ_SEED=$s FUZZ_ROUNDS=24 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -1; cp gpurun_out/fuzz_log.jsonl gpurun_out/${T}_fuzz_log_seed$s.jsonl; done
