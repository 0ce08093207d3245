#!/bin/bash
# Round 6: same-process A/B of decoder variants (scripts/ab_libs.py), then the bench line (+ the same under rocprofv3 --kernel-trace --stats).
#   gpurun -- 'TAG=r06a VARIANTS="fmin" bash scripts/r6_ab_and_bench.sh'
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${TAG:-r06a}
libs="product"; for v in ${VARIANTS:-}; do libs="$libs snappier_amd/variants/libsnappier_hip_$v.so"; done
if [ -n "${VARIANTS:-}" ]; then
  MODE=decode DATA=${DATA:-html,low,mixed} REPS=${REPS:-6} timeout 600 python scripts/ab_libs.py $libs 2>&1 | grep '^{' > gpurun_out/${T}_ab_decode.jsonl; cat gpurun_out/${T}_ab_decode.jsonl | cut -c1-330
fi
if [ -z "${NO_BENCH:-}" ]; then
  timeout 900 python bench.py ${BENCH_ARGS:-} > gpurun_out/${T}_bench_line.json 2> gpurun_out/${T}_bench.err; tail -c 600 gpurun_out/${T}_bench_line.json; echo; tail -3 gpurun_out/${T}_bench.err
  rm -rf gpurun_out/${T}_prof
  (cd /tmp && BENCH_NO_PLAIN=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/${T}_prof -o bench -- python $OLDPWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-live-traffic --no-other-configs > $OLDPWD/gpurun_out/${T}_bench_line_under_rocprof.json 2> $OLDPWD/gpurun_out/${T}_rocprof.err)
  f=$(find gpurun_out/${T}_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${T}_bench_kernel_stats.csv; head -6 gpurun_out/${T}_bench_kernel_stats.csv | cut -c1-160
fi
