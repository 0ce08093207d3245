#!/bin/bash
# Kernel durations of ONE large block through the host API (k_tag_cand / k_tag_scan / k_tag_index / k_decode_chains_frag), rocprofv3 --kernel-trace --stats.
#   gpurun -- 'bash scripts/time_tag_index.sh [bytes]'      SNAPPIER_HIP_LIB selects a variant
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$PWD
rm -rf /tmp/ti_prof; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ti_prof -o x -- python $R/scripts/host_api_rates.py ${1:-1073741824} > /tmp/ti_prof.log 2>&1); grep bytes /tmp/ti_prof.log | cut -c1-250
f=$(find /tmp/ti_prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('k_tag_', 'k_decode_chains_frag', 'k_fragment_starts')):
        print(r['Name'].split('(')[0][-40:], 'calls', r['Calls'], 'avg_ms', round(float(r['AverageNs']) / 1e6, 3))
PY
