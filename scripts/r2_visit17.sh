#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/r2v17_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2v17_prof -o hostapi -- python scripts/host_api_rates.py 1073741824 > gpurun_out/r2v17.log 2>&1
grep bytes gpurun_out/r2v17.log
f=$(find gpurun_out/r2v17_prof -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('k_tag','k_fragment','k_decompress','k_compress','k_gather','k_crc','k_span','k_frame')):
        print(n.split('(')[0][-40:], r['Calls'], round(float(r['AverageNs'])/1e6,3),'ms')
PY
