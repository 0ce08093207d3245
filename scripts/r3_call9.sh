#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/r03u_ab_decode.jsonl
OUT=gpurun_out/r03u_ab_decode.jsonl DATASETS="html mixed" AB_REPS="1 2 3" REPS=5 bash scripts/ab_variants.sh > /dev/null 2>&1
python - <<'PY'
import json, collections
r=collections.defaultdict(list)
for l in open("gpurun_out/r03u_ab_decode.jsonl"):
    try: d=json.loads(l)
    except Exception: print("BAD", l[:160]); continue
    r[(d["variant"], d["data"])].append((min(d["decompress_ms"]), d["roundtrip_ok"]))
for k in sorted(r): print(k, r[k])
PY
