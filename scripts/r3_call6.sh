#!/bin/bash
# Round 3, GPU call 6: code-generation flags on the decoder (scheduling strategy, early if-conversion), html-like + mixed.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=${TAG:-r03f}
OUT=gpurun_out/${T}_ab_decode.jsonl DATASETS="html" AB_REPS="1 2" REPS=5 bash scripts/ab_variants.sh > /dev/null 2>&1
python - <<'PY'
import json, collections
r=collections.defaultdict(list)
for l in open("gpurun_out/r03f_ab_decode.jsonl"):
    try: d=json.loads(l)
    except Exception: print("BAD", l[:160]); continue
    r[(d["variant"], d["data"])].append((min(d["decompress_ms"]), d["roundtrip_ok"]))
for k in sorted(r): print(k, r[k])
PY
