#!/bin/bash
# Round 3, GPU call 5: decoder experiments E1 (pieces up front), E2 (early write-out), E3 (aligned write-out) and their combinations.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=${TAG:-r03e}
for v in e123 e23; do
  SNAPPIER_HIP_LIB=$PWD/snappier_amd/variants/libsnappier_hip_$v.so timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/${T}_pytest_$v.txt
  echo "$v: $(tail -1 gpurun_out/${T}_pytest_$v.txt)"
done
OUT=gpurun_out/${T}_ab_decode.jsonl DATASETS="html mixed" AB_REPS="1 2" REPS=5 bash scripts/ab_variants.sh > /dev/null 2>&1
python - <<'PY'
import json, collections
r=collections.defaultdict(list)
for l in open("gpurun_out/r03e_ab_decode.jsonl"):
    try: d=json.loads(l)
    except Exception: print("BAD", l[:160]); continue
    r[(d["variant"], d["data"])].append((min(d["decompress_ms"]), d["roundtrip_ok"]))
for k in sorted(r): print(k, r[k])
PY
