#!/usr/bin/env python3
"""Turns the text scripts/pmc_passes.sh prints (per-kernel counter totals over a `bench.py --steps 1 --warmup 0` run, in which
every codec kernel is launched twice) into the JSON bench.py replays as `roofline.traffic`.
   python scripts/make_traffic_json.py gpurun_out/r03_pmc_passes.txt profiles/r03_hbm_traffic.json [commit]
The optional third argument is the commit the profiled kernels were built from; bench.py quotes it next to the replayed traffic so
that a stale replay is detectable (`roofline.traffic_source.kernels_commit`)."""
import json, re, sys

src, dst = sys.argv[1], sys.argv[2]
commit = sys.argv[3] if len(sys.argv) > 3 else "unrecorded"
blocks = 163840
# launches of each kernel in that run, from its wavefront count (the first decompress call of a context takes the pre-pass + list
# kernel, the second -- its context now knows the batch is 64 KiB blocks -- the plain one-workgroup-per-block kernel)
WAVES_PER_LAUNCH = {"k_compress_lanes": 2560, "k_decompress_chains": 163840, "k_decompress_chains_list": 8192, "k_decompress": 163840}
acc = {}
for line in open(src):
    m = re.match(r"\s+(k_\w+)\s+(\w+)\s+([0-9.e+]+)", line)
    if m:
        acc.setdefault(m.group(1), {})[m.group(2)] = float(m.group(3))
out = {"note": "rocprofv3 --pmc passes (scripts/pmc_passes.sh; FETCH_SIZE and WRITE_SIZE in separate passes) over bench.py --steps 1 --warmup 0: "
               "one untimed setup pass + one step; totals are divided by each kernel's launch count in that run (from SQ_WAVES). 163840 blocks of 64 KiB "
               "(10 GiB), MI355X. FETCH_SIZE / WRITE_SIZE are reported in KB (x1024). Calibration as in profiles/r01k_pmc_calibration.json: "
               "scattered narrow reads (the compressor's tables, the decompressor's back-references) are counted at 64 B per miss exactly.",
       "blocks": blocks, "source_commit": commit, "launches_per_kernel": {}, "kernels": {}, "instruction_mix_per_launch": {}}
for k, c in acc.items():
    launches = max(1, round(c.get("SQ_WAVES", 0) / WAVES_PER_LAUNCH[k])) if k in WAVES_PER_LAUNCH and c.get("SQ_WAVES") else 2
    out["launches_per_kernel"][k] = launches
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c and not k.endswith("_small"):
        f, w = c["FETCH_SIZE"] * 1024 / launches, c["WRITE_SIZE"] * 1024 / launches
        out["kernels"][k] = {"fetch_bytes": f, "write_bytes": w, "fetch_bytes_per_block": f / blocks, "write_bytes_per_block": w / blocks}
    mix = {n: v / launches for n, v in c.items() if n not in ("FETCH_SIZE", "WRITE_SIZE")}
    if mix and not k.endswith("_small"):
        out["instruction_mix_per_launch"][k] = mix
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
