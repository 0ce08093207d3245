#!/usr/bin/env python3
"""Turns the text scripts/pmc_passes.sh prints (per-kernel counter totals over a `bench.py --steps 1 --warmup 0` run, in which
every codec kernel is launched twice) into the JSON bench.py replays as `roofline.traffic`.
   python scripts/make_traffic_json.py gpurun_out/r2p_pmc_passes.txt profiles/r02p_hbm_traffic.json"""
import json, re, sys

src, dst = sys.argv[1], sys.argv[2]
launches, blocks = 2, 163840
acc = {}
for line in open(src):
    m = re.match(r"\s+(k_\w+)\s+(\w+)\s+([0-9.e+]+)", line)
    if m:
        acc.setdefault(m.group(1), {})[m.group(2)] = float(m.group(3))
out = {"note": "rocprofv3 --pmc passes (scripts/pmc_passes.sh; FETCH_SIZE and WRITE_SIZE in separate passes) over bench.py --steps 1 --warmup 0: "
               "one untimed setup pass + one step, so each kernel is launched twice and the totals are halved here. 163840 blocks of 64 KiB "
               "(10 GiB), MI355X. FETCH_SIZE / WRITE_SIZE are reported in KB (x1024). Calibration as in profiles/r01k_pmc_calibration.json: "
               "scattered narrow reads (the compressor's tables, the decompressor's back-references) are counted at 64 B per miss exactly.",
       "blocks": blocks, "kernels": {}, "instruction_mix_per_launch": {}}
for k, c in acc.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c and not k.endswith("_small"):
        f, w = c["FETCH_SIZE"] * 1024 / launches, c["WRITE_SIZE"] * 1024 / launches
        out["kernels"][k] = {"fetch_bytes": f, "write_bytes": w, "fetch_bytes_per_block": f / blocks, "write_bytes_per_block": w / blocks}
    mix = {n: v / launches for n, v in c.items() if n not in ("FETCH_SIZE", "WRITE_SIZE")}
    if mix and not k.endswith("_small"):
        out["instruction_mix_per_launch"][k] = mix
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
