#!/usr/bin/env python3
"""Decode per corpus file for every library under snappier_amd/variants/ (same compressed input, interleaved).   python scripts/r5_ab_by_file.py [blocks]"""
import glob, json, os, subprocess, sys
nb = sys.argv[1] if len(sys.argv) > 1 else "32768"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lib in sorted(glob.glob(os.path.join(root, "snappier_amd", "variants", "libsnappier_hip_*.so"))):
    env = dict(os.environ, SNAPPIER_HIP_LIB=lib, MODES="default")
    env.pop("SNAPPIER_HIP_DECODE", None)
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "decode_by_file.py"), nb], env=env, capture_output=True, text=True, timeout=600)
    rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    print(json.dumps({"lib": os.path.basename(lib), "ms": {x["file"]: x["default"]["ms"] for x in rows}}), flush=True)
    if r.returncode != 0:
        print(r.stderr[-800:])
