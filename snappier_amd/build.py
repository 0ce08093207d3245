"""Builds the gfx950 native libraries in-tree with hipcc (cross-compiles without a GPU).

    python snappier_amd/build.py            # libsnappier_hip.so (+ libsnappier_datagen.so, bench/test helper)
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fconstexpr-steps=100000000", "-Wall", "-Wno-sometimes-uninitialized",
         "-Wno-unused-function", "-Wl,-rpath,/opt/rocm/lib"]

LIBS = {
    "libsnappier_hip.so": ["decode_chains.hip", "decompress.hip", "decompress_small.hip", "tag_index.hip", "compress_lanes.hip", "compress_win.hip", "crc32c.hip", "framing.hip", "frame_scan.hip",
                           "capi_ctx.hip", "capi_pool.hip", "capi_batch.hip", "capi_host.hip", "capi_frame.hip"],
    "libsnappier_datagen.so": ["datagen.hip"],
}


def _stale(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    inc = os.path.join(HERE, "..", "include")
    deps = sources + [os.path.join(d, f) for d in (CSRC, inc) for f in os.listdir(d) if f.endswith(".h")]   # every header either directory holds
    return any(os.path.getmtime(s) > t for s in deps)


def build_native(force: bool = False, verbose: bool = False) -> list[str]:
    """Compile every HIP library for gfx950; returns the paths of the built .so files."""
    built = []
    for lib, srcs in LIBS.items():
        target = os.path.join(HERE, lib)
        sources = [os.path.join(CSRC, s) for s in srcs]
        if force or _stale(target, sources):
            cmd = [HIPCC] + FLAGS + sources + ["-o", target]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
        built.append(target)
    return built


if __name__ == "__main__":
    for p in build_native(force="--force" in sys.argv, verbose=True):
        print(p)
