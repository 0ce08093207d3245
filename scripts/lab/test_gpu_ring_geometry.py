"""The ring decoder (decompress.hip, FRONT = 4) in a geometry that makes its ordering rules bite: a 1 KiB ring with 512-byte batches.  In round 4
that build failed its round trip (profiles/r04o_ring_sizes_before_fix.txt): a batch decoded AHEAD requested far pieces that had not left the
ring yet, and a batch decoded for itself read the previous batch's bytes before they were written out -- the shipped 2 KiB geometry hides both
most of the time.  Builds the variant with hipcc (scripts/build_variant.sh) and decodes html-like, low-entropy and mixed blocks through it in a
child process (the library is chosen at import: SNAPPIER_HIP_LIB)."""
import json
import os
import shutil
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_ring_decoder_with_a_one_kib_ring_round_trips(tmp_path):
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this host: the variant cannot be built")
    env = dict(os.environ, OBJ_CACHE=str(tmp_path / "obj"), LAB="1", SRC="decompress.hip")   # the ring is a lab front end (csrc/lab/decompress_r04.hip)
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "build_variant.sh"), "test_ring1k", "-DSNP_D_RING=1024", "-DSNP_D_RING_SPAN=512"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = os.path.join(ROOT, "snappier_amd", "variants", "libsnappier_hip_test_ring1k.so")
    try:
        env = dict(os.environ, SNAPPIER_HIP_LIB=lib, SNAPPIER_HIP_DECODE="ring")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "ring_first_contact.py"), "4096"], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        assert [row["data"] for row in rows] == ["html", "low", "mixed"]
        for row in rows:
            assert row["decode"] == "ring" and row["bad_blocks"] == 0 and row["bad_status"] == 0, row
    finally:
        if os.path.exists(lib):
            os.remove(lib)
