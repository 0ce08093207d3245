"""bench.py's cpu_baseline legs run on the GPU box only (N = 1, rank 0); this keeps them from rotting: the same function on a tiny sample here --
oracle, its -DORACLE_FAST build, C++ snappy when this host has one, each on 1 and N threads -- must return the documented fields, a value
equal to the harmonic combination of its best bit-exact legs, and round trips that are checked inside (they assert)."""
import importlib.util
import os

import numpy as np

from conftest import ROOT, read_testdata
import datagen


def _bench():
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_cpu_baseline_fields_and_value():
    import oracle as O
    B = _bench()
    so_before = O.pyoracle._SO
    try:
        raw = datagen.html_like_blocks(read_testdata("html"), 0, 96)
        c = B.cpu_baseline(raw, O.HASH_CRC32C)
    finally:
        O.pyoracle._SO = so_before                              # (cpu_baseline points the binding at its natively built objects)
        O.pyoracle._lib = None
    for k in ("value", "unit", "cores", "host_cpus", "cpu_model", "kind", "compress_GBps", "compress_leg", "decompress_GBps", "decompress_leg",
              "legs", "crc32c_GBps", "sample", "note"):
        assert k in c, k
    assert c["kind"] == "port" and c["value"] > 0
    assert abs(c["value"] - 1.0 / (1.0 / c["compress_GBps"] + 1.0 / c["decompress_GBps"])) < 2e-3
    assert c["compress_leg"].startswith("oracle")              # the compress leg must be one whose bytes are Snappier's
    assert set(c["legs"]) == {"oracle", "oracle_fast", "libsnappy"}
    for name in ("oracle", "oracle_fast"):
        if c["legs"][name]:
            assert "1_thread" in c["legs"][name]
            for leg in c["legs"][name].values():
                assert leg["compress_GBps"] > 0 and leg["decompress_GBps"] > 0 and leg["passes"] >= 1
    if isinstance(c["legs"]["libsnappy"], dict):
        assert "1_thread" in c["legs"]["libsnappy"]["legs"]
