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


def test_pmc_rows_reads_rocprofv3_counter_csv(tmp_path):
    """bench.py's live traffic measurement parses what `rocprofv3 --pmc` writes: one row per dispatch and counter, values in KB; helper kernels
    (a few wavefronts) and other counters are left out, the list kernel is its own key."""
    B = _bench()
    d = tmp_path / "run" / "host" / "123"
    d.mkdir(parents=True)
    head = '"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name","Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp"\n'
    def row(i, grid, name, counter, value):
        return f'{i},{i},"Agent 2",1,1,1,{grid},7,"{name}",64,0,0,50,0,64,"{counter}",{value},1,2\n'
    lanes = "void (anonymous namespace)::k_compress_lanes<0, 1u, false>(unsigned char const*, unsigned int)"
    chains = "void (anonymous namespace)::k_decompress_chains<true>(unsigned char const*)"
    clist = "void (anonymous namespace)::k_decompress_chains_list<true>(unsigned char const*)"
    (d / "pmc_counter_collection.csv").write_text(head + row(1, 163840, lanes, "FETCH_SIZE", 1000.0) + row(2, 163840, lanes, "FETCH_SIZE", 3000.0) +
                                                  row(3, 10485760, chains, "FETCH_SIZE", 500.0) + row(4, 524288, clist, "FETCH_SIZE", 600.0) +
                                                  row(5, 64, "k_max_len(unsigned int const*)", "FETCH_SIZE", 9.0) + row(6, 163840, lanes, "WRITE_SIZE", 7.0) +
                                                  row(7, 1024, "(anonymous namespace)::k_compress_lanes_helper()", "FETCH_SIZE", 1.0))
    per = B.pmc_rows(str(tmp_path / "run"), "FETCH_SIZE")
    assert per == {"k_compress_lanes": [1000.0 * 1024, 3000.0 * 1024], "k_decompress_chains": [500.0 * 1024], "k_decompress_chains_list": [600.0 * 1024]}
    assert B.pmc_rows(str(tmp_path / "run"), "WRITE_SIZE") == {"k_compress_lanes": [7.0 * 1024]}
