#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-executes itself under
                                                             torch.distributed.run, one process per GPU, 127.0.0.1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --config 5 ...                          BASELINE.json configs[4] as the measured workload

Workload (config.workload): BASELINE.json configs[1] -- 10 GiB of 64 KiB html-like blocks per GPU (163 840 blocks:
the html fixture tiled at a per-block offset + ~1 % byte mutations, seed 0x5EED0001; weak scaling, rank r owns
blocks [r*163840, (r+1)*163840)).  One STEP = one pass of the hot path over the batch: block-compress every block
(snp_compress_batch), block-decompress every block back (snp_decompress_batch), then the final directory gather
(all_gather of per-block lengths + status over RCCL when N > 1).  Inputs are resident in HBM before the timed region.

value = uncompressed bytes that made the whole round trip, all GPUs, per second of wall time (max over ranks).
Both directions are also reported separately from HIP-event kernel times, each with its roofline fraction:
algorithmic bytes per launch = sum over blocks of (U_b + C_b)    /  average launch duration, against
the 8 TB/s HBM3E peak.

When N > 1 (or with --config5-lines) the SAME run then measures BASELINE.json configs[4] -- the mixed-corpus blocks,
block-sharded across the ranks -- and reports it as `config5_lines` (codec only / + directory gather / + compaction
and payload gather to rank 0): one driver command covers the 8-GPU workload the north star names.  Never `value`.

cpu_baseline (rank 0, N = 1): three CPU implementations on a bounded sample of the same blocks, each on 1 thread, min(cores, 64) threads
and all cores -- the C oracle (oracle/snappy_oracle.c, a port of the same algorithm; rebuilt on this host with -O3 -march=native when gcc
is here), its -DORACLE_FAST build (16-byte literal / self-copy moves, the scalar stand-in for CopyHelpers.cs:64-230) and C++ snappy through
dlopen when this host has one; `value` = the fastest BIT-EXACT round trip they offer (best oracle-build compress leg + best decompress leg).

`roofline.traffic` is MEASURED by the invocation: rank 0 (N = 1) spawns two child runs of this script under `rocprofv3 --pmc` (FETCH_SIZE, then
WRITE_SIZE; --no-live-traffic skips them and replays the committed profile, which is also the fallback when a pass fails).

Also in the line (round 4): `value_plain_workspace` (the same kernels through a second context whose hash-table workspace is ONE plain
allocation: what the placement search is worth), `workspace_search` (candidates, transient bytes and seconds of that search), `per_rank`
(min / max / mean / per-rank list of the kernel times, the search seconds and the directory gather: decomposes an N > 1 line).
"""
from __future__ import annotations

import argparse
import ctypes
import ctypes.util
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BLOCK = 65536
HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
CORPUS = ["alice29.txt", "asyoulik.txt", "fireworks.jpeg", "geo.protodata", "html", "html_x_4", "kppkn.gtb", "lcet10.txt",
          "paper-100k.pdf", "plrabn12.txt", "urls.10K"]       # SnappyTests.cs:8-19 corpus order


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def native_oracle(fast: bool = False):
    """The oracle rebuilt for THIS host (-O3 -march=native), in a temporary directory; falls back to the in-tree -O2 -msse4.2 build
    (which is what the parity tests use) when there is no compiler.  fast = -DORACLE_FAST: literal / self-copy paths of the
    decompressor move 16 bytes at a time (the scalar stand-in for CopyHelpers.cs:64-230's SSSE3 path; tests/test_oracle_fast.py
    holds it to the plain build's results) -- used by cpu_baseline only.  Returns (path or None, build description)."""
    src = os.path.join(ROOT, "oracle", "snappy_oracle.c")
    try:
        d = tempfile.mkdtemp(prefix="snp_oracle_")
        so = os.path.join(d, "libsnappy_oracle_native%s.so" % ("_fast" if fast else ""))
        subprocess.run(["gcc", "-O3", "-march=native", "-std=c11", "-fPIC", "-shared"] + (["-DORACLE_FAST"] if fast else []) + ["-o", so, src],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
        return so, "-O3 -march=native%s (built on this host)" % (" -DORACLE_FAST" if fast else "")
    except Exception:
        return None, "-O2 -msse4.2 (in-tree build; no compiler on this host)"


def time_libsnappy(sample: np.ndarray, nb: int, threads_list, budget_s: float):
    """Google's C++ snappy (what Snappier is a port of), if this host has one: compress / uncompress of the same blocks through its
    C API, on 1 thread and on each thread count of threads_list (blocks striped over a thread pool; ctypes releases the GIL during the
    calls).  Returns a dict or a string saying it is absent."""
    from concurrent.futures import ThreadPoolExecutor
    path = ctypes.util.find_library("snappy")
    L = None
    for cand in ([path] if path else []) + ["libsnappy.so.1", "/opt/conda/lib/libsnappy.so.1", "/usr/lib/x86_64-linux-gnu/libsnappy.so.1"]:
        try:
            L = ctypes.CDLL(cand)
            break
        except OSError:
            L = None
    if L is None:
        return "libsnappy: not found on this host (dlopen)"
    L.snappy_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    L.snappy_uncompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    L.snappy_max_compressed_length.restype = ctypes.c_size_t
    L.snappy_max_compressed_length.argtypes = [ctypes.c_size_t]
    cap = L.snappy_max_compressed_length(BLOCK)
    comp = np.empty(nb * cap, dtype=np.uint8)
    clen = np.zeros(nb, dtype=np.uint64)
    dlen = np.zeros(nb, dtype=np.uint64)
    back = np.empty(nb * BLOCK, dtype=np.uint8)
    raw_off = np.arange(nb, dtype=np.uint64) * np.uint64(BLOCK)
    raw_len = np.full(nb, BLOCK, dtype=np.uint64)
    comp_off = np.arange(nb, dtype=np.uint64) * np.uint64(cap)
    bad = []
    # the per-block loop runs in C (oracle/snappy_oracle.c: orc_foreign_codec_batch takes the function pointer), one call per thread
    import oracle as O
    OL = O.lib()
    OL.orc_foreign_codec_batch.restype = ctypes.c_uint64
    OL.orc_foreign_codec_batch.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
    f_comp = ctypes.cast(L.snappy_compress, ctypes.c_void_p)
    f_dec = ctypes.cast(L.snappy_uncompress, ctypes.c_void_p)

    def comp_range(r):
        if OL.orc_foreign_codec_batch(f_comp, sample.ctypes.data, raw_off.ctypes.data, raw_len.ctypes.data, r[0], r[1], comp.ctypes.data, comp_off.ctypes.data, cap, clen.ctypes.data):
            bad.append(r)

    def dec_range(r):
        if OL.orc_foreign_codec_batch(f_dec, comp.ctypes.data, comp_off.ctypes.data, clen.ctypes.data, r[0], r[1], back.ctypes.data, raw_off.ctypes.data, BLOCK, dlen.ctypes.data):
            bad.append(r)

    def leg(threads, blocks):
        """one untimed pass (every thread touches its stripe of the output buffers), then the best of two timed passes per direction"""
        blocks = min(blocks, nb)
        edges = np.linspace(0, blocks, threads + 1).astype(np.int64)
        ranges = [(int(edges[i]), int(edges[i + 1])) for i in range(threads) if edges[i + 1] > edges[i]]
        t_c, t_d = [], []
        with ThreadPoolExecutor(len(ranges)) as ex:
            for rep in range(3):
                t0 = time.perf_counter()
                list(ex.map(comp_range, ranges))
                t1 = time.perf_counter()
                list(ex.map(dec_range, ranges))
                t2 = time.perf_counter()
                if rep:
                    t_c.append(t1 - t0)
                    t_d.append(t2 - t1)
        u = float(blocks) * BLOCK
        return {"threads": threads, "blocks": blocks, "compress_GBps": round(u / min(t_c) / 1e9, 3), "decompress_GBps": round(u / min(t_d) / 1e9, 3),
                "round_trip_GBps": round(u / (min(t_c) + min(t_d)) / 1e9, 3)}

    legs = {"1_thread": leg(1, max(256, min(nb, int(budget_s * 0.08e9 / BLOCK))))}
    for t in threads_list:
        if t > 1:
            legs[f"{t}_threads"] = leg(t, nb)
    if bad or not np.array_equal(back[: legs["1_thread"]["blocks"] * BLOCK], sample[: legs["1_thread"]["blocks"] * BLOCK]):
        return "libsnappy: round trip failed"
    return {"library": str(getattr(L, "_name", "libsnappy")), "legs": legs,
            "note": "C++ snappy's COMPRESSED bytes differ from Snappier's crc32c-hash bytes (other hash): its compress leg is a speed reference, not a "
                    "parity one; its decompress leg is bit-exact by construction (a Snappy block has one decoding)"}


def cpu_baseline(raw_sample: np.ndarray, variant: int):
    """The CPU beside the GPU, on a bounded sample of the same blocks (~25-30 s in total): the C oracle ("port": the same algorithm,
    scalar copies), its -DORACLE_FAST build (16-byte literal / self-copy moves: what Snappier's CopyHelpers does with SSSE3), and C++
    snappy when this host has one -- each on 1 thread, min(cores, 64) threads and all cores.  `value` = the fastest BIT-EXACT round
    trip the host offers: the best compress leg whose bytes are Snappier's (oracle builds) + the best decompress leg of all."""
    import oracle as O
    nb = raw_sample.size // BLOCK
    ncpu = os.cpu_count() or 1
    threads = max(1, min(ncpu, 64))
    in_off = (np.arange(nb, dtype=np.uint64) * np.uint64(BLOCK)).astype(np.uint64)
    in_len = np.full(nb, BLOCK, dtype=np.uint32)

    def use(so):
        if so:                                                  # same binding, another object
            O.pyoracle._SO = so
            O.pyoracle._lib = None

    stride = O.max_compressed_length(BLOCK)
    cbuf = np.empty(nb * stride, dtype=np.uint8)                # reused by every leg: no allocation and no first-touch page faults in a timed pass
    dbuf = np.empty(raw_sample.size, dtype=np.uint8)

    def leg(nthreads: int, blocks: int, budget_s: float):
        """round trips of the first `blocks` blocks with `nthreads` threads: one untimed pass (every thread touches ITS stripe of the two output buffers: pages
        land on the thread's NUMA node), then up to 3 timed passes within the budget; the BEST pass of each direction counts (a baseline should not be
        understated by a noisy neighbour)"""
        blocks = min(blocks, nb)
        s = raw_sample[: blocks * BLOCK]
        out, out_off, out_len, status = O.compress_batch(s, in_off[:blocks], in_len[:blocks], variant, nthreads, out=cbuf)
        O.decompress_batch(out, out_off, out_len, in_off[:blocks], in_len[:blocks], s.size, nthreads, out=dbuf)
        reps, t_c, t_d = 0, [], []
        t_start = time.perf_counter()
        while True:
            t0 = time.perf_counter()
            out, out_off, out_len, status = O.compress_batch(s, in_off[:blocks], in_len[:blocks], variant, nthreads, out=cbuf)
            t1 = time.perf_counter()
            dec, dlen, dst = O.decompress_batch(out, out_off, out_len, in_off[:blocks], in_len[:blocks], s.size, nthreads, out=dbuf)
            t2 = time.perf_counter()
            t_c.append(t1 - t0)
            t_d.append(t2 - t1)
            reps += 1
            assert (status == 0).all() and (dst == 0).all()
            if reps >= 3 or (time.perf_counter() - t_start) > budget_s:
                break
        assert dec[: s.size].tobytes() == s.tobytes()
        u = float(blocks) * BLOCK
        return {"threads": nthreads, "blocks": blocks, "passes": reps, "compress_GBps": round(u / min(t_c) / 1e9, 3), "decompress_GBps": round(u / min(t_d) / 1e9, 3),
                "round_trip_GBps": round(u / (min(t_c) + min(t_d)) / 1e9, 3)}

    def legs_of(so, budget):
        use(so)
        O.compress_batch(raw_sample[: 64 * BLOCK], in_off[:64], in_len[:64], variant, threads)      # warm up / page in
        L = {f"{threads}_threads": leg(threads, nb, budget), "1_thread": leg(1, max(256, nb // 32), budget * 0.6)}
        if ncpu > threads:
            L[f"all_{ncpu}_cpus"] = leg(ncpu, nb, budget * 0.6)
        return L

    so_plain, how_plain = native_oracle(False)
    so_fast, how_fast = native_oracle(True)
    plain = legs_of(so_plain, 4.0)
    fast = legs_of(so_fast, 4.0) if so_fast else None
    # CRC-32C alone (the framing format's per-chunk checksum, Crc32CAlgorithm.cs:41-158): hardware crc32 instruction, 8 bytes per step
    crc = {}
    for t in (1, threads):
        t0 = time.perf_counter()
        crc_reps = 0
        while crc_reps < 3 and time.perf_counter() - t0 < 2.0:
            O.crc32c_batch(raw_sample, in_off, in_len, True, t)
            crc_reps += 1
        crc[f"{t}_thread" + ("s" if t > 1 else "")] = round(float(nb) * BLOCK * crc_reps / (time.perf_counter() - t0) / 1e9, 3)
    snappy = time_libsnappy(raw_sample, min(nb, 16384), [threads] + ([ncpu] if ncpu > threads else []), 4.0)
    # the fastest bit-exact round trip: compress by an oracle build (Snappier's bytes), decompress by whatever is fastest
    cands_c = [(v["compress_GBps"], f"oracle{tag} {k}") for tag, L in (("", plain), (" -DORACLE_FAST", fast)) if L for k, v in L.items()]
    cands_d = [(v["decompress_GBps"], f"oracle{tag} {k}") for tag, L in (("", plain), (" -DORACLE_FAST", fast)) if L for k, v in L.items()]
    if isinstance(snappy, dict):
        cands_d += [(v["decompress_GBps"], f"libsnappy {k}") for k, v in snappy["legs"].items()]
    best_c, best_d = max(cands_c), max(cands_d)
    value = round(1.0 / (1.0 / best_c[0] + 1.0 / best_d[0]), 3)
    used = max(int(w.split()[-1].split("_")[1 if w.split()[-1].startswith("all_") else 0]) for w in (best_c[1], best_d[1]))
    return {
        "value": value, "unit": "GB/s uncompressed, compress+decompress round trip",
        "cores": used, "host_cpus": ncpu, "cpu_model": cpu_model(), "kind": "port",
        "range_seen": "11-21 GB/s across the processes of rounds 4-6 on this host model (a shared 256-thread host: the same build moves by 2 x between processes); "
                      "the value is a per-leg maximum over thread counts, which flatters the CPU slightly -- a stated baseline, never the target",
        "compress_GBps": best_c[0], "compress_leg": best_c[1], "decompress_GBps": best_d[0], "decompress_leg": best_d[1],
        "legs": {"oracle": plain, "oracle_fast": fast, "libsnappy": snappy},
        "crc32c_GBps": crc,
        "sample": f"{nb} of the same html-like 64 KiB blocks (blocks striped over the threads); per leg one untimed pass that touches the reused output buffers, then up to 3 timed passes, the best of each direction counts; C oracle built {how_plain} and "
                  f"{how_fast} (hash = SSE4.2 crc32, as Snappier on x64/.NET 8+); value = the fastest bit-exact legs of the two directions combined",
        "note": "Snappier's C# path is not runnable on this host (no .NET runtime).  The oracle is a C port of the same algorithm; its -DORACLE_FAST build "
                "moves literals and self-copies 16 bytes at a time as Snappier's SIMD path does (CopyHelpers.cs:64-230), and C++ snappy's decoder is the "
                "code Snappier's is a port of: the best of these is the closest this host gets to Snappier's own speed",
    }


def sclk_reader(dev_index: int):
    """-> a function that reads this GPU's current shader clock in MHz from sysfs (hwmon freq1_input of the PCI device torch reports: one
    file read, ~50 us -- rocm-smi takes 80 ms, longer than the launches it would sample), or None when the host does not expose it."""
    import glob
    try:
        p = torch.cuda.get_device_properties(dev_index)
        base = f"/sys/bus/pci/devices/{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        files = glob.glob(os.path.join(base, "hwmon", "hwmon*", "freq1_input"))
        if not files:
            return None
        path = files[0]

        def level(name):                                        # the line pp_dpm_<name> marks with '*': the current level of that clock domain
            try:
                with open(os.path.join(base, name)) as f:
                    for line in f:
                        if "*" in line:
                            return int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
            except (OSError, ValueError, IndexError):
                pass
            return None

        def read():
            with open(path) as f:
                sclk = round(int(f.read().strip()) / 1e6)
            return {"sclk": sclk, "mclk": level("pp_dpm_mclk"), "fclk": level("pp_dpm_fclk")}
        read()
        return read
    except Exception:                                           # noqa: BLE001
        return None


def pmc_rows(directory: str, counter: str) -> dict:
    """{kernel: [bytes of `counter` per dispatch]} from the *counter_collection.csv files rocprofv3 --pmc left under `directory` (one row per
    dispatch and counter; FETCH_SIZE / WRITE_SIZE are in KB).  Only the codec kernels' full-size launches (>= 64 wavefronts)."""
    import csv
    import glob
    import re
    per = {}
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                m = re.search(r"(k_(?:compress|decompress|decode)[a-z_0-9]*)", r["Kernel_Name"])
                if m and r["Counter_Name"] == counter and int(r["Grid_Size"]) >= 64 * 64:      # (not the few-wavefront helpers)
                    per.setdefault(m.group(1), []).append(float(r["Counter_Value"]) * 1024.0)
    return per


def live_traffic(nb: int, hash_name: str, config: int):
    """HBM traffic of the codec kernels MEASURED on this box in this bench invocation: two child runs of this script (one untimed setup pass +
    one step each) under `rocprofv3 --pmc`, ONE counter per pass (FETCH_SIZE, then WRITE_SIZE: they do not fit one pass, MI355X_MICROARCH.md),
    no trace domains.  Returns {kernel: {"fetch_bytes", "write_bytes", "launches"}} per LAUNCH (mean over the launches of the child), or a string
    saying why it could not be measured (the line then falls back to the replayed profile and says so).  Counters are in KB; the calibration of
    profiles/r01k_pmc_calibration.json applies: scattered narrow accesses (both codec kernels) are counted exactly."""
    import shutil
    tool = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(tool):
        return "rocprofv3 not found on this host"
    acc = {}
    for counter, key in (("FETCH_SIZE", "fetch_bytes"), ("WRITE_SIZE", "write_bytes")):
        d = tempfile.mkdtemp(prefix="snp_pmc_")
        cmd = [tool, "--pmc", counter, "-d", d, "-o", "pmc", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__),
               "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-live-traffic", "--blocks", str(nb), "--hash", hash_name, "--config", str(config)]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        env.update(BENCH_NO_PLAIN="1", TMPDIR="/tmp")
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, timeout=90, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        except Exception as e:                                  # noqa: BLE001
            return f"rocprofv3 --pmc {counter} pass failed: {type(e).__name__}"
        per = pmc_rows(d, counter)
        shutil.rmtree(d, ignore_errors=True)
        if not per:
            return f"rocprofv3 --pmc {counter} pass produced no rows for the codec kernels"
        for k, v in per.items():
            acc.setdefault(k, {})[key] = sum(v) / len(v)
            acc[k]["launches"] = len(v)
    return {k: v for k, v in acc.items() if "fetch_bytes" in v and "write_bytes" in v}


def traffic_profile():
    """Per-block FETCH_SIZE + WRITE_SIZE of the committed PMC passes (rocprofv3 --pmc, one counter per pass, same workload).
    Replayed, never measured in the bench process: the object says which profile, from which commit, over how many launches."""
    for cand in ("r05_hbm_traffic.json", "r04_hbm_traffic.json", "r03_hbm_traffic.json", "r02p_hbm_traffic.json", "r02_hbm_traffic.json", "r01k_hbm_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", cand)) as f:
                doc = json.load(f)
            return doc["kernels"], {"file": f"profiles/{cand}", "kernels_commit": doc.get("source_commit", "unrecorded (profile predates round 3)"),
                                    "launches_profiled": doc.get("launches_per_kernel", "unrecorded")}
        except (OSError, KeyError, ValueError):
            continue
    return {}, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=163840, help="64 KiB blocks per GPU (163840 = 10 GiB)")
    ap.add_argument("--hash", choices=["crc32c", "mul"], default="crc32c")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-blocks", type=int, default=16384)
    ap.add_argument("--no-live-traffic", action="store_true", help="do not spawn the two rocprofv3 --pmc child runs that measure roofline.traffic (replay the committed profile instead)")
    ap.add_argument("--config", type=int, choices=[2, 5], default=2,
                    help="the MEASURED workload (`value`): 2 = BASELINE configs[1] (html-like blocks, the headline); 5 = configs[4] (mixed corpus)")
    ap.add_argument("--config5-lines", action="store_true",
                    help="after the measured workload, also run configs[4] (mixed corpus, block-sharded) and report config5_lines; "
                         "on by default when --gpus > 1")
    ap.add_argument("--thorough-search", action="store_true",
                    help="build the hash-table workspace with the thorough placement search (SNP_OPT_TABLE_PROBE_TRIES = 24, up to 3/4 of device memory, seconds) "
                         "instead of the library's bounded default; `value` then says so in config.workspace")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip `other_configs` (N = 1 only: BASELINE configs[2], configs[3] and the configs[4] share, 3 steps each, and the CRC kernel; ~15 s)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "RANK" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher -- one process per GPU over RCCL, rendezvous on 127.0.0.1
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        print(f"bench.py needs an MI355X: the codec has no CPU fallback (rank {rank} of {world})", file=sys.stderr, flush=True)
        if "RANK" in os.environ:
            time.sleep(1.0)             # (under a launcher the first failing rank gets the others killed: let every rank say it first)
        sys.exit(1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1 or "RANK" in os.environ             # under torch.distributed.run: always take the RCCL path
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)          # backend "nccl" IS RCCL on ROCm

    import snappier_amd as S
    from snappier_amd import batch as SB, datagen as SD, sharding

    variant = S.HASH_CRC32C if args.hash == "crc32c" else S.HASH_MUL
    nb = args.blocks
    free, _tot = torch.cuda.mem_get_info()
    need = nb * (2 * BLOCK + 76512) + (3 << 30)
    reduced_from = None
    if free < need:
        reduced_from = nb
        nb = int((free - (3 << 30)) // (2 * BLOCK + 76512)) // 1024 * 1024
        print(f"[bench] only {free >> 30} GiB free: reduced to {nb} blocks per GPU", file=sys.stderr)

    td = os.path.join(ROOT, "tests", "golden", "testdata")
    with open(os.path.join(td, "html"), "rb") as f:
        html = f.read()
    cd = SB.BlockCodec(local_rank, variant)
    thorough = args.thorough_search or bool(os.environ.get("BENCH_TABLE_TRIES"))
    if not os.environ.get("BENCH_NO_RESERVE"):
        # what a service does at start-up: the device's hash-table workspace is built before the buffers exist.  With the library's
        # DEFAULT options (round 6: this is what produces `value`): the bounded placement search -- at most two workspaces' worth of candidate pieces,
        # never more than half of free memory, < 1 s.  --thorough-search asks for the 24-workspace search instead (seconds, up to 3/4 of device memory:
        # worth 0-5 % depending on where the driver placed things).  Untimed either way; its cost is `workspace_search`.
        if thorough:
            from snappier_amd import _native as N0
            cd.ctx.set_option(N0.OPT_TABLE_PROBE_TRIES, int(os.environ.get("BENCH_TABLE_TRIES", "24")))
            cd.ctx.set_option(N0.OPT_TABLE_PROBE_MAX_BYTES, int(free // 4 * 3))        # (the third kind of device memory has been seen to begin beyond the first half)
        cd.ctx.reserve_compress(nb)

    def make_blocks(kind: int):
        if kind == 5:       # mixed corpus: block b takes corpus file b mod 11, rank r owns [r*nb, (r+1)*nb)
            def corpus_file(f):     # html_x_4 = html repeated four times
                return html * 4 if f == "html_x_4" else open(os.path.join(td, f), "rb").read()
            return SD.corpus_blocks([corpus_file(f) for f in CORPUS], rank * nb, nb, SD.MIXED_SEED, dev)
        return SD.html_like_blocks(html, rank * nb, nb, dev)

    raw = make_blocks(args.config)
    in_off, in_len = cd.uniform_layout(nb)
    comp = torch.empty(nb * cd.comp_stride, dtype=torch.uint8, device=dev)
    comp_off = torch.arange(nb, dtype=torch.int64, device=dev) * cd.comp_stride
    back = torch.empty_like(raw)
    torch.cuda.synchronize()

    ev = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731  (recorded on the stream the kernels run on)
    t_comp, t_dec = [], []

    def step(record: bool, settle: bool = False):
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        _o, _oo, out_len, status = cd.compress(raw, in_off, in_len, out=comp, out_off=comp_off)
        e1.record()
        if settle:                                          # setup pass only (never timed): the context's FIRST decode then finds its stream idle
            torch.cuda.synchronize()                        # and decides its layout from a sample of this batch, as a service's first
                                                            # request does -- it is not queued behind 100 ms of its own compress either
        dlen, dst = cd.decompress(comp, comp_off, out_len, back, in_off, in_len)
        e2.record()
        if distributed:                                     # the one exchange step: (length, status) directory
            sharding.gather_directory(out_len, status, nb * world)
        if record:
            t_comp.append((e0, e1))
            t_dec.append((e1, e2))
        return out_len, status, dlen, dst

    step(False, settle=True)        # setup pass (never timed): first-use allocations, workspace placement search
    for _ in range(args.warmup):
        step(False)

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, reps):
        """max over ranks of the wall time of `reps` calls of fn, bracketed by barrier + synchronize on both sides"""
        barrier()
        t0 = time.perf_counter()
        r = None
        for _ in range(reps):
            r = fn()
        barrier()
        el = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, r

    elapsed, (out_len, status, dlen, dst) = timed(lambda: step(True), args.steps)

    def verified(out_len, status, dlen, dst):
        return int((status != 0).sum()) == 0 and int((dst != 0).sum()) == 0 and bool((dlen == BLOCK).all()) and torch.equal(back, raw)

    # ---- verification (outside the timed region): every status OK, decode(encode(x)) == x -----------------------
    if not verified(out_len, status, dlen, dst):
        sys.exit("[bench] round trip is NOT bit-exact -- number invalid")
    c_bytes = float(out_len.to(torch.int64).sum().item())
    u_bytes = float(nb) * BLOCK
    ms_c = float(np.mean([a.elapsed_time(b) for a, b in t_comp]))
    ms_d = float(np.mean([a.elapsed_time(b) for a, b in t_dec]))
    cpu_sample = raw[: min(args.cpu_sample_blocks, nb) * BLOCK].cpu().numpy() if (world == 1 and rank == 0 and not args.no_cpu_baseline) else None

    # ---- the decoder ALONE and the shader clock around the launches: inside the step the decode launch follows a ~96 ms compress
    # launch that is bound by memory latency, and runs 5-9 % slower than in a loop of its own.  Decode-only launches back to back (events), and
    # the clock sampled from sysfs while each kind of launch runs and right after it (host-side sampling of a free-running clock: indicative).
    alone = None
    if world == 1:
        rd = sclk_reader(local_rank)
        def sample_during(fn, wait_s):
            """launch fn (asynchronous), read the clock wait_s later (the launch is still running), then right after it has finished"""
            torch.cuda.synchronize()
            fn()
            time.sleep(wait_s)
            during = rd() if rd else None
            torch.cuda.synchronize()
            after = rd() if rd else None
            return during, after
        a_ev = []
        torch.cuda.synchronize()
        for _ in range(6):
            e0, e1 = ev(), ev()
            e0.record()
            dl_a, ds_a = cd.decompress(comp, comp_off, out_len, back, in_off, in_len)
            e1.record()
            a_ev.append((e0, e1))
        torch.cuda.synchronize()
        a_ms = [a.elapsed_time(b) for a, b in a_ev]
        clk = {}
        if rd:
            clk["idle_before"] = rd()
            clk["during_compress"], clk["after_compress"] = sample_during(lambda: cd.compress(raw, in_off, in_len, out=comp, out_off=comp_off), ms_c * 0.6e-3)
            def comp_then_dec():
                _o, _oo, ol_, _s = cd.compress(raw, in_off, in_len, out=comp, out_off=comp_off)
                cd.decompress(comp, comp_off, ol_, back, in_off, in_len)
            clk["during_decode_in_step"], clk["after_decode_in_step"] = sample_during(comp_then_dec, (ms_c + ms_d * 0.5) * 1e-3)
            def dec4():
                for _ in range(4):
                    cd.decompress(comp, comp_off, out_len, back, in_off, in_len)
            clk["during_decode_alone"], clk["after_decode_alone"] = sample_during(dec4, ms_d * 3.5e-3)
        if not (int((ds_a != 0).sum()) == 0 and torch.equal(back, raw)):
            sys.exit("[bench] decode-alone round trip is NOT bit-exact")
        alone = {"decompress_ms_alone": round(float(np.mean(a_ms[1:])), 3), "launches_ms": [round(x, 3) for x in a_ms],
                 "how": "six decode-only launches back to back after the timed steps (HIP events on the launch stream; the first follows idle time and is left out of the mean)",
                 "clocks_mhz": clk or "not readable on this host (no hwmon freq1_input for the device)",
                 "clocks_how": "shader clock (sysfs hwmon freq1_input) and the current memory / fabric clock levels (pp_dpm_mclk, pp_dpm_fclk) of this GPU, read from the host (NOTE: targets, not the effective clock -- cycles / duration from PMC shows 2.13-2.15 GHz for the decode launch that follows a compress launch against 2.37-2.38 steady: profiles/r06p_decode_effective_clock.txt, DESIGN 7.1) -- while the launch runs (60 % into a compress launch; half way into the decode launch "
                             "that follows a compress launch; during the fourth of four back-to-back decode launches) and right after it finished"}

    # ---- what the placement search is worth: the same kernels on a PLAIN one-allocation workspace (second context, SNP_OPT_TABLE_PROBE_TRIES = 1) ----
    lanes = nb >= 32768                                        # (layout 0: the lane compressor from 32 768 fragments on)
    search = {"candidates": int(S.lib().snp_ctx_counter(cd.ctx.handle, 3)), "transient_bytes": int(S.lib().snp_ctx_counter(cd.ctx.handle, 5)),
              "seconds": round(S.lib().snp_ctx_counter(cd.ctx.handle, 4) / 1e6, 3), "chosen_set_probe_ms": S.lib().snp_ctx_counter(cd.ctx.handle, 2) / 1e3,
              "where": "snp_ctx_reserve_compress before the buffers exist (untimed start-up work)" if not os.environ.get("BENCH_NO_RESERVE") else "first compress call (untimed setup pass)"}
    plain = None
    if lanes and not os.environ.get("BENCH_NO_PLAIN"):
        from snappier_amd import _native as N
        cd2 = SB.BlockCodec(local_rank, variant)
        cd2.ctx.set_option(N.OPT_TABLE_PROBE_TRIES, 1)
        p_comp, p_dec = [], []
        def plain_step():
            e0, e1, e2 = ev(), ev(), ev()
            e0.record()
            _o, _oo, ol, stt = cd2.compress(raw, in_off, in_len, out=comp, out_off=comp_off)
            e1.record()
            dl, ds = cd2.decompress(comp, comp_off, ol, back, in_off, in_len)
            e2.record()
            p_comp.append((e0, e1))
            p_dec.append((e1, e2))
            return ol, stt, dl, ds
        plain_step()                                            # setup pass: its workspace allocation
        p_comp.clear(); p_dec.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            r = plain_step()
        torch.cuda.synchronize()
        p_el = (time.perf_counter() - t0) / 3
        if not verified(*r):
            sys.exit("[bench] plain-workspace round trip is NOT bit-exact")
        plain = {"value_GBps_this_gpu": round(u_bytes / p_el / 1e9, 3), "steps": 3,
                 "compress_ms": round(float(np.mean([a.elapsed_time(b) for a, b in p_comp])), 3),
                 "decompress_ms": round(float(np.mean([a.elapsed_time(b) for a, b in p_dec])), 3),
                 "workspace": "one hipMalloc of 64 KiB per fragment (SNP_OPT_TABLE_PROBE_TRIES = 1: no placement search), second context, same buffers"}
        del cd2
    # ---- what the library's DEFAULT options give: a fresh pool (every context of this process is gone), no option set, no reserve call -------
    default_search = None
    if lanes and world == 1 and thorough and not os.environ.get("BENCH_NO_DEFAULT_SEARCH"):
        import gc
        from snappier_amd import batch as SB2
        comp_stride = cd.comp_stride
        cd = None
        gc.collect()                                            # the device's table pool dies with its last context
        cd = SB2.BlockCodec(local_rank, variant)
        assert cd.comp_stride == comp_stride
        d_comp, d_dec = [], []
        def default_step():
            e0, e1, e2 = ev(), ev(), ev()
            e0.record()
            _o, _oo, ol, stt = cd.compress(raw, in_off, in_len, out=comp, out_off=comp_off)
            e1.record()
            dl, ds = cd.decompress(comp, comp_off, ol, back, in_off, in_len)
            e2.record()
            d_comp.append((e0, e1)); d_dec.append((e1, e2))
            return ol, stt, dl, ds
        t_s0 = time.perf_counter()
        default_step()                                          # the first large compress call of the process's new pool: pays the bounded search
        torch.cuda.synchronize()
        first_call_s = time.perf_counter() - t_s0
        d_comp.clear(); d_dec.clear()
        t0 = time.perf_counter()
        for _ in range(3):
            r = default_step()
        torch.cuda.synchronize()
        d_el = (time.perf_counter() - t0) / 3
        if not verified(*r):
            sys.exit("[bench] default-search round trip is NOT bit-exact")
        default_search = {"value_GBps_this_gpu": round(u_bytes / d_el / 1e9, 3), "steps": 3,
                          "compress_ms": round(float(np.mean([a.elapsed_time(b) for a, b in d_comp])), 3),
                          "decompress_ms": round(float(np.mean([a.elapsed_time(b) for a, b in d_dec])), 3),
                          "first_call_seconds": round(first_call_s, 3),
                          "search": {"candidates": int(cd.ctx.counter(3)), "transient_bytes": int(cd.ctx.counter(5)), "seconds": round(cd.ctx.counter(4) / 1e6, 3)},
                          "workspace": "library defaults: no option set, no snp_ctx_reserve_compress; the pool is built by the first compress call with this process's buffers already allocated"}
    # ---- per-rank decomposition (N >= 1): which rank was slow, and in which part; the directory gather alone ----
    t_gd = 0.0
    if distributed:
        barrier()
        g0 = time.perf_counter()
        for _ in range(3):
            sharding.gather_directory(out_len, status, nb * world)
        torch.cuda.synchronize()
        t_gd = (time.perf_counter() - g0) / 3 * 1e3
    per_rank = sharding.rank_stats({"compress_ms": ms_c, "decompress_ms": ms_d, "workspace_search_s": search["seconds"],
                                    "directory_gather_ms": t_gd, "plain_compress_ms": plain["compress_ms"] if plain else 0.0}, device=dev)
    if distributed:
        assert dist.get_world_size() == args.gpus, "rccl_ranks != n_gpus"

    # ---- configs[4]: the mixed corpus, block-sharded, and the lines around the codec (never part of `value`) ----------
    extra = {}
    if args.config == 5 or args.config5_lines or world > 1:
        if args.config != 5:
            del raw
            raw = make_blocks(5)                                # same buffers otherwise: comp, back are reused
            step(False)
        t5_comp, t5_dec = [], []
        def codec_only():
            e0, e1, e2 = ev(), ev(), ev()
            e0.record()
            _o, _oo, ol, stt = cd.compress(raw, in_off, in_len, out=comp, out_off=comp_off)
            e1.record()
            dl, ds = cd.decompress(comp, comp_off, ol, back, in_off, in_len)
            e2.record()
            t5_comp.append((e0, e1))
            t5_dec.append((e1, e2))
            return ol, stt, dl, ds
        def codec_and_directory():
            ol, stt, dl, ds = codec_only()
            sharding.gather_directory(ol, stt, nb * world)
            return ol, stt, dl, ds
        reps = max(1, min(args.steps, 3))
        t_codec, (ol5, st5, dl5, ds5) = timed(codec_only, reps)
        t_dir, _ = timed(codec_and_directory, reps)
        def compact_and_gather():       # what a real 8-GPU job ends with: compact this rank's blocks, directory, payload to rank 0
            stream, _dst = cd.compact(comp, comp_off, ol5)
            all_len, _all_st, _offs = sharding.gather_directory(ol5, st5, nb * world)
            return sharding.gather_payload(stream, all_len, nb * world, dst=0)
        t_gather, gathered = timed(compact_and_gather, reps)
        if not verified(ol5, st5, dl5, ds5):
            sys.exit("[bench] configs[4] round trip is NOT bit-exact")
        c5 = float(ol5.to(torch.int64).sum().item())
        extra = {"workload": f"configs[4]: {nb * world} mixed-corpus 64 KiB blocks ({nb * world * BLOCK / 2**30:.0f} GiB), block b = corpus file b mod 11, "
                             f"rank r owns blocks [r*{nb}, (r+1)*{nb})",
                 "n_gpus": world, "reps": reps,
                 "codec_only_GBps": round(float(nb) * BLOCK * world / (t_codec / reps) / 1e9, 2),
                 "codec_plus_directory_GBps": round(float(nb) * BLOCK * world / (t_dir / reps) / 1e9, 2),
                 "compact_plus_payload_gather_ms": round(t_gather / reps * 1e3, 2),
                 "payload_gather_GBps_compressed": (round(float(gathered.numel()) / (t_gather / reps) / 1e9, 2)
                                                    if gathered is not None and t_gather > 0 else None),
                 "rank0_compress_ms": round(float(np.mean([a.elapsed_time(b) for a, b in t5_comp])), 3),
                 "rank0_decompress_ms": round(float(np.mean([a.elapsed_time(b) for a, b in t5_dec])), 3),
                 "rank0_compression_ratio": round(c5 / u_bytes, 4),
                 "verified": "decode(encode(x)) == x for every block of every rank's shard" if True else None}

    # ---- other_configs (N = 1): BASELINE configs[2] (low entropy), configs[3] (framing) and the configs[4] share (mixed corpus) on the same buffers,
    # 3 steps each, verified; and the CRC kernel alone.  Never part of `value`.
    other = None
    if world == 1 and not args.no_other_configs and not extra:
        other = []
        def block_config(name, data):
            tc, tdd = [], []
            def one():
                e0, e1, e2 = ev(), ev(), ev()
                e0.record()
                _o, _oo, ol, stt = cd.compress(data, in_off, in_len, out=comp, out_off=comp_off)
                e1.record()
                dl, ds = cd.decompress(comp, comp_off, ol, back, in_off, in_len)
                e2.record()
                tc.append((e0, e1)); tdd.append((e1, e2))
                return ol, stt, dl, ds
            one()
            tc.clear(); tdd.clear()
            for _ in range(3):
                ol, stt, dl, ds = one()
            torch.cuda.synchronize()
            ok = int((stt != 0).sum()) == 0 and int((ds != 0).sum()) == 0 and bool((dl == BLOCK).all()) and torch.equal(back, data)
            cb = float(ol.to(torch.int64).sum().item())
            mc = float(np.mean([a.elapsed_time(b) for a, b in tc])); md = float(np.mean([a.elapsed_time(b) for a, b in tdd]))
            other.append({"workload": name, "blocks": nb, "steps": 3, "compression_ratio": round(cb / u_bytes, 4), "verified": ok,
                          "compress_GBps": round(u_bytes / mc / 1e6, 2), "decompress_GBps": round(u_bytes / md / 1e6, 2),
                          "compress_ms": round(mc, 3), "decompress_ms": round(md, 3),
                          "roofline_compress_frac": round((u_bytes + cb) / mc / 1e6 / HBM_PEAK_GBPS, 5),
                          "roofline_decompress_frac": round((u_bytes + cb) / md / 1e6 / HBM_PEAK_GBPS, 5)})
        del raw
        raw = SD.low_entropy_blocks(0, nb, dev)
        block_config(f"configs[2]: {nb} low-entropy (~90 % match) 64 KiB blocks", raw)
        del raw
        raw = make_blocks(5)
        block_config(f"configs[4], one GPU's share: {nb} mixed-corpus 64 KiB blocks (block b = corpus file b mod 11)", raw)
        del raw
        raw = make_blocks(2)
        # the CRC kernel alone (framing's masked CRC-32C of every 64 KiB chunk)
        tcr = []
        cd.crc32c(raw, in_off, in_len, masked=True)
        for _ in range(3):
            e0, e1 = ev(), ev()
            e0.record(); crc = cd.crc32c(raw, in_off, in_len, masked=True); e1.record()
            tcr.append((e0, e1))
        torch.cuda.synchronize()
        ms_crc = float(np.mean([a.elapsed_time(b) for a, b in tcr]))
        other.append({"workload": f"masked CRC-32C of {nb} 64 KiB chunks (k_crc32c)", "steps": 3, "ms": round(ms_crc, 3), "GBps": round(u_bytes / ms_crc / 1e6, 1),
                      "roofline_frac": round(u_bytes / ms_crc / 1e6 / HBM_PEAK_GBPS, 4)})
        # configs[3]: the framing format end to end on the device (compress + CRC + raw-vs-compressed + headers; then header walk + decode + CRC verify)
        from snappier_amd import _native as N2
        f_out = torch.empty(N2.lib().snp_frame_max_encoded_length(raw.numel()), dtype=torch.uint8, device=dev)
        f_work = torch.empty(N2.lib().snp_frame_encode_workspace(raw.numel()), dtype=torch.uint8, device=dev)
        def t3(fn):
            fn()
            ts = []
            for _ in range(3):
                e0, e1 = ev(), ev()
                e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            return float(np.mean(ts)), r
        ms_e, (framed, written) = t3(lambda: cd.frame_encode(raw, f_out, f_work))
        w = int(written.item())
        back.zero_()
        d_work = torch.empty(N2.lib().snp_frame_decode_workspace(nb + 16), dtype=torch.uint8, device=dev)
        ms_w, res = t3(lambda: cd.frame_decode(framed, w, back, nb + 16, d_work))
        ok4 = res.cpu().tolist() == [int(u_bytes), 0] and torch.equal(back, raw)
        other.append({"workload": f"configs[3]: SnappyStream framing over {nb} html-like 64 KiB chunks, device resident (encode; header walk + decode + CRC verify)",
                      "steps": 3, "framed_bytes": w, "verified_crc_and_bytes": ok4, "frame_encode_GBps": round(u_bytes / ms_e / 1e6, 2),
                      "frame_decode_verify_GBps": round(u_bytes / ms_w / 1e6, 2), "frame_encode_ms": round(ms_e, 3), "frame_decode_ms": round(ms_w, 3)})
        del f_out, f_work, d_work
    if rank == 0:
        total_u = u_bytes * world
        ms_per_step = elapsed / args.steps * 1e3
        alg = u_bytes + c_bytes                                     # U + C for compress, C + U for decompress
        pmc, pmc_src = traffic_profile()
        live = None
        if world == 1 and not args.no_live_traffic and not os.environ.get("BENCH_NO_LIVE_TRAFFIC"):
            live = live_traffic(nb, args.hash, args.config)      # two child runs under rocprofv3 --pmc: ~15 s each
        def roof(ms, kernel):
            a = alg / (ms * 1e-3) / 1e9
            t = pmc.get(kernel)
            if isinstance(live, dict) and kernel in live:
                traffic = int(live[kernel]["fetch_bytes"] + live[kernel]["write_bytes"])
                source = {"how": "MEASURED on this box by this invocation: two child runs of bench.py (--steps 1 --warmup 0) under rocprofv3 --pmc, FETCH_SIZE and "
                                 "WRITE_SIZE in separate passes, mean per launch of this kernel", "launches_profiled": live[kernel]["launches"],
                          "fetch_bytes": int(live[kernel]["fetch_bytes"]), "write_bytes": int(live[kernel]["write_bytes"]),
                          "calibration": "profiles/r01k_pmc_calibration.json: scattered narrow accesses are counted exactly (raw counter x 1024)"}
            else:
                traffic = int((t["fetch_bytes_per_block"] + t["write_bytes_per_block"]) * nb) if t and args.hash == "crc32c" and args.config == 2 else None
                source = (dict(pmc_src, how="replayed: per-block FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes over this workload x blocks of this run; "
                                            "not measured in this process" + (f" (live measurement: {live})" if isinstance(live, str) else ""),
                               launches_this_run=args.steps) if traffic and pmc_src else None)
            return {"bound": "hbm", "kernel": kernel, "achieved": round(a, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(a / HBM_PEAK_GBPS, 5), "traffic": traffic,
                    "traffic_source": source,
                    "avg_launch_ms": round(ms, 4),
                    "algorithmic_bytes_per_launch": int(alg),
                    "uncompressed_GBps": round(u_bytes / (ms * 1e-3) / 1e9, 2)}
        r_c = roof(ms_c, "k_compress_lanes" if lanes else "k_compress_win")
        r_d = roof(ms_d, "k_decode_chains")
        if alone:
            r_d.update(alone)
            r_d["frac_alone"] = round(alg / (alone["decompress_ms_alone"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)
        if lanes:
            r_c["table_workspace_probe"] = {"chosen_ms": search["chosen_set_probe_ms"], "candidates": search["candidates"]}
        if lanes and args.config == 2:
            # What bounds the lane compressor is not U + C but the random accesses of its hash tables: the reference
            # parse makes 9928 probes + 4228 post-copy inserts per fragment of this workload (counted on the CPU with the oracle), each
            # into a uniformly random bucket of a 10.7 GB workspace.  Rates: scripts/microbench_random_table.hip on this GPU model
            # (profiles/r01d_microbench_random_table.jsonl: 20.26 G dependent load + store probes/s, 23.57 G write-only inserts/s).
            # floor = that table traffic alone as load + store.  Round 3 issues a probe as ONE atomic exchange, which the floor does
            # not model: frac_of_floor > 1 means the kernel now beats the two-request form of its own traffic.
            floor_ms = nb * (9928 / 20.26e9 + 4228 / 23.57e9) * 1e3
            r_c["random_access_floor"] = {"table_probes_per_fragment": 9928, "table_inserts_per_fragment": 4228,
                                          "floor_ms": round(floor_ms, 1), "frac_of_floor": round(floor_ms / ms_c, 3),
                                          "floor_models": "probe = dependent load + store (the round-1/2 kernel); round 3 probes with one atomic exchange",
                                          # the exchange probes alone at the best rate the bare table walk reaches on a workspace spread evenly over
                                          # three kinds of device memory (30.3 ms per 163840 x 4096 exchanges, profiles/r03y_microbench_memory_kinds.jsonl)
                                          "exchange_probes_alone_ms": round(nb * 9928 / (163840 * 4096 / 30.3e-3) * 1e3, 1)}
        line = {
            "metric": "uncompressed GB/s block compress+decompress, 64 KiB blocks",
            "value": round(total_u / (elapsed / args.steps) / 1e9, 3),
            "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": (f"configs[1]: {nb} html-like 64 KiB blocks ({nb * BLOCK / 2**30:.2f} GiB) per GPU, " if args.config == 2 else
                                    f"configs[4]: {nb} mixed-corpus 64 KiB blocks ({nb * BLOCK / 2**30:.2f} GiB) per GPU, ") +
                                   (f"REDUCED from {reduced_from} blocks (device memory short), " if reduced_from else "") +
                                   "step = compress all + decompress all (+ RCCL length/status gather when N > 1)",
                       "blocks_per_gpu": nb, "block_bytes": BLOCK, "hash_variant": args.hash,
                       "layout": "decompress: one block per wavefront (k_decode_chains: sub-chain tag parse over 2 KiB super-windows through an LDS table of tag advances, 64 tags per execution batch staged in LDS); compress: one fragment per lane, hash tables in an HBM workspace of 16 pieces spread over the kinds of device memory, probe + insert as one atomic exchange (>= 32768 fragments), else one per wavefront: table in LDS, from 4096 fragments joined by a second population with cache-resident global table slots",
                       "workspace": ("hash-table workspace built by the untimed setup pass" if os.environ.get("BENCH_NO_RESERVE") else
                                     "the device's hash-table workspace built by snp_ctx_reserve_compress before the buffers are allocated (a service's start-up: untimed, like the setup pass; its cost is workspace_search), " +
                                     ("with the THOROUGH placement search asked for explicitly (--thorough-search: SNP_OPT_TABLE_PROBE_TRIES = 24; what the default bounded search gives is value_default_search)" if thorough else
                                      "LIBRARY DEFAULT options: the bounded placement search (<= two workspaces' worth of candidates, <= half of free memory)")),
                       "value_range_seen": "placement of the 10.7 GB of hash tables decides +-8 % of the compressor's rate (random read-modify-writes: where hipMalloc landed, DESIGN 4.3/7.3): round 5 boxes 86.9-101.6 GB/s (plain 81.0-90.1, default search 91.9-101.2, thorough 96.0-101.6); round 6, default search on four boxes: 94.6 / 96.3 / 96.6 / 101.0 (plain 80.3-84.2, thorough 100.8)",
                       "rccl_ranks": dist.get_world_size() if distributed else 1,
                       "compression_ratio": round(c_bytes / u_bytes, 4), "parallelism": f"block-sharded x{world}, no data-path collective"},
            "compress_GBps": round(u_bytes * world / (ms_c * 1e-3) / 1e9, 2) if world == 1 else None,
            "decompress_GBps": round(u_bytes * world / (ms_d * 1e-3) / 1e9, 2) if world == 1 else None,
            "roofline": r_c if ms_c >= ms_d else r_d,               # the dominant kernel
            "roofline_compress": r_c, "roofline_decompress": r_d,
            "verified": "decode(encode(x)) == x for every block, all status OK",
            "workspace_search": search if lanes else None,
            "value_plain_workspace": plain,
            "value_default_search": default_search,
            "per_rank": per_rank,
        }
        if extra:
            line["config5_lines"] = extra
        if other:
            line["other_configs"] = other
        if cpu_sample is not None:
            line["cpu_baseline"] = cpu_baseline(cpu_sample, variant)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
