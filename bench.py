#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-executes itself under
                                                             torch.distributed.run, one process per GPU, 127.0.0.1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --config 5 ...                          BASELINE.json configs[4]: mixed-corpus blocks, block-sharded

Workload (config.workload): BASELINE.json configs[1] -- 10 GiB of 64 KiB html-like blocks per GPU (163 840 blocks:
the html fixture tiled at a per-block offset + ~1 % byte mutations, seed 0x5EED0001; weak scaling, rank r owns
blocks [r*163840, (r+1)*163840)).  One STEP = one pass of the hot path over the batch: block-compress every block
(snp_compress_batch), block-decompress every block back (snp_decompress_batch), then the final directory gather
(all_gather of per-block lengths + status over RCCL when N > 1).  Inputs are resident in HBM before the timed region.

value = uncompressed bytes that made the whole round trip, all GPUs, per second of wall time (max over ranks).
Both directions are also reported separately from HIP-event kernel times, each with its roofline fraction:
algorithmic bytes per launch = sum over blocks of (U_b + C_b)    /  average launch duration, against
the 8 TB/s HBM3E peak.  cpu_baseline = the C oracle (oracle/snappy_oracle.c, a port of the same algorithm) timed on
this host's cores over a bounded sample of the same blocks.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BLOCK = 65536
HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(raw_sample: np.ndarray, variant: int):
    """Oracle ("port") on the host cores over a bounded sample; returns the cpu_baseline object."""
    import oracle as O
    nb = raw_sample.size // BLOCK
    threads = max(1, min(os.cpu_count() or 1, 64))
    in_off = (np.arange(nb, dtype=np.uint64) * np.uint64(BLOCK)).astype(np.uint64)
    in_len = np.full(nb, BLOCK, dtype=np.uint32)
    O.compress_batch(raw_sample[: 64 * BLOCK], in_off[:64], in_len[:64], variant, threads)      # warm up / page in
    reps, t_c, t_d = 0, 0.0, 0.0
    t_start = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        out, out_off, out_len, status = O.compress_batch(raw_sample, in_off, in_len, variant, threads)
        t1 = time.perf_counter()
        dec, dlen, dst = O.decompress_batch(out, out_off, out_len, in_off, in_len, raw_sample.size, threads)
        t2 = time.perf_counter()
        t_c += t1 - t0
        t_d += t2 - t1
        reps += 1
        assert (status == 0).all() and (dst == 0).all()
        if reps >= 3 or (time.perf_counter() - t_start) > 20.0:
            break
    assert dec.tobytes() == raw_sample.tobytes()
    u = float(nb) * BLOCK * reps
    return {
        "value": round(u / (t_c + t_d) / 1e9, 3), "unit": "GB/s uncompressed, compress+decompress round trip",
        "cores": threads, "host_cpus": os.cpu_count(), "kind": "port",
        "compress_GBps": round(u / t_c / 1e9, 3), "decompress_GBps": round(u / t_d / 1e9, 3),
        "sample": f"{nb} of the same html-like 64 KiB blocks x {reps} passes, {threads} threads (blocks striped), "
                  f"C oracle built -O2 -msse4.2 (hash = SSE4.2 crc32, as Snappier on x64/.NET 8+)",
        "note": "Snappier's C# path is not runnable on this host (no .NET runtime); the oracle is a C port of the same algorithm",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=163840, help="64 KiB blocks per GPU (163840 = 10 GiB)")
    ap.add_argument("--hash", choices=["crc32c", "mul"], default="crc32c")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-blocks", type=int, default=16384)
    ap.add_argument("--config", type=int, choices=[2, 5], default=2,
                    help="2 = BASELINE configs[1] (html-like blocks, the headline); 5 = configs[4] (mixed-corpus blocks, "
                         "block-sharded, plus the compaction and payload-gather lines)")
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
        sys.exit(f"bench.py needs an MI355X: the codec has no CPU fallback (rank {rank} of {world})")
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
    if free < need:
        nb = int((free - (3 << 30)) // (2 * BLOCK + 76512)) // 1024 * 1024
        print(f"[bench] only {free >> 30} GiB free: reduced to {nb} blocks per GPU", file=sys.stderr)

    with open(os.path.join(ROOT, "tests", "golden", "testdata", "html"), "rb") as f:
        html = f.read()
    cd = SB.BlockCodec(local_rank, variant)
    if args.config == 5:        # mixed corpus: block b takes corpus file b mod 11, rank r owns [r*nb, (r+1)*nb)
        td = os.path.join(ROOT, "tests", "golden", "testdata")
        names = ["alice29.txt", "asyoulik.txt", "fireworks.jpeg", "geo.protodata", "html", "html_x_4", "kppkn.gtb", "lcet10.txt",
                 "paper-100k.pdf", "plrabn12.txt", "urls.10K"]
        def corpus_file(f):     # html_x_4 = html repeated four times (SnappyTests.cs:8-19 corpus order)
            return html * 4 if f == "html_x_4" else open(os.path.join(td, f), "rb").read()
        raw = SD.corpus_blocks([corpus_file(f) for f in names], rank * nb, nb, SD.MIXED_SEED, dev)
    else:
        raw = SD.html_like_blocks(html, rank * nb, nb, dev)
    in_off, in_len = cd.uniform_layout(nb)
    comp = torch.empty(nb * cd.comp_stride, dtype=torch.uint8, device=dev)
    comp_off = torch.arange(nb, dtype=torch.int64, device=dev) * cd.comp_stride
    back = torch.empty_like(raw)
    torch.cuda.synchronize()

    ev = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731  (recorded on the stream the kernels run on)
    t_comp, t_dec = [], []

    def step(record: bool):
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        _o, _oo, out_len, status = cd.compress(raw, in_off, in_len, out=comp, out_off=comp_off)
        e1.record()
        dlen, dst = cd.decompress(comp, comp_off, out_len, back, in_off, in_len)
        e2.record()
        if distributed:                                     # the one exchange step: (length, status) directory
            sharding.gather_directory(out_len, status, nb * world)
        if record:
            t_comp.append((e0, e1))
            t_dec.append((e1, e2))
        return out_len, status, dlen, dst

    step(False)                     # setup pass (never timed): first-use allocations, workspace placement search
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

    # ---- config 5: the lines around the codec (never part of `value`) -------------------------------------------------
    extra = {}
    if args.config == 5:
        def codec_only():
            _o, _oo, ol, stt = cd.compress(raw, in_off, in_len, out=comp, out_off=comp_off)
            cd.decompress(comp, comp_off, ol, back, in_off, in_len)
            return ol, stt
        def compact_and_gather():       # what a real 8-GPU job ends with: compact this rank's blocks, directory, payload to rank 0
            ol, stt = out_len, status
            stream, _dst = cd.compact(comp, comp_off, ol)
            all_len, _all_st, _offs = sharding.gather_directory(ol, stt, nb * world)
            got = sharding.gather_payload(stream, all_len, nb * world, dst=0)
            return got
        reps = max(1, min(args.steps, 3))
        t_codec, _ = timed(codec_only, reps)
        t_gather, gathered = timed(compact_and_gather, reps)
        extra = {"codec_only_GBps": round(float(nb) * BLOCK * world / (t_codec / reps) / 1e9, 2),
                 "codec_plus_directory_GBps": round(float(nb) * BLOCK * world / (elapsed / args.steps) / 1e9, 2),
                 "compact_plus_payload_gather_ms": round(t_gather / reps * 1e3, 2),
                 "payload_gather_GBps_compressed": (round(float(gathered.numel()) / (t_gather / reps) / 1e9, 2)
                                                    if gathered is not None and t_gather > 0 else None)}

    # ---- verification (outside the timed region): every status OK, decode(encode(x)) == x -----------------------
    ok = int((status != 0).sum()) == 0 and int((dst != 0).sum()) == 0 and bool((dlen == BLOCK).all()) and torch.equal(back, raw)
    if not ok:
        sys.exit("[bench] round trip is NOT bit-exact -- number invalid")
    c_bytes = float(out_len.to(torch.int64).sum().item())
    u_bytes = float(nb) * BLOCK
    ms_c = float(np.mean([a.elapsed_time(b) for a, b in t_comp]))
    ms_d = float(np.mean([a.elapsed_time(b) for a, b in t_dec]))

    if rank == 0:
        total_u = u_bytes * world
        ms_per_step = elapsed / args.steps * 1e3
        alg = u_bytes + c_bytes                                     # U + C for compress, C + U for decompress
        # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per
        # pass, same workload): per-block figures x blocks of this run.  See profiles/r01k_hbm_traffic.json for the caveat
        # on the gfx950 FETCH_SIZE calibration.
        pmc, pmc_file = {}, None
        for cand in ("r02p_hbm_traffic.json", "r02_hbm_traffic.json", "r01k_hbm_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", cand)) as f:
                    pmc, pmc_file = json.load(f)["kernels"], cand
                break
            except OSError:
                continue
        def roof(ms, kernel):
            a = alg / (ms * 1e-3) / 1e9
            t = pmc.get(kernel)
            traffic = int((t["fetch_bytes_per_block"] + t["write_bytes_per_block"]) * nb) if t and args.hash == "crc32c" and args.config == 2 else None
            return {"bound": "hbm", "kernel": kernel, "achieved": round(a, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(a / HBM_PEAK_GBPS, 5), "traffic": traffic,
                    "traffic_source": (f"replayed: per-block FETCH_SIZE + WRITE_SIZE of profiles/{pmc_file} (separate rocprofv3 --pmc "
                                       "passes over this workload) x blocks of this run; not measured in this process") if traffic else None,
                    "avg_launch_ms": round(ms, 4),
                    "algorithmic_bytes_per_launch": int(alg),
                    "uncompressed_GBps": round(u_bytes / (ms * 1e-3) / 1e9, 2)}
        lanes = nb >= 16384
        r_c = roof(ms_c, "k_compress_lanes" if lanes else "k_compress_win")
        r_d = roof(ms_d, "k_decompress_chains" if "k_decompress_chains" in pmc else "k_decompress")
        if lanes:
            r_c["table_workspace_probe"] = {"chosen_ms": S.lib().snp_ctx_counter(cd.ctx.handle, 2) / 1e3,
                                            "candidates": S.lib().snp_ctx_counter(cd.ctx.handle, 3)}
        if lanes and args.config == 2:
            # What bounds the lane compressor is not U + C but the random 4-byte read-modify-writes of its hash tables
            # (DESIGN.md 4.3).  Rates: scripts/microbench_random_table.hip on this GPU model (profiles/
            # r01d_microbench_random_table.jsonl: 20.26 G read+write probes/s, 23.57 G write-only inserts/s); counts: the
            # reference parse makes 9928 probes + 4228 post-copy inserts per fragment of this workload (first 256 blocks, counted on the CPU
            # with the oracle).  floor = table traffic alone, nothing else in the kernel.
            floor_ms = nb * (9928 / 20.26e9 + 4228 / 23.57e9) * 1e3
            r_c["random_access_floor"] = {"table_probes_per_fragment": 9928, "table_inserts_per_fragment": 4228,
                                          "floor_ms": round(floor_ms, 1), "frac_of_floor": round(floor_ms / ms_c, 3)}
        line = {
            "metric": "uncompressed GB/s block compress+decompress, 64 KiB blocks",
            "value": round(total_u / (elapsed / args.steps) / 1e9, 3),
            "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": ("configs[1]: 10 GiB of 64 KiB html-like blocks per GPU, " if args.config == 2 else
                                    "configs[4]: 10 GiB of mixed-corpus 64 KiB blocks per GPU (80 GiB at 8 GPUs), ") +
                                   "step = compress all + decompress all (+ RCCL length/status gather when N > 1)",
                       "blocks_per_gpu": nb, "block_bytes": BLOCK, "hash_variant": args.hash,
                       "layout": "decompress: one block per wavefront (sub-chain tag parse over 2 KiB super-windows, 64 tags per execution batch staged in LDS); compress: one fragment per lane with HBM tables (>= 16384 fragments), else one per wavefront with the table in LDS",
                       "rccl_ranks": dist.get_world_size() if distributed else 1,
                       "compression_ratio": round(c_bytes / u_bytes, 4), "parallelism": f"block-sharded x{world}, no data-path collective"},
            "compress_GBps": round(u_bytes * world / (ms_c * 1e-3) / 1e9, 2) if world == 1 else None,
            "decompress_GBps": round(u_bytes * world / (ms_d * 1e-3) / 1e9, 2) if world == 1 else None,
            "roofline": r_c if ms_c >= ms_d else r_d,               # the dominant kernel
            "roofline_compress": r_c, "roofline_decompress": r_d,
            "verified": "decode(encode(x)) == x for every block, all status OK",
        }
        if extra:
            line["config5_lines"] = extra
        if world == 1 and not args.no_cpu_baseline:
            ns = min(args.cpu_sample_blocks, nb)
            line["cpu_baseline"] = cpu_baseline(raw[: ns * BLOCK].cpu().numpy(), variant)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
