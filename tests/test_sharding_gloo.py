"""world_size-2 gloo test of the multi-GPU path's host logic (snappier_amd/sharding.py): contiguous block shards,
the all_gather of the (length, status) directory, global offsets and the optional payload gather.  The per-rank
"codec" here is the CPU oracle standing in for the HIP kernels (tests may use it as a stand-in; bench.py does not)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as O
from conftest import ROOT, read_testdata
import datagen


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, nblocks, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from snappier_amd import sharding
        html = read_testdata("html")
        first, last = sharding.shard_range(nblocks, rank, world)
        x = datagen.html_like_blocks(html, first, last - first)
        in_off = np.arange(last - first, dtype=np.uint64) * np.uint64(65536)
        in_len = np.full(last - first, 65536, dtype=np.uint32)
        out, out_off, out_len, status = O.compress_batch(x, in_off, in_len)
        all_len, all_status, offsets = sharding.gather_directory(torch.from_numpy(out_len.astype(np.int64)),
                                                                 torch.from_numpy(status.astype(np.int64)), nblocks)
        compact = np.concatenate([out[int(out_off[b]):int(out_off[b]) + int(out_len[b])] for b in range(last - first)])
        payload = sharding.gather_payload(torch.from_numpy(compact), all_len, nblocks, dst=0)
        stats = sharding.rank_stats({"compress_ms": 10.0 + rank, "decompress_ms": 5.0 - rank, "search_s": 0.5 * rank})   # (stub codec: made-up timings)
        if rank == 0:
            q.put((all_len.numpy(), all_status.numpy(), offsets.numpy(), payload.numpy(), stats))
    finally:
        dist.destroy_process_group()


def test_shard_ranges_cover_everything():
    from snappier_amd import sharding
    for n in (0, 1, 7, 8, 163840, 1310720):
        for w in (1, 2, 4, 8):
            r = [sharding.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_directory_and_payload_gather():
    world, nblocks = 2, 7          # ragged on purpose: ranks own 4 and 3 blocks
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nblocks, q)) for r in range(world)]
    for p in procs:
        p.start()
    all_len, all_status, offsets, payload, stats = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference of the same job
    html = read_testdata("html")
    x = datagen.html_like_blocks(html, 0, nblocks)
    ref = [O.compress(x[b * 65536:(b + 1) * 65536].tobytes()) for b in range(nblocks)]
    assert all_len.tolist() == [len(r) for r in ref]
    assert (all_status == 0).all()
    assert offsets.tolist() == np.concatenate([[0], np.cumsum([len(r) for r in ref])[:-1]]).tolist()
    assert payload.tobytes() == b"".join(ref)
    # the per-rank decomposition bench.py prints at N > 1 (sharding.rank_stats): min / max / mean / per_rank of every scalar
    assert stats["compress_ms"] == {"min": 10.0, "max": 11.0, "mean": 10.5, "per_rank": [10.0, 11.0]}
    assert stats["decompress_ms"]["per_rank"] == [5.0, 4.0] and stats["search_s"]["max"] == 0.5
    # and the concatenation decodes block by block at the gathered offsets
    for b in range(nblocks):
        s = int(offsets[b])
        assert O.decompress(payload[s:s + int(all_len[b])].tobytes()) == x[b * 65536:(b + 1) * 65536].tobytes()


def test_bench_spawns_one_process_per_gpu_by_itself():
    """`python bench.py --gpus 2` (no launcher) must re-execute itself under torch.distributed.run with a 127.0.0.1
    rendezvous: here, without a GPU, each of the two ranks then stops at the "needs an MI355X" check."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--blocks", "64"],
                       capture_output=True, text=True, timeout=300, env=env)
    text = r.stdout + r.stderr
    if "needs an MI355X" not in text:
        pytest.skip("a GPU is present: the spawn path is exercised by the driver's own multi-GPU run")
    assert r.returncode != 0
    assert "rank 0 of 2" in text and "rank 1 of 2" in text, text[-2000:]
