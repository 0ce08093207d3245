"""Multi-GPU sharding of independent blocks.

Every 64 KiB block / framing chunk is independent in both directions (SnappyCompressor.cs:40-80 clears the hash
table per fragment; framing chunks are self-contained, SnappyStreamCompressor.cs:194-230), so blocks are dealt to
ranks in contiguous ranges and the codec itself needs NO collective.  The only exchange step is the final
directory gather: all_gather of the per-block compressed lengths + status words (8 B per block), from which every
rank derives the global output offsets; the payload gather to one rank is optional and reported separately
(gather-to-one is inbound-limited on xGMI: 7 links x ~153 GB/s).
One process per GPU; backend "nccl" (= RCCL) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(nblocks: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [first, last) block range of `rank`; sizes differ by at most one."""
    base, rem = divmod(nblocks, world)
    first = rank * base + min(rank, rem)
    return first, first + base + (1 if rank < rem else 0)


def gather_directory(out_len: torch.Tensor, status: torch.Tensor, nblocks: int, group=None):
    """all_gather the per-block (length, status) of every rank's contiguous shard.

    Returns (all_len, all_status, offsets): int64 tensors of size nblocks; offsets = exclusive prefix sum of the
    lengths = where block b lands in the concatenated compressed stream."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    per = (nblocks + world - 1) // world
    first, last = shard_range(nblocks, rank, world)
    assert out_len.numel() == last - first == status.numel()
    packed = torch.zeros(per, 2, dtype=torch.int64, device=out_len.device)
    packed[: last - first, 0] = out_len.to(torch.int64)
    packed[: last - first, 1] = status.to(torch.int64)
    if world > 1:
        gathered = torch.empty(world * per, 2, dtype=torch.int64, device=out_len.device)
        dist.all_gather_into_tensor(gathered, packed, group=group)
    else:
        gathered = packed
    pieces = []
    for r in range(world):
        f, l = shard_range(nblocks, r, world)
        pieces.append(gathered[r * per: r * per + (l - f)])
    allv = torch.cat(pieces, dim=0)
    all_len, all_status = allv[:, 0].contiguous(), allv[:, 1].contiguous()
    offsets = torch.cumsum(all_len, 0) - all_len
    return all_len, all_status, offsets


def gather_payload(compact: torch.Tensor, all_len: torch.Tensor, nblocks: int, dst: int = 0, group=None):
    """Optional: concatenate every rank's compacted compressed bytes on rank `dst` (point-to-point over xGMI)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        return compact
    sizes = []
    for r in range(world):
        f, l = shard_range(nblocks, r, world)
        sizes.append(int(all_len[f:l].sum().item()))
    if rank == dst:
        out = torch.empty(sum(sizes), dtype=torch.uint8, device=compact.device)
        pos, reqs = 0, []
        for r in range(world):
            view = out[pos: pos + sizes[r]]
            if r == dst:
                view.copy_(compact[: sizes[r]])
            elif sizes[r]:
                reqs.append(dist.irecv(view, src=r, group=group))
            pos += sizes[r]
        for q in reqs:
            q.wait()
        return out
    if sizes[rank]:
        dist.send(compact[: sizes[rank]].contiguous(), dst=dst, group=group)
    return None


def rank_stats(values: dict, device=None, group=None) -> dict:
    """Per-rank scalars (kernel ms, seconds of workspace search ...) -> {name: {"min", "max", "mean", "per_rank": [...]}} on every rank:
    one all_gather of a float64 vector.  Lets a multi-GPU bench line be decomposed: which rank was slow, and in which part."""
    names = sorted(values)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    mine = torch.tensor([float(values[k]) for k in names], dtype=torch.float64, device=device)
    if world > 1:
        allv = torch.empty(world * len(names), dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(allv, mine, group=group)
        allv = allv.view(world, len(names)).cpu()
    else:
        allv = mine.view(1, len(names)).cpu()
    return {k: {"min": float(allv[:, i].min()), "max": float(allv[:, i].max()), "mean": float(allv[:, i].mean()),
                "per_rank": [round(float(v), 4) for v in allv[:, i]]} for i, k in enumerate(names)}
