#!/usr/bin/env python3
"""Hand-built, legal Snappy streams that no 64 KiB-fragment compressor emits, chosen against the sub-chain decoder (HISTORY.md §4.1c):
tag periods that never put a region's first byte on a tag start, literal bodies made of long-literal tag bytes, copy-4 tags.
Each stream is checked against the oracle and timed through every decoder front end.   python scripts/adversarial_streams.py [blocks]
Prints one JSON line per (stream, front end)."""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def varint(v):
    out = bytearray()
    while v >= 128:
        out.append((v & 127) | 128)
        v >>= 7
    out.append(v)
    return bytes(out)


def lit(body):
    n = len(body)
    assert 1 <= n
    if n <= 60:
        return bytes([(n - 1) << 2]) + body
    if n <= 256:
        return bytes([60 << 2, n - 1]) + body
    return bytes([61 << 2, (n - 1) & 255, (n - 1) >> 8]) + body


def copy4(length, off):
    return bytes([3 | ((length - 1) << 2)]) + int(off).to_bytes(4, "little")


def copy2(length, off):
    return bytes([2 | ((length - 1) << 2)]) + int(off).to_bytes(2, "little")


def build(kind, total=65536):
    out = bytearray()
    produced = 0
    def emit(tag, n):
        nonlocal produced
        out.extend(tag)
        produced += n
    if kind == "copy4_len4_period5":            # 5-byte tags: a region start (multiple of 32) hits a tag start every 160 bytes only
        emit(lit(b"abcd"), 4)
        while produced + 4 <= total:
            emit(copy4(4, 4), 4)
    elif kind == "copy4_len64_period5":
        emit(lit(b"abcd"), 4)
        while produced + 64 <= total:
            emit(copy4(64, 4), 64)
    elif kind == "literals_of_f4":              # literal bodies of 0xF4: every chain that starts inside one reads a 3-byte-length literal
        while produced + 60 <= total:
            emit(lit(bytes([0xF4]) * 60), 60)
    elif kind == "literals_of_ff_period61":     # 60-byte literals of 0xFF (as a tag: copy-4, 5 bytes): chains inside the bodies step by 5 and meet a
        while produced + 60 <= total:           # tag start (every 61 bytes) only after ~300 bytes: the true chain does NOT merge within the cap
            emit(lit(bytes([0xFF]) * 60), 60)
    elif kind == "literals_of_14_period7":      # every byte 0x14 = "literal of 6": all seven phases are self-consistent chains, only every
        while produced + 6 <= total:            # seventh lane starts on the true one (224 bytes apart): the fallback runs in every super-window
            emit(lit(bytes([0x14]) * 6), 6)
    elif kind == "copy2_offsets_f4f4":          # copy-2 tags whose offset bytes are long-literal tag bytes (offset 0xF4F4 needs 62 KiB behind it: use a literal run first)
        emit(lit(bytes(range(256)) * 250), 64000)
        while produced + 4 <= total:
            emit(copy2(4, 0xF4F4 if produced >= 0xF4F4 else 4), 4)
    elif kind == "period7_mix":                 # literal(1) + copy-4: 2 + 5 = 7-byte period
        emit(lit(b"abcdefgh"), 8)
        while produced + 5 <= total:
            emit(lit(b"x"), 1)
            emit(copy4(4, 8), 4)
    elif kind == "edge_sweep":                  # built against the TWO-BATCH execution (decode_chains.hip): 1-byte literals and copies whose (offset, length)
        import random                           # sweep every relation to a batch of 64 tags -- source ends / starts at the held batch's first byte, in the
        rnd = random.Random(7)                  # 64 bytes of history below it, one byte beyond the history, inside the own batch, pattern copies
        emit(lit(bytes(range(200))), 200)
        k = 0
        while produced + 70 <= total:
            if k % 3 == 0:
                emit(lit(bytes([rnd.randrange(256)])), 1)
            else:
                ln = 1 + (k * 7) % 64
                off = min(produced, 1 + (k * 13) % 260 if k % 5 else rnd.choice((1, 2, 3, 5, 63, 64, 65, 66, 127, 128, 129)))
                emit(copy2(ln, off), ln)
            k += 1
    elif kind == "long_literals_between_batches":   # literals of 65..130 bytes (copied by the whole wave, nothing held across them) between short runs of
        import random                               # copies that reach back over them: the first batch after a drain has no history in the stage
        rnd = random.Random(11)
        emit(lit(bytes(range(100))), 100)
        k = 0
        while produced + 400 <= total:
            n = 65 + (k * 5) % 66
            emit(lit(bytes((k + i) & 255 for i in range(n))), n)
            for j in range(1 + k % 7):
                ln = 4 + (k + 3 * j) % 61
                off = min(produced, (1, 3, 60, 64, 65, 70, 130, 131, 200)[(k + j) % 9])
                emit(copy2(ln, off), ln)
            k += 1
    elif kind == "two_slot_literals":           # literals of 64 .. 130 bytes -- one slot of a batch, two slots (65 .. 128), the whole wave (> 128) -- between
        import random                           # runs of short tags of every length, so that the second slot falls on every lane incl. the 64th (it must then
        rnd = random.Random(23)                 # wait for the next batch with its first), on the stage's limit, and -- the last literal -- on the block's end
        emit(lit(bytes(range(90))), 90)
        k = 0
        lens = (64, 65, 66, 100, 127, 128, 129, 130, 96, 65, 128)
        while produced + 400 <= total - 128:
            n = lens[k % len(lens)]
            emit(lit(bytes((3 * k + i) & 255 for i in range(n))), n)
            for j in range((k * 7) % 67):
                if j % 3:
                    emit(copy2(1 + (k + j) % 9, min(produced, 1 + (5 * j + k) % 300)), 1 + (k + j) % 9)
                else:
                    emit(lit(bytes([rnd.randrange(256)])), 1)
            if k % 5 == 0:                                                  # a long run of 64-byte copies: the stage fills up (2 KiB) in mid-batch
                for j in range(40):
                    emit(copy2(64, min(produced, 64 + j)), 64)
            k += 1
        tail = total - produced
        if 65 <= tail <= 128:
            emit(lit(bytes(tail)), tail)
        else:
            emit(lit(bytes(tail - 100)), tail - 100)
            emit(lit(bytes(range(100))), 100)                               # a two-slot literal that ends with the block
    if produced < total:
        emit(lit(bytes(total - produced)), total - produced)
    return varint(total) + bytes(out)


KINDS = ("copy4_len4_period5", "copy4_len64_period5", "literals_of_f4", "literals_of_ff_period61", "literals_of_14_period7",
         "copy2_offsets_f4f4", "period7_mix", "edge_sweep", "long_literals_between_batches", "two_slot_literals")


def main():
    import torch
    import snappier_amd as S
    from snappier_amd import batch as SB
    from oracle import pyoracle as O
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    mode = os.environ.get("SNAPPIER_HIP_DECODE", "chains")
    for kind in KINDS:
        stream = build(kind)
        want = O.decompress(stream)
        assert len(want) == 65536
        cd = SB.BlockCodec(0, S.HASH_CRC32C)
        one = torch.from_numpy(np.frombuffer(stream, dtype=np.uint8).copy()).cuda()
        stride = (len(stream) + 63) // 64 * 64
        comp = torch.zeros(nb * stride, dtype=torch.uint8, device="cuda")
        comp.view(nb, stride)[:, : len(stream)] = one
        in_off = torch.arange(nb, dtype=torch.int64, device="cuda") * stride
        in_len = torch.full((nb,), len(stream), dtype=torch.int32, device="cuda")
        out = torch.zeros(nb * 65536, dtype=torch.uint8, device="cuda")
        out_off = torch.arange(nb, dtype=torch.int64, device="cuda") * 65536
        out_cap = torch.full((nb,), 65536, dtype=torch.int32, device="cuda")
        ms = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dlen, st = cd.decompress(comp, in_off, in_len, out, out_off, out_cap)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        wt = torch.from_numpy(np.frombuffer(want, dtype=np.uint8).copy()).cuda()
        ok = bool((st == 0).all()) and bool((out.view(nb, 65536) == wt).all())
        tags = None
        print(json.dumps({"stream": kind, "decode": mode, "blocks": nb, "compressed_bytes": len(stream), "ms": round(min(ms), 3),
                          "GBps_uncompressed": round(nb * 65536 / min(ms) / 1e6, 1), "equals_oracle": ok}), flush=True)


if __name__ == "__main__":
    main()
