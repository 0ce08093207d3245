"""Several devices behind ONE caller: a chunk codec that keeps a context per device and stripes whole-chunk ranges over them.

Why it needs no collective: the framing format's chunks (SnappyStreamCompressor.cs:166-230: every 64 KiB of input becomes one self-contained
chunk) and the block format's fragments (SnappyCompressor.cs:40-80: the hash table is reset every 65 536 bytes) are independent, and one process
sees every GPU of the node.  So the host cuts the input into contiguous ranges of whole chunks, each range goes through one context on one
worker thread (host-pointer entry points of include/snappier_hip.h, nothing new in the C-ABI), and the host concatenates the results in range
order -- the "length directory" is the list of bytes each range produced.  `devices` may name a device more than once ([0, 0]: two contexts,
two streams on one GPU), which is how the one-GPU test box exercises it.  The torch.distributed form for device-resident batches (one process
per GPU, RCCL directory gather) is snappier_amd/sharding.py; the C# twin is csharp/Snappier.Gpu/MultiDeviceChunkCodec.cs.
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _native as N
from .context import Context
from .snappy import Snappy, _view, frame_decode, frame_encode

_HEADER_LEN = 10        # ff 06 00 00 "sNaPpY"


def split_ranges(nchunks: int, workers: int, min_chunks: int):
    """Contiguous ranges of whole chunks, one per worker (fewer when the input is small): [(first, count), ...]."""
    k = min(workers, max(1, nchunks // max(1, min_chunks)))
    base, extra = divmod(nchunks, k)
    out, first = [], 0
    for r in range(k):
        cnt = base + (1 if r < extra else 0)
        out.append((first, cnt))
        first += cnt
    return out


def chunk_table(src: np.ndarray):
    """([(offset, total bytes) of every chunk], end) of a framed buffer, from the 4-byte chunk headers alone (SnappyStreamDecompressor.cs:53-75);
    a malformed tail ends the table early (end < len(src)): the caller then lets one context report it exactly as the reference would."""
    table, pos, n = [], 0, int(src.size)
    while pos + 4 <= n:
        size = int(src[pos + 1]) | (int(src[pos + 2]) << 8) | (int(src[pos + 3]) << 16)
        if pos + 4 + size > n:
            break
        table.append((pos, 4 + size))
        pos += 4 + size
    return table, pos


class MultiDeviceCodec:
    def __init__(self, devices, hash_variant: int = N.HASH_CRC32C, min_chunks_per_range: int = 16):
        if not devices:
            raise ValueError("devices: at least one device index")
        self.contexts = [Context(d, hash_variant) for d in devices]
        self._pool = ThreadPoolExecutor(max_workers=len(self.contexts))
        self.min_chunks = max(1, int(min_chunks_per_range))
        self.last_directory: list[int] = []          # bytes each range produced in the last call (the merged length directory)

    def close(self):
        self._pool.shutdown(wait=True)
        for c in self.contexts:
            c.close()
        self.contexts = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ranges(self, nchunks: int):
        return split_ranges(nchunks, len(self.contexts), self.min_chunks)

    def _run(self, jobs):
        """jobs: [(context index, callable)] -> results in job order; the first failing job IN ORDER raises (stream order, as the
        sequential reference reports the first bad chunk)."""
        futs = [self._pool.submit(fn) for _i, fn in jobs]
        results, first_exc = [], None
        for f in futs:
            try:
                results.append(f.result())
            except Exception as e:      # noqa: BLE001  (every job is awaited: no worker keeps running behind a raised exception)
                results.append(None)
                first_exc = first_exc or e
        if first_exc:
            raise first_exc
        return results

    # ---- framing format --------------------------------------------------------------------------------------------------
    def frame_encode(self, data) -> bytes:
        """SnappyStream compress of one buffer: the same bytes as frame_encode on one device (stream identifier once, then the chunks)."""
        src = _view(data)
        nchunks = (src.size + N.BLOCK_SIZE - 1) // N.BLOCK_SIZE
        if nchunks <= self.min_chunks:
            out = frame_encode(src, self.contexts[0])
            self.last_directory = [len(out)]
            return out
        jobs = []
        for i, (first, cnt) in enumerate(self._ranges(nchunks)):
            piece = src[first * N.BLOCK_SIZE:min(src.size, (first + cnt) * N.BLOCK_SIZE)]
            jobs.append((i, (lambda p=piece, c=self.contexts[i]: frame_encode(p, c))))
        parts = self._run(jobs)
        parts = [parts[0]] + [p[_HEADER_LEN:] for p in parts[1:]]        # EnsureStreamHeaderWritten: once  SnappyStreamCompressor.cs:148-157
        self.last_directory = [len(p) for p in parts]
        return b"".join(parts)

    def frame_decode(self, data) -> bytes:
        """SnappyStream decompress with CRC verification; the first failing chunk in stream order decides the exception."""
        src = _view(data)
        table, end = chunk_table(src)
        if len(table) <= self.min_chunks or end != src.size:
            out = frame_decode(src, self.contexts[0])          # small, or a malformed tail: one context reports exactly what the reference would
            self.last_directory = [len(out)]
            return out
        jobs = []
        for i, (first, cnt) in enumerate(self._ranges(len(table))):
            lo = table[first][0]
            hi = table[first + cnt - 1][0] + table[first + cnt - 1][1]
            jobs.append((i, (lambda p=src[lo:hi], c=self.contexts[i]: frame_decode(p, c))))
        parts = self._run(jobs)
        self.last_directory = [len(p) for p in parts]
        return b"".join(parts)

    # ---- block format ----------------------------------------------------------------------------------------------------
    def compress(self, data) -> bytes:
        """Snappy.CompressToArray of one buffer: varint(length) then the fragments, each range of fragments on its own device."""
        src = _view(data)
        nfrag = (src.size + N.BLOCK_SIZE - 1) // N.BLOCK_SIZE
        if nfrag <= self.min_chunks:
            out = Snappy.CompressToArray(src, self.contexts[0])
            self.last_directory = [len(out)]
            return out

        def varint_len(v: int) -> int:
            k = 1
            while v >= 128:
                v >>= 7
                k += 1
            return k

        def varint(v: int) -> bytes:
            b = bytearray()
            while v >= 128:
                b.append((v & 0x7F) | 0x80)
                v >>= 7
            b.append(v)
            return bytes(b)

        jobs = []
        for i, (first, cnt) in enumerate(self._ranges(nfrag)):
            piece = src[first * N.BLOCK_SIZE:min(src.size, (first + cnt) * N.BLOCK_SIZE)]
            # a range compressed on its own starts with the varint of ITS length: dropped, the whole buffer's goes in front once
            jobs.append((i, (lambda p=piece, c=self.contexts[i]: Snappy.CompressToArray(p, c)[varint_len(p.size):])))
        parts = self._run(jobs)
        self.last_directory = [len(p) for p in parts]
        return varint(src.size) + b"".join(parts)
