"""Device-resident batch API: thousands of independent 64 KiB blocks per launch, torch tensors as HBM buffers.

torch is plumbing here (allocation, streams, torch.distributed); the work is done by the snp_*_batch entry points.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _native as N
from .context import Context
from .errors import raise_for_status


def _p(t: torch.Tensor | None):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() else C.c_void_p(None)


class BlockCodec:
    """Wraps one snp_ctx bound to torch's current stream on `device`."""

    def __init__(self, device: int | torch.device = 0, hash_variant: int = N.HASH_CRC32C):
        self.device = torch.device("cuda", device) if isinstance(device, int) else device
        torch.cuda.set_device(self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.ctx = Context(self.device.index or 0, hash_variant, stream=stream)
        self.comp_stride = (N.lib().snp_max_compressed_length(N.BLOCK_SIZE) + 15) // 16 * 16

    def _bind(self):
        """Follow torch's current stream, so our launches are ordered with the tensors' producers and consumers."""
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    # -- layout helpers --------------------------------------------------------------------------------------
    def uniform_layout(self, nblocks: int, block: int = N.BLOCK_SIZE, last_len: int | None = None):
        off = torch.arange(nblocks, dtype=torch.int64, device=self.device) * block
        ln = torch.full((nblocks,), block, dtype=torch.int32, device=self.device)
        if last_len is not None and nblocks:
            ln[-1] = last_len
        return off, ln

    # -- hot path --------------------------------------------------------------------------------------------
    def compress(self, data: torch.Tensor, in_off: torch.Tensor, in_len: torch.Tensor, out: torch.Tensor | None = None,
                 out_off: torch.Tensor | None = None):
        """-> (out, out_off, out_len, status).  Block b's output starts at out_off[b] (default stride comp_stride)."""
        self._bind()
        nb = in_len.numel()
        if out_off is None:
            out_off = torch.arange(nb, dtype=torch.int64, device=self.device) * self.comp_stride
        if out is None:
            out = torch.empty(nb * self.comp_stride, dtype=torch.uint8, device=self.device)
        out_len = torch.empty(nb, dtype=torch.int32, device=self.device)
        status = torch.empty(nb, dtype=torch.int32, device=self.device)
        st = self.ctx.lib.snp_compress_batch(self.ctx.handle, _p(data), _p(in_off), _p(in_len), nb, _p(out), _p(out_off),
                                        _p(out_len), _p(status))
        raise_for_status(st, self.ctx.handle)
        return out, out_off, out_len, status

    def decompress(self, comp: torch.Tensor, in_off: torch.Tensor, in_len: torch.Tensor, out: torch.Tensor,
                   out_off: torch.Tensor, out_cap: torch.Tensor):
        """-> (out_len, status)."""
        self._bind()
        nb = in_len.numel()
        out_len = torch.empty(nb, dtype=torch.int32, device=self.device)
        status = torch.empty(nb, dtype=torch.int32, device=self.device)
        st = self.ctx.lib.snp_decompress_batch(self.ctx.handle, _p(comp), _p(in_off), _p(in_len), nb, _p(out), _p(out_off),
                                          _p(out_cap), _p(out_len), _p(status))
        raise_for_status(st, self.ctx.handle)
        return out_len, status

    def compact(self, data: torch.Tensor, in_off: torch.Tensor, in_len: torch.Tensor):
        """Concatenate the blocks (snp_concat_batch): -> (stream tensor sized to the exact total, dst_off).  Needs the
        total on the host (one sync) to size the result."""
        self._bind()
        nb = in_len.numel()
        lens = in_len.to(torch.int64)
        dst_off = torch.cumsum(lens, 0) - lens
        total = int(lens.sum().item()) if nb else 0
        out = torch.empty(total, dtype=torch.uint8, device=self.device)
        st = self.ctx.lib.snp_concat_batch(self.ctx.handle, _p(data), _p(in_off), _p(in_len), nb, _p(out), _p(dst_off))
        raise_for_status(st, self.ctx.handle)
        return out, dst_off

    def crc32c(self, data: torch.Tensor, in_off: torch.Tensor, in_len: torch.Tensor, masked: bool = False):
        self._bind()
        nb = in_len.numel()
        crc = torch.empty(nb, dtype=torch.int32, device=self.device)
        st = self.ctx.lib.snp_crc32c_batch(self.ctx.handle, _p(data), _p(in_off), _p(in_len), nb, int(masked), _p(crc))
        raise_for_status(st, self.ctx.handle)
        return crc

    # -- framing, device resident (config 4) --------------------------------------------------------------------
    def frame_encode(self, raw: torch.Tensor, out: torch.Tensor | None = None, work: torch.Tensor | None = None):
        """-> (framed tensor (capacity-sized), written: 1-element int64 tensor on device).  `out` / `work` may be
        passed in to reuse buffers across calls (sizes: snp_frame_max_encoded_length / snp_frame_encode_workspace)."""
        self._bind()
        n = raw.numel()
        cap = N.lib().snp_frame_max_encoded_length(n)
        need = N.lib().snp_frame_encode_workspace(n)
        if out is None:
            out = torch.empty(cap, dtype=torch.uint8, device=self.device)
        if work is None:
            work = torch.empty(need, dtype=torch.uint8, device=self.device)
        if out.numel() < cap or work.numel() < need:
            raise ValueError("frame_encode: out/work buffers too small")
        written = torch.zeros(1, dtype=torch.int64, device=self.device)
        st = self.ctx.lib.snp_frame_encode_device(self.ctx.handle, _p(raw), n, _p(out), cap, _p(written), _p(work))
        raise_for_status(st, self.ctx.handle)
        return out, written

    def frame_decode_chunks(self, framed: torch.Tensor, chunk_type, body_off, body_len, chunk_crc, out, out_off, out_cap):
        self._bind()
        nc = body_len.numel()
        out_len = torch.empty(nc, dtype=torch.int32, device=self.device)
        status = torch.empty(nc, dtype=torch.int32, device=self.device)
        st = self.ctx.lib.snp_frame_decode_chunks_device(self.ctx.handle, _p(framed), _p(chunk_type), _p(body_off),
                                                    _p(body_len), _p(chunk_crc), nc, _p(out), _p(out_off), _p(out_cap),
                                                    _p(out_len), _p(status))
        raise_for_status(st, self.ctx.handle)
        return out_len, status

    def frame_decode(self, framed: torch.Tensor, nbytes: int, out: torch.Tensor, max_chunks: int, work: torch.Tensor | None = None):
        """Framed stream without a chunk table, all on the device (snp_frame_decode_device): -> 2-element int64 tensor
        (bytes written, status).  The chunk headers are walked by a device kernel (serial, ~1 us per chunk)."""
        self._bind()
        need = N.lib().snp_frame_decode_workspace(max_chunks)
        if work is None:
            work = torch.empty(need, dtype=torch.uint8, device=self.device)
        if work.numel() < need:
            raise ValueError("frame_decode: work buffer too small")
        result = torch.zeros(2, dtype=torch.int64, device=self.device)
        st = self.ctx.lib.snp_frame_decode_device(self.ctx.handle, _p(framed), nbytes, _p(out), out.numel(), max_chunks, _p(work), _p(result))
        raise_for_status(st, self.ctx.handle)
        return result
