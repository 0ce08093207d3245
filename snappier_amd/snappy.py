"""Host-side mirror of Snappier's public block API (Snappier/Snappy.cs:10-283) over the C-ABI.

Same names, argument meaning and error behaviour as the reference's static class, so the parity tests read like
Snappier.Tests/SnappyTests.cs.  Every call runs on the GPU through libsnappier_hip.so; buffers are bytes-like.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .context import Context, default_context
from .errors import InsufficientBufferException, InvalidDataException, raise_for_status


def _view(b):
    a = b if isinstance(b, np.ndarray) else np.frombuffer(b, dtype=np.uint8)
    if a.dtype != np.uint8 or not a.flags.c_contiguous:
        a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


def _ptr(a: np.ndarray):
    return C.c_void_p(a.ctypes.data if a.size else None)


class Snappy:
    """Routines for performing Snappy compression and decompression on raw data blocks (Snappy.cs:8-10)."""

    @staticmethod
    def GetMaxCompressedLength(inputLength: int) -> int:                     # Snappy.cs:20-24
        v = N.lib().snp_max_compressed_length(inputLength)
        if v < 0:
            raise ValueError("inputLength")
        return v

    @staticmethod
    def TryCompress(input, output, ctx: Context | None = None):             # Snappy.cs:55-67 -> (ok, bytesWritten)
        ctx = ctx or default_context()
        src, dst = _view(input), output
        w = C.c_size_t(0)
        st = ctx.lib.snp_try_compress(ctx.handle, _ptr(src), src.size, _ptr(dst), dst.size, C.byref(w))
        if st == N.ERR_OUTPUT_TOO_SMALL:
            return False, 0
        raise_for_status(st, ctx.handle)
        return True, w.value

    @staticmethod
    def Compress(input, output, ctx: Context | None = None) -> int:         # Snappy.cs:37-45
        ok, n = Snappy.TryCompress(input, output, ctx)
        if not ok:
            raise InsufficientBufferException("Output buffer is too small.")
        return n

    @staticmethod
    def CompressToArray(input, ctx: Context | None = None) -> bytes:        # Snappy.cs:121-129 (and CompressToMemory :99-113)
        src = _view(input)
        buf = np.empty(Snappy.GetMaxCompressedLength(src.size), dtype=np.uint8)
        n = Snappy.Compress(src, buf, ctx)
        return buf[:n].tobytes()

    CompressToMemory = CompressToArray

    @staticmethod
    def GetUncompressedLength(input) -> int:                                 # Snappy.cs:136-137
        src = _view(input)
        v = C.c_uint32(0)
        st = N.lib().snp_get_uncompressed_length(_ptr(src), src.size, C.byref(v), None)
        raise_for_status(st)
        return v.value

    @staticmethod
    def TryDecompress(input, output, ctx: Context | None = None):           # Snappy.cs:172-186 -> (ok, bytesWritten)
        ctx = ctx or default_context()
        src, dst = _view(input), output
        w = C.c_size_t(0)
        st = ctx.lib.snp_try_decompress(ctx.handle, _ptr(src), src.size, _ptr(dst), dst.size, C.byref(w))
        if st == N.ERR_OUTPUT_TOO_SMALL:
            return False, 0
        raise_for_status(st, ctx.handle)
        return True, w.value

    @staticmethod
    def Decompress(input, output, ctx: Context | None = None) -> int:       # Snappy.cs:153-162
        ok, n = Snappy.TryDecompress(input, output, ctx)
        if not ok:
            raise InsufficientBufferException("Output buffer is too small.")
        return n

    @staticmethod
    def DecompressToArray(input, ctx: Context | None = None) -> bytes:      # Snappy.cs:271-281 (and DecompressToMemory :223-235)
        src = _view(input)
        try:
            length = Snappy.GetUncompressedLength(src)
        except InvalidDataException:
            if src.size == 0 or (src.size < 5 and all(int(b) & 0x80 for b in src)):
                raise InvalidDataException(N.ERR_INCOMPLETE)                  # preamble cut short: Snappy.cs:229-232
            raise
        buf = np.empty(length, dtype=np.uint8)
        n = Snappy.Decompress(src, buf, ctx)
        return buf[:n].tobytes()

    DecompressToMemory = DecompressToArray


def crc32c(data, masked: bool = False, ctx: Context | None = None) -> int:
    """Crc32CAlgorithm.Compute (+ ApplyMask when masked)  Crc32CAlgorithm.cs:41-44,156-158 -- on the GPU."""
    ctx = ctx or default_context()
    src = _view(data)
    v = C.c_uint32(0)
    st = ctx.lib.snp_crc32c(ctx.handle, _ptr(src), src.size, int(masked), C.byref(v))
    raise_for_status(st, ctx.handle)
    return v.value


def frame_encode(data, ctx: Context | None = None) -> bytes:
    """Whole-buffer SnappyStream compress: stream identifier + one chunk per 64 KiB (SnappyStreamCompressor.cs)."""
    ctx = ctx or default_context()
    src = _view(data)
    cap = N.lib().snp_frame_max_encoded_length(src.size)
    out = np.empty(cap, dtype=np.uint8)
    w = C.c_size_t(0)
    st = ctx.lib.snp_frame_encode(ctx.handle, _ptr(src), src.size, _ptr(out), out.size, C.byref(w))
    raise_for_status(st, ctx.handle)
    return out[: w.value].tobytes()


def frame_decode(data, ctx: Context | None = None) -> bytes:
    """Whole-buffer SnappyStream decompress with CRC verification (SnappyStreamDecompressor.cs:38-208)."""
    ctx = ctx or default_context()
    src = _view(data)
    total = C.c_uint64(0)
    st = N.lib().snp_frame_decoded_length(_ptr(src), src.size, C.byref(total))
    # header-walk errors are re-reported by snp_frame_decode in stream order, after the chunks before them
    out = np.empty(total.value, dtype=np.uint8)
    w = C.c_size_t(0)
    st = ctx.lib.snp_frame_decode(ctx.handle, _ptr(src), src.size, _ptr(out), out.size, C.byref(w))
    raise_for_status(st, ctx.handle)
    return out[: w.value].tobytes()
