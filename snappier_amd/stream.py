"""SnappyStream: file-like wrapper for the Snappy framing format, mirroring Snappier/SnappyStream.cs:14-658.

Only the format rules live here (SnappyStreamCompressor.cs:18-21,166-261; SnappyStreamDecompressor.cs:38-208); all
codec and CRC arithmetic runs on the GPU through snp_frame_encode / snp_frame_decode.  The .NET plumbing (async,
8192-byte inner reads, single-async-op guard) is out of scope.
"""
from __future__ import annotations

import enum
import io

from . import _native as N
from .context import Context, default_context
from .errors import InvalidOperationException
from .snappy import frame_decode, frame_encode

_HEADER = bytes([0xFF, 0x06, 0x00, 0x00, 0x73, 0x4E, 0x61, 0x50, 0x70, 0x59])


class CompressionMode(enum.Enum):
    Decompress = 0
    Compress = 1


class SnappyStream(io.RawIOBase):
    def __init__(self, stream, mode: CompressionMode, leaveOpen: bool = False, ctx: Context | None = None):
        super().__init__()
        self._inner = stream
        self._mode = mode
        self._leave_open = leaveOpen
        self._ctx = ctx or default_context()
        self._pending = bytearray()     # compress: bytes not yet forming a whole 64 KiB chunk
        self._header_written = False
        self._decoded = None            # decompress: decoded payload, filled on first read
        self._rpos = 0

    # ---- capabilities (SnappyStream.cs:88-96) ----
    def readable(self):
        return self._mode == CompressionMode.Decompress

    def writable(self):
        return self._mode == CompressionMode.Compress

    def seekable(self):
        return False

    # ---- compress side ----
    def _emit(self, raw: bytes):
        enc = frame_encode(raw, self._ctx)
        if self._header_written:
            enc = enc[len(_HEADER):]                 # EnsureStreamHeaderWritten: once  SnappyStreamCompressor.cs:148-157
        self._header_written = True
        self._inner.write(enc)

    def write(self, b) -> int:
        if self._mode != CompressionMode.Compress:
            raise InvalidOperationException("Cannot write to a decompression stream.")
        self._pending += bytes(b)
        whole = len(self._pending) // N.BLOCK_SIZE * N.BLOCK_SIZE     # CompressInput: only full 64 KiB chunks  :166-192
        if whole:
            self._emit(bytes(self._pending[:whole]))
            del self._pending[:whole]
        return len(b)

    def flush(self):
        if self._mode == CompressionMode.Compress and not self.closed:
            if self._pending or not self._header_written:             # Flush: partial chunk  :82-97
                self._emit(bytes(self._pending))
                self._pending.clear()
            if hasattr(self._inner, "flush"):
                self._inner.flush()

    # ---- decompress side ----
    def _fill(self):
        if self._decoded is None:
            self._decoded = frame_decode(self._inner.read(), self._ctx)

    def readinto(self, b) -> int:
        if self._mode != CompressionMode.Decompress:
            raise InvalidOperationException("Cannot read from a compression stream.")
        self._fill()
        n = min(len(b), len(self._decoded) - self._rpos)
        b[:n] = self._decoded[self._rpos:self._rpos + n]
        self._rpos += n
        return n

    def close(self):
        if not self.closed:
            try:
                self.flush()
            finally:
                super().close()
                if not self._leave_open and hasattr(self._inner, "close"):
                    self._inner.close()
