"""Exception types mirroring the reference's error taxonomy (Snappier/Internal/ThrowHelper.cs:8-36)."""
from . import _native as N


class InvalidDataException(ValueError):
    """System.IO.InvalidDataException: corrupt block / stream (bad offset, data too long, incomplete block,
    invalid stream length, chunk CRC mismatch, unknown chunk type)."""

    def __init__(self, status: int, message: str | None = None):
        super().__init__(message or N.status_string(status))
        self.status = status


class InsufficientBufferException(ValueError):
    """System.ArgumentException("Output buffer is too small.")  ThrowHelper.cs:18-19"""


class InvalidOperationException(RuntimeError):
    """System.InvalidOperationException (overlapping spans, device failure, misuse)."""


_INVALID_DATA = {N.ERR_BAD_OFFSET, N.ERR_TOO_LONG, N.ERR_INCOMPLETE, N.ERR_BAD_LENGTH, N.ERR_CRC_MISMATCH,
                 N.ERR_CHUNK_TYPE, N.ERR_TRUNCATED_STREAM}


def raise_for_status(st: int, ctx=None):
    if st == N.OK:
        return
    if st in _INVALID_DATA:
        raise InvalidDataException(st)
    if st == N.ERR_OUTPUT_TOO_SMALL:
        raise InsufficientBufferException(N.status_string(st))
    if st == N.ERR_BAD_ARG:
        raise ValueError(N.status_string(st))
    detail = ""
    if st == N.ERR_DEVICE and ctx is not None:
        detail = ": " + N.lib().snp_ctx_last_error(ctx).decode()
    raise InvalidOperationException(N.status_string(st) + detail)
