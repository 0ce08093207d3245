"""snappier_amd -- MI355X (gfx950) Snappy block codec behind Snappier's API.

The package is a thin host mirror of the reference's public surface (Snappy, SnappyStream) plus a device-resident
batch API; all codec work happens in hand-written HIP kernels (csrc/) reached through the C-ABI in
include/snappier_hip.h.  Importing it requires the built libsnappier_hip.so -- there is no CPU fallback.
"""
from ._native import (BLOCK_SIZE, HASH_CRC32C, HASH_MUL, MAX_BLOCK_COMPRESSED, lib, status_string)  # noqa: F401
from .context import Context, default_context  # noqa: F401
from .errors import InsufficientBufferException, InvalidDataException, InvalidOperationException  # noqa: F401
from .snappy import Snappy, crc32c, frame_decode, frame_encode  # noqa: F401
from .stream import CompressionMode, SnappyStream  # noqa: F401
from .multidevice import MultiDeviceCodec  # noqa: F401

lib()   # fail loudly at import time if the native library is missing or incomplete
