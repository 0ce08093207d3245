"""ctypes binding of oracle/libsnappy_oracle.so (plain-C restatement of Snappier's block codec).

TEST INFRASTRUCTURE ONLY -- the checker the HIP path is compared against, and bench.py's cpu_baseline ("port").
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsnappy_oracle.so")

OK, ERR_OUTPUT_TOO_SMALL, ERR_BAD_OFFSET, ERR_TOO_LONG, ERR_INCOMPLETE, ERR_BAD_LENGTH = 0, 1, 2, 3, 4, 5
ERR_CRC_MISMATCH, ERR_CHUNK_TYPE, ERR_OVERLAP, ERR_BAD_ARG, ERR_TRUNCATED_STREAM = 6, 7, 8, 9, 11
HASH_CRC32C, HASH_MUL = 0, 1
BLOCK_SIZE = 65536


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    src = os.path.join(_HERE, "snappy_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s", "-B", "libsnappy_oracle.so"], check=True)
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, u16p, u32p, u64p, i32p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint16, C.c_uint32, C.c_uint64, C.c_int32))
        sz = C.c_size_t
        L.orc_max_compressed_length.restype = C.c_int64
        L.orc_max_compressed_length.argtypes = [C.c_int64]
        L.orc_max_fragment_compressed_length.restype = C.c_int64
        L.orc_max_fragment_compressed_length.argtypes = [C.c_int64]
        L.orc_frame_max_encoded_length.restype = C.c_int64
        L.orc_frame_max_encoded_length.argtypes = [C.c_int64]
        L.orc_crc32c.restype = C.c_uint32
        L.orc_crc32c.argtypes = [C.c_void_p, sz]
        L.orc_crc32c_bitwise.restype = C.c_uint32
        L.orc_crc32c_bitwise.argtypes = [C.c_void_p, sz]
        L.orc_crc32c_append.restype = C.c_uint32
        L.orc_crc32c_append.argtypes = [C.c_uint32, C.c_void_p, sz]
        L.orc_crc32c_mask.restype = C.c_uint32
        L.orc_crc32c_mask.argtypes = [C.c_uint32]
        L.orc_crc32c_u32_step_bitwise.restype = C.c_uint32
        L.orc_crc32c_u32_step_bitwise.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_varint_write.restype = C.c_int
        L.orc_varint_write.argtypes = [C.c_void_p, sz, C.c_uint32]
        L.orc_varint_read.restype = C.c_int
        L.orc_varint_read.argtypes = [C.c_void_p, sz, u32p, C.POINTER(C.c_int)]
        L.orc_get_uncompressed_length.restype = C.c_int
        L.orc_get_uncompressed_length.argtypes = [C.c_void_p, sz, u32p, u32p]
        L.orc_table_size.restype = C.c_uint32
        L.orc_table_size.argtypes = [C.c_uint32]
        L.orc_hash.restype = C.c_uint32
        L.orc_hash.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
        L.orc_log2_floor.restype = C.c_int
        L.orc_log2_floor.argtypes = [C.c_uint32]
        L.orc_left_shift_overflows.restype = C.c_int
        L.orc_left_shift_overflows.argtypes = [C.c_uint8, C.c_int]
        L.orc_find_match_length.restype = C.c_int
        L.orc_find_match_length.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_compress.restype = C.c_int
        L.orc_compress.argtypes = [C.c_void_p, sz, C.c_void_p, sz, C.c_int, C.POINTER(sz)]
        L.orc_decompress.restype = C.c_int
        L.orc_decompress.argtypes = [C.c_void_p, sz, C.c_void_p, sz, C.POINTER(sz)]
        L.orc_frame_encode.restype = C.c_int
        L.orc_frame_encode.argtypes = [C.c_void_p, sz, C.c_void_p, sz, C.c_int, C.POINTER(sz)]
        L.orc_frame_decode.restype = C.c_int
        L.orc_frame_decode.argtypes = [C.c_void_p, sz, C.c_void_p, sz, C.POINTER(sz)]
        L.orc_frame_decoded_length.restype = C.c_int
        L.orc_frame_decoded_length.argtypes = [C.c_void_p, sz, u64p]
        L.orc_compress_batch.restype = None
        L.orc_compress_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_decompress_batch.restype = None
        L.orc_decompress_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_crc32c_batch.restype = None
        L.orc_crc32c_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def _buf(b):
    """bytes-like -> (ctypes pointer-compatible object, length, keepalive)."""
    a = np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else np.ascontiguousarray(b, dtype=np.uint8)
    return a.ctypes.data if a.size else None, a.size, a


class OracleError(Exception):
    def __init__(self, status: int):
        super().__init__(f"oracle status {status}")
        self.status = status


def max_compressed_length(n: int) -> int:
    return lib().orc_max_compressed_length(n)


def crc32c(data, masked: bool = False) -> int:
    p, n, _k = _buf(data)
    c = lib().orc_crc32c(p, n)
    return lib().orc_crc32c_mask(c) if masked else c


def crc32c_bitwise(data) -> int:
    p, n, _k = _buf(data)
    return lib().orc_crc32c_bitwise(p, n)


def crc32c_mask(x: int) -> int:
    return lib().orc_crc32c_mask(x)


def varint_write(v: int, cap: int = 5) -> bytes:
    out = (C.c_uint8 * 8)()
    n = lib().orc_varint_write(out, cap, v)
    return bytes(out[:n])


def varint_read(data):
    """-> (status, value, bytes_read)"""
    p, n, _k = _buf(data)
    v, br = C.c_uint32(0), C.c_int(0)
    st = lib().orc_varint_read(p, n, C.byref(v), C.byref(br))
    return st, v.value, br.value


def get_uncompressed_length(data) -> int:
    p, n, _k = _buf(data)
    v, hb = C.c_uint32(0), C.c_uint32(0)
    st = lib().orc_get_uncompressed_length(p, n, C.byref(v), C.byref(hb))
    if st != OK:
        raise OracleError(st)
    return v.value


def hash_bytes(b: int, mask: int, variant: int) -> int:
    return lib().orc_hash(b, mask, variant)


def table_size(n: int) -> int:
    return lib().orc_table_size(n)


def find_match_length(s1: bytes, s2: bytes, length: int) -> int:
    """Mirrors the reference test harness (SnappyCompressorTests.cs:82-95): s1 || s2 || zero padding in one array."""
    arr = np.frombuffer(s1 + s2 + b"\0" * max(0, length - len(s2)) + b"\0" * 16, dtype=np.uint8).copy()
    base = arr.ctypes.data
    return lib().orc_find_match_length(base, base + len(s1), base + len(s1) + length)


def compress(data, variant: int = HASH_CRC32C, cap: int | None = None) -> bytes:
    p, n, _k = _buf(data)
    if cap is None:
        cap = max_compressed_length(n)
    out = np.empty(max(cap, 1), dtype=np.uint8)
    w = C.c_size_t(0)
    st = lib().orc_compress(p, n, out.ctypes.data, cap, variant, C.byref(w))
    if st != OK:
        raise OracleError(st)
    return out[: w.value].tobytes()


def decompress(data, cap: int | None = None) -> bytes:
    p, n, _k = _buf(data)
    if cap is None:
        try:
            cap = get_uncompressed_length(data)
        except OracleError:
            cap = 0
    out = np.empty(max(cap, 1), dtype=np.uint8)
    w = C.c_size_t(0)
    st = lib().orc_decompress(p, n, out.ctypes.data, cap, C.byref(w))
    if st != OK:
        raise OracleError(st)
    return out[: w.value].tobytes()


def decompress_status(data, cap: int | None = None) -> int:
    try:
        decompress(data, cap)
        return OK
    except OracleError as e:
        return e.status


def frame_encode(data, variant: int = HASH_CRC32C) -> bytes:
    p, n, _k = _buf(data)
    cap = lib().orc_frame_max_encoded_length(n)
    out = np.empty(cap, dtype=np.uint8)
    w = C.c_size_t(0)
    st = lib().orc_frame_encode(p, n, out.ctypes.data, cap, variant, C.byref(w))
    if st != OK:
        raise OracleError(st)
    return out[: w.value].tobytes()


def frame_decoded_length(data) -> int:
    p, n, _k = _buf(data)
    v = C.c_uint64(0)
    st = lib().orc_frame_decoded_length(p, n, C.byref(v))
    if st != OK:
        raise OracleError(st)
    return v.value


def frame_decode(data) -> bytes:
    p, n, _k = _buf(data)
    cap = frame_decoded_length(data)
    out = np.empty(max(cap, 1), dtype=np.uint8)
    w = C.c_size_t(0)
    st = lib().orc_frame_decode(p, n, out.ctypes.data, cap, C.byref(w))
    if st != OK:
        raise OracleError(st)
    return out[: w.value].tobytes()


def _stripe(nblocks: int, threads: int):
    threads = max(1, min(threads, nblocks))
    edges = np.linspace(0, nblocks, threads + 1).astype(np.int64)
    return [(int(edges[i]), int(edges[i + 1])) for i in range(threads) if edges[i + 1] > edges[i]]


def compress_batch(inp: np.ndarray, in_off: np.ndarray, in_len: np.ndarray, variant: int = HASH_CRC32C, threads: int = 1, out: np.ndarray | None = None):
    """Independent <=64 KiB blocks -> (out, out_off, out_len, status); fixed stride max_compressed_length(65536).
    out: a caller-owned buffer of nb * stride bytes to reuse (timing legs: no allocation, no first-touch page faults inside the call)."""
    nb = len(in_len)
    stride = max_compressed_length(BLOCK_SIZE)
    if out is None:
        out = np.empty(nb * stride, dtype=np.uint8)
    assert out.size >= nb * stride
    out_off = (np.arange(nb, dtype=np.uint64) * np.uint64(stride)).astype(np.uint64)
    out_len = np.zeros(nb, dtype=np.uint32)
    status = np.zeros(nb, dtype=np.int32)
    inp = np.ascontiguousarray(inp, dtype=np.uint8)
    in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
    in_len = np.ascontiguousarray(in_len, dtype=np.uint32)
    L = lib()

    def run(r):
        L.orc_compress_batch(inp.ctypes.data, in_off.ctypes.data, in_len.ctypes.data, r[0], r[1], out.ctypes.data,
                             out_off.ctypes.data, out_len.ctypes.data, status.ctypes.data, variant)

    ranges = _stripe(nb, threads)
    if len(ranges) <= 1:
        for r in ranges:
            run(r)
    else:
        with ThreadPoolExecutor(len(ranges)) as ex:
            list(ex.map(run, ranges))
    return out, out_off, out_len, status


def decompress_batch(inp: np.ndarray, in_off: np.ndarray, in_len: np.ndarray, out_off: np.ndarray, out_cap: np.ndarray,
                     out_size: int, threads: int = 1, out: np.ndarray | None = None):
    """out: a caller-owned buffer of >= out_size bytes to reuse (timing legs); default: a fresh zeroed one."""
    nb = len(in_len)
    if out is None:
        out = np.zeros(max(out_size, 1), dtype=np.uint8)
    assert out.size >= max(out_size, 1)
    out_len = np.zeros(nb, dtype=np.uint32)
    status = np.zeros(nb, dtype=np.int32)
    inp = np.ascontiguousarray(inp, dtype=np.uint8)
    in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
    in_len = np.ascontiguousarray(in_len, dtype=np.uint32)
    out_off = np.ascontiguousarray(out_off, dtype=np.uint64)
    out_cap = np.ascontiguousarray(out_cap, dtype=np.uint32)
    L = lib()

    def run(r):
        L.orc_decompress_batch(inp.ctypes.data, in_off.ctypes.data, in_len.ctypes.data, r[0], r[1], out.ctypes.data,
                               out_off.ctypes.data, out_cap.ctypes.data, out_len.ctypes.data, status.ctypes.data)

    ranges = _stripe(nb, threads)
    if len(ranges) <= 1:
        for r in ranges:
            run(r)
    else:
        with ThreadPoolExecutor(len(ranges)) as ex:
            list(ex.map(run, ranges))
    return out, out_len, status


def crc32c_batch(inp: np.ndarray, in_off: np.ndarray, in_len: np.ndarray, masked: bool = False, threads: int = 1):
    nb = len(in_len)
    out = np.zeros(nb, dtype=np.uint32)
    inp = np.ascontiguousarray(inp, dtype=np.uint8)
    in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
    in_len = np.ascontiguousarray(in_len, dtype=np.uint32)
    L = lib()

    def run(r):
        L.orc_crc32c_batch(inp.ctypes.data, in_off.ctypes.data, in_len.ctypes.data, r[0], r[1], int(masked), out.ctypes.data)

    ranges = _stripe(nb, threads)
    if len(ranges) <= 1:
        for r in ranges:
            run(r)
    else:
        with ThreadPoolExecutor(len(ranges)) as ex:
            list(ex.map(run, ranges))
    return out
