"""ctypes binding of tests/window_model.c -- the executable CPU model of the window compressor's algorithm
(snappier_amd/csrc/compress_win.hip).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "window_model.c")
_SO = os.path.join(_HERE, "libwindow_model.so")


class Stats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in
                "dense_rounds sparse_rounds cuts tokens dense_tokens dense_advance long_resolves events".split()]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
            subprocess.run(["gcc", "-O2", "-std=c11", "-Wall", "-shared", "-fPIC", "-o", _SO, _SRC], check=True)
        L = C.CDLL(_SO)
        L.wm_compress.restype = C.c_size_t
        L.wm_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(Stats)]
        _lib = L
    return _lib


def compress(data: bytes, variant: int, np_: int = 2, cap: int = 32, stats: Stats | None = None) -> bytes:
    out = C.create_string_buffer(len(data) + len(data) // 6 + 64 + 40 * (len(data) // 65536 + 1))
    n = lib().wm_compress(data, len(data), out, variant, np_, cap, C.byref(stats) if stats is not None else None)
    return out.raw[:n]
