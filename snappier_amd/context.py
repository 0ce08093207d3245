"""Codec context: one HIP device + stream + reusable HBM scratch (snp_ctx).

Snappy.* in the reference is re-entrant because each call news up a compressor (Snappy.cs:64,174,225); here each
thread gets its own default Context instead.
"""
from __future__ import annotations

import ctypes as C
import threading

from . import _native as N
from .errors import InvalidOperationException


class Context:
    def __init__(self, device: int = 0, hash_variant: int = N.HASH_CRC32C, stream: int | None = None):
        """stream: a hipStream_t handle as an int (torch.cuda.current_stream().cuda_stream; 0 = the default stream),
        or None for a private non-blocking stream owned by the context."""
        self._h = C.c_void_p()
        self.lib = N.lib()          # (the product library -- or the lab one under SNAPPIER_HIP_LAB=1: every call on this context goes to the library it was created from)
        st = self.lib.snp_ctx_create(device, hash_variant, None, C.byref(self._h))
        if st != N.OK:
            self._h = C.c_void_p()
            raise InvalidOperationException(
                f"snp_ctx_create(device={device}) failed: {N.status_string(st)} -- the codec runs on a HIP device only")
        self.device = device
        self.hash_variant = hash_variant
        if stream is not None:
            self.set_stream(stream)

    def set_stream(self, stream: int):
        st = self.lib.snp_ctx_set_stream(self._h, C.c_void_p(stream))
        if st != N.OK:
            raise InvalidOperationException(N.status_string(st))

    @property
    def handle(self):
        return self._h

    def counter(self, which: int) -> int:
        """snp_ctx_counter: 0 = large blocks decoded per fragment, 1 = large blocks that fell back to one wavefront, 6 = tag indexes that needed
        the look-back pass (2..5: the table workspace search)."""
        return int(self.lib.snp_ctx_counter(self._h, which))

    def set_option(self, option: int, value: int):
        """snp_ctx_set_option (N.OPT_*): kernels and memory behaviour only, never results."""
        st = self.lib.snp_ctx_set_option(self._h, option, value)
        if st != N.OK:
            raise ValueError(f"snp_ctx_set_option({option}, {value}): {N.status_string(st)}")

    def reserve_compress(self, nfragments: int):
        """snp_ctx_reserve_compress: build the lane compressor's hash-table workspace for batches of up to `nfragments` fragments now
        (a service's start-up, before its buffers crowd the device) instead of on the first large compress call."""
        st = self.lib.snp_ctx_reserve_compress(self._h, int(nfragments))
        if st != N.OK:
            raise InvalidOperationException(f"snp_ctx_reserve_compress({nfragments}): {N.status_string(st)}: {self.lib.snp_ctx_last_error(self._h).decode()}")

    def get_option(self, option: int) -> int:
        v = C.c_int64(0)
        st = self.lib.snp_ctx_get_option(self._h, option, C.byref(v))
        if st != N.OK:
            raise ValueError(f"snp_ctx_get_option({option}): {N.status_string(st)}")
        return int(v.value)

    def synchronize(self):
        st = self.lib.snp_ctx_synchronize(self._h)
        if st != N.OK:
            raise InvalidOperationException(self.lib.snp_ctx_last_error(self._h).decode())

    def close(self):
        if self._h:
            self.lib.snp_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_tls = threading.local()


def default_context(hash_variant: int | None = None) -> Context:
    key = N.HASH_CRC32C if hash_variant is None else hash_variant
    cache = getattr(_tls, "ctx", None)
    if cache is None:
        cache = _tls.ctx = {}
    key = (key, N.lab_requested())         # (a context belongs to the library it was created from)
    if key not in cache:
        cache[key] = Context(0, key[0])
    return cache[key]


def close_default_contexts() -> None:
    """Closes this thread's cached default contexts (a device's table pool lives as long as its last context: tests of the pool's lifetime
    call this first, so that the order they run in does not matter)."""
    cache = getattr(_tls, "ctx", None)
    if cache:
        for c in cache.values():
            c.close()
        cache.clear()
