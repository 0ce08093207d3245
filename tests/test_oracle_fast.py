"""bench.py's cpu_baseline times a -DORACLE_FAST build of the oracle (16-byte literal / self-copy moves, the scalar stand-in for
CopyHelpers.cs:64-230's SSSE3 path).  That build is never the checker -- but its results must be the plain oracle's: same status,
same bytes, on the corpus, on truncated and corrupted streams and on overlap-heavy blocks."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle as O
from conftest import CORPUS, ROOT, read_testdata
import datagen


@pytest.fixture(scope="module")
def fast(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("oracle_fast") / "libsnappy_oracle_fast.so")
    subprocess.run(["gcc", "-O2", "-msse4.2", "-std=c11", "-fPIC", "-shared", "-DORACLE_FAST", "-o", so,
                    os.path.join(ROOT, "oracle", "snappy_oracle.c")], check=True)
    L = C.CDLL(so)
    L.orc_decompress.restype = C.c_int
    L.orc_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    return L


def _both(fast, z: bytes, cap: int):
    plain = O.lib()
    res = []
    for L in (plain, fast):
        out = np.full(cap + 64, 0xA5, dtype=np.uint8)               # 64 guard bytes: the fast paths may not write past `cap`
        w = C.c_size_t(0)
        buf = np.frombuffer(z, dtype=np.uint8) if len(z) else np.zeros(1, dtype=np.uint8)
        st = L.orc_decompress(buf.ctypes.data, len(z), out.ctypes.data, cap, C.byref(w))
        assert (out[cap:] == 0xA5).all(), "wrote past the capacity"
        res.append((st, w.value, out[: w.value].tobytes() if st == 0 else b""))
    return res


def test_fast_build_equals_plain_on_corpus_and_low_entropy(fast):
    for name in CORPUS:
        data = read_testdata(name)
        for lo in range(0, len(data), 65536):
            blk = data[lo: lo + 65536]
            z = O.compress(blk)
            a, b = _both(fast, z, len(blk))
            assert a == b and a[0] == 0 and a[2] == blk, name
    for b in range(8):
        blk = datagen.low_entropy_block(b).tobytes()
        z = O.compress(blk)
        a, c = _both(fast, z, len(blk))
        assert a == c and a[2] == blk


def test_fast_build_equals_plain_on_broken_streams(fast):
    rng = np.random.default_rng(4242)
    text = read_testdata("html") + read_testdata("alice29.txt")
    for r in range(400):
        n = int(rng.integers(1, 9000))
        lo = int(rng.integers(0, len(text) - n))
        blk = text[lo: lo + n]
        z = bytearray(O.compress(blk))
        how = int(rng.integers(0, 4))
        if how == 0:
            z = z[: int(rng.integers(0, len(z)))]
        elif how == 1:
            for _ in range(int(rng.integers(1, 4))):
                z[int(rng.integers(0, len(z)))] ^= 1 << int(rng.integers(0, 8))
        elif how == 2:
            z += bytes(rng.integers(0, 256, int(rng.integers(1, 20)), dtype=np.uint8))
        cap = n if rng.integers(0, 4) else max(1, n - int(rng.integers(1, 20)))
        a, b = _both(fast, bytes(z), cap)
        assert a == b, f"round {r} how {how}"
    for name in ("baddata1.snappy", "baddata2.snappy", "baddata3.snappy"):
        a, b = _both(fast, read_testdata(name), 131072)
        assert a == b
