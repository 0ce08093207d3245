"""Device-side synthetic workload generators (libsnappier_datagen.so) -- bench / test plumbing, not the codec.

CPU statement of the same arithmetic: tests/datagen.py.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(HERE, "libsnappier_datagen.so")
HTML_SEED = 0x5EED0001
LOWENT_SEED = 0x5EED0003
MIXED_SEED = 0x5EED0005
_lib = None


def _l():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise ImportError(f"{_LIB} is missing: run `python snappier_amd/build.py`")
        _lib = C.CDLL(_LIB)
        _lib.snp_gen_corpus_blocks.restype = C.c_int
        _lib.snp_gen_corpus_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32,
                                               C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
        _lib.snp_gen_low_entropy_blocks.restype = C.c_int
        _lib.snp_gen_low_entropy_blocks.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
    return _lib


def corpus_blocks(files: list[bytes], first_block: int, nblocks: int, seed: int, device, block: int = 65536) -> torch.Tensor:
    """Blocks first_block .. first_block+nblocks of the tiled+mutated corpus (file = block index mod len(files))."""
    dev = torch.device(device)
    cat = np.frombuffer(b"".join(files), dtype=np.uint8)
    lens = np.array([len(f) for f in files], dtype=np.uint32)
    offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.uint64)]).astype(np.uint64)
    d_cat = torch.from_numpy(cat.copy()).to(dev)
    d_off = torch.from_numpy(offs.view(np.int64).copy()).to(dev)
    d_len = torch.from_numpy(lens.view(np.int32).copy()).to(dev)
    out = torch.empty(nblocks * block, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    # the grid dimension is the block count; stay far below the 2^31 limit per launch
    step = 1 << 20
    for s in range(0, nblocks, step):
        k = min(step, nblocks - s)
        rc = _l().snp_gen_corpus_blocks(d_cat.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), len(files), first_block + s, k,
                                        seed, block, out.data_ptr() + s * block, stream)
        if rc:
            raise RuntimeError(f"snp_gen_corpus_blocks failed: hipError {rc}")
    return out


def html_like_blocks(html: bytes, first_block: int, nblocks: int, device, block: int = 65536) -> torch.Tensor:
    return corpus_blocks([html], first_block, nblocks, HTML_SEED, device, block)


def low_entropy_blocks(first_block: int, nblocks: int, device, block: int = 65536) -> torch.Tensor:
    dev = torch.device(device)
    out = torch.empty(nblocks * block, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    rc = _l().snp_gen_low_entropy_blocks(first_block, nblocks, LOWENT_SEED, block, out.data_ptr(), stream)
    if rc:
        raise RuntimeError(f"snp_gen_low_entropy_blocks failed: hipError {rc}")
    return out
