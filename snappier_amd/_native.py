"""ctypes binding of libsnappier_hip.so -- the C-ABI declared in include/snappier_hip.h.

The library is the product: if it is missing or a symbol is absent this module raises, there is no fallback.
torch is imported first on purpose: torch bundles its own libamdhip64.so.7; loading it before our library makes
the dynamic loader resolve our DT_NEEDED libamdhip64.so.7 to that same object, so torch tensors and our kernels
share one HIP runtime (one context, one set of streams).
"""
from __future__ import annotations

import ctypes as C
import os
import re

import torch  # noqa: F401  (must precede CDLL -- see module docstring)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SNAPPIER_HIP_LIB") or os.path.join(HERE, "libsnappier_hip.so")   # override: kernel-variant A/B runs (scripts/build_variant.sh)
HEADER_PATH = os.path.join(HERE, "..", "include", "snappier_hip.h")

(OK, ERR_OUTPUT_TOO_SMALL, ERR_BAD_OFFSET, ERR_TOO_LONG, ERR_INCOMPLETE, ERR_BAD_LENGTH, ERR_CRC_MISMATCH,
 ERR_CHUNK_TYPE, ERR_OVERLAP, ERR_BAD_ARG, ERR_DEVICE, ERR_TRUNCATED_STREAM) = range(12)
HASH_CRC32C, HASH_MUL = 0, 1
BLOCK_SIZE = 65536
MAX_BLOCK_COMPRESSED = 76491
# snp_option (include/snappier_hip.h)
(OPT_DECODE_LAYOUT, OPT_SMALL_BLOCK_MAX, OPT_SMALL_BLOCK_MIN_BATCH, OPT_COMPRESS_LAYOUT, OPT_COMPRESS_WINDOW_MAX_BATCH,
 OPT_TABLE_PROBE_TRIES, OPT_TABLE_PROBE_MAX_BYTES, OPT_PARALLEL_DECODE_MIN, OPT_FENCED, OPT_DECODE_LEFTOVERS, OPT_CRC_KERNEL,
 OPT_COMPRESS_WINDOW_POSITIONS, OPT_COMPRESS_WINDOW_GLOBAL_MIN_BATCH, OPT_COMPRESS_LANE_STORES, OPT_COMPRESS_LANE_PROBES,
 OPT_COMPRESS_LANES_PER_WAVEFRONT, OPT_COMPRESS_SLICE, OPT_COMPRESS_SMALL_INPUT_LDS, OPT_COMPRESS_SMALL_INPUT_LANES, OPT_FRAME_SCAN,
 OPT_DECODE_LDS_THROTTLE, OPT_COMPRESS_WINDOW_GLOBAL_SLOTS, OPT_COMPRESS_WINDOW_DUAL_MIN_BATCH) = range(1, 24)
OPT_CRC_TABLE_FREE = OPT_CRC_KERNEL     # deprecated name (rounds 1-4), same number
# SNP_OPT_DECODE_LAYOUT / SNP_OPT_COMPRESS_LAYOUT values
DECODE_AUTO, DECODE_WAVE_ONLY, DECODE_SMALL_LANES, DECODE_SMALL_TEAM4, DECODE_SMALL_TEAM8, DECODE_SMALL_TEAM16, DECODE_SERIAL = range(7)
COMPRESS_AUTO, COMPRESS_LANES, COMPRESS_WINDOW_LDS, COMPRESS_WINDOW_GLOBAL, COMPRESS_WINDOW_DUAL = 0, 2, 3, 4, 5


def declared_symbols() -> list[str]:
    """Every function name declared in include/snappier_hip.h."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(snp_[a-z0-9_]+)\s*\(", text)))


_lib = None
_lab = None
LAB_PATH = os.path.join(HERE, "variants", "libsnappier_hip_lab.so")


def lab_requested() -> bool:
    """SNAPPIER_HIP_LAB=1: the explicit opt-in of the A/B scripts (scripts/) to the LAB library -- the product sources built with
    -DSNAPPIER_HIP_DEBUG_ENV plus the round-4 decoder front ends of scripts/lab/, the only build in which the SNAPPIER_HIP_* knobs act."""
    return os.environ.get("SNAPPIER_HIP_LAB", "") == "1"


def lab_path() -> str:
    """snappier_amd/variants/libsnappier_hip_lab.so -- built by `LAB=1 bash scripts/build_variant.sh lab`, never implicitly (no compiler runs at import)."""
    if not os.path.exists(LAB_PATH):
        raise ImportError(f"{LAB_PATH} is missing: SNAPPIER_HIP_LAB=1 asks for the lab library; build it with `LAB=1 bash scripts/build_variant.sh lab`")
    return LAB_PATH


def lib() -> C.CDLL:
    """The product library (libsnappier_hip.so, or the file SNAPPIER_HIP_LIB names).  It reads no environment: kernels and layouts are chosen per
    context through snp_ctx_set_option (Context.set_option).  Only with SNAPPIER_HIP_LAB=1 -- the A/B scripts -- is the LAB library loaded
    instead; a stray SNAPPIER_HIP_* variable without it is ignored, with a warning.  A Context remembers the library it was created from."""
    global _lib, _lab
    if not os.environ.get("SNAPPIER_HIP_LIB") and lab_requested():
        if _lab is None:
            _lab = _load(lab_path())
        return _lab
    if _lib is None:
        stray = sorted(k for k in os.environ if k.startswith("SNAPPIER_HIP_") and k not in ("SNAPPIER_HIP_LIB", "SNAPPIER_HIP_LAB"))
        if stray and not os.environ.get("SNAPPIER_HIP_LIB"):
            import warnings
            warnings.warn(f"{', '.join(stray)} ignored: libsnappier_hip.so reads no environment (use Context.set_option, or SNAPPIER_HIP_LAB=1 for the lab build)")
        _lib = _load(LIB_PATH)
    return _lib


def _load(path: str) -> C.CDLL:
    LIB_PATH = path
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python snappier_amd/build.py` "
                          "(there is no CPU fallback for the codec)")
    L = C.CDLL(LIB_PATH)
    # every symbol the header declares must be exported; an installed copy without the repo's include/ directory
    # falls back to the signatures bound below (a missing one still raises AttributeError there)
    missing = [s for s in declared_symbols() if not hasattr(L, s)] if os.path.exists(HEADER_PATH) else []
    if missing:
        raise ImportError(f"libsnappier_hip.so does not export: {missing}")
    vp, sz, u32, u64, i32, i64 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_int, C.c_int64
    szp = C.POINTER(sz)
    sig = {
        "snp_ctx_create": (i32, [i32, i32, vp, C.POINTER(vp)]),
        "snp_ctx_destroy": (None, [vp]),
        "snp_ctx_set_stream": (i32, [vp, vp]),
        "snp_ctx_last_error": (C.c_char_p, [vp]),
        "snp_ctx_synchronize": (i32, [vp]),
        "snp_ctx_counter": (u64, [vp, i32]),
        "snp_ctx_set_option": (i32, [vp, i32, i64]),
        "snp_ctx_get_option": (i32, [vp, i32, C.POINTER(i64)]),
        "snp_ctx_reserve_compress": (i32, [vp, u32]),
        "snp_status_string": (C.c_char_p, [i32]),
        "snp_version": (C.c_char_p, []),
        "snp_max_compressed_length": (i64, [i64]),
        "snp_max_fragment_compressed_length": (i64, [i64]),
        "snp_get_uncompressed_length": (i32, [vp, sz, C.POINTER(u32), C.POINTER(u32)]),
        "snp_try_compress": (i32, [vp, vp, sz, vp, sz, szp]),
        "snp_try_decompress": (i32, [vp, vp, sz, vp, sz, szp]),
        "snp_try_compress_segments": (i32, [vp, C.POINTER(vp), szp, u32, vp, sz, szp]),
        "snp_try_decompress_segments": (i32, [vp, C.POINTER(vp), szp, u32, vp, sz, szp]),
        "snp_crc32c": (i32, [vp, vp, sz, i32, C.POINTER(u32)]),
        "snp_frame_max_encoded_length": (i64, [i64]),
        "snp_frame_encode": (i32, [vp, vp, sz, vp, sz, szp]),
        "snp_frame_decoded_length": (i32, [vp, sz, C.POINTER(u64)]),
        "snp_frame_decode": (i32, [vp, vp, sz, vp, sz, szp]),
        "snp_compress_batch": (i32, [vp, vp, vp, vp, u32, vp, vp, vp, vp]),
        "snp_decompress_batch": (i32, [vp, vp, vp, vp, u32, vp, vp, vp, vp, vp]),
        "snp_crc32c_batch": (i32, [vp, vp, vp, vp, u32, i32, vp]),
        "snp_concat_batch": (i32, [vp, vp, vp, vp, u32, vp, vp]),
        "snp_frame_encode_workspace": (u64, [u64]),
        "snp_frame_encode_device": (i32, [vp, vp, u64, vp, u64, vp, vp]),
        "snp_frame_decode_chunks_device": (i32, [vp, vp, vp, vp, vp, vp, u32, vp, vp, vp, vp, vp]),
        "snp_frame_decode_workspace": (u64, [u32]),
        "snp_frame_decode_device": (i32, [vp, vp, u64, vp, u64, u32, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return L


def status_string(st: int) -> str:
    return lib().snp_status_string(st).decode()
