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
LIB_PATH = os.environ.get("SNAPPIER_HIP_LIB") or os.path.join(HERE, "libsnappier_hip.so")   # override: kernel-variant A/B runs (a LAB=1 build, if knobs are to act)
HEADER_PATH = os.path.join(HERE, "..", "include", "snappier_hip.h")

(OK, ERR_OUTPUT_TOO_SMALL, ERR_BAD_OFFSET, ERR_TOO_LONG, ERR_INCOMPLETE, ERR_BAD_LENGTH, ERR_CRC_MISMATCH,
 ERR_CHUNK_TYPE, ERR_OVERLAP, ERR_BAD_ARG, ERR_DEVICE, ERR_TRUNCATED_STREAM) = range(12)
HASH_CRC32C, HASH_MUL = 0, 1
BLOCK_SIZE = 65536
MAX_BLOCK_COMPRESSED = 76491
# snp_option (include/snappier_hip.h)
(OPT_DECODE_LAYOUT, OPT_SMALL_BLOCK_MAX, OPT_SMALL_BLOCK_MIN_BATCH, OPT_COMPRESS_LAYOUT, OPT_COMPRESS_WINDOW_MAX_BATCH,
 OPT_TABLE_PROBE_TRIES, OPT_TABLE_PROBE_MAX_BYTES, OPT_PARALLEL_DECODE_MIN, OPT_FENCED, OPT_DECODE_LEFTOVERS, OPT_CRC_KERNEL) = range(1, 12)


def declared_symbols() -> list[str]:
    """Every function name declared in include/snappier_hip.h."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(snp_[a-z0-9_]+)\s*\(", text)))


_lib = None
_lab = None
LAB_PATH = os.path.join(HERE, "variants", "libsnappier_hip_lab.so")
# what selects the LAB library (see lib()): every SNAPPIER_HIP_* knob except the library override itself
_NOT_KNOBS = ("SNAPPIER_HIP_LIB",)


def debug_knobs_set() -> bool:
    return any(k.startswith("SNAPPIER_HIP_") and k not in _NOT_KNOBS for k in os.environ)


def lab_path() -> str:
    """snappier_amd/variants/libsnappier_hip_lab.so, built on demand (LAB=1 scripts/build_variant.sh lab: the product sources with
    -DSNAPPIER_HIP_DEBUG_ENV plus the decoder front ends of csrc/lab/)."""
    import subprocess
    script = os.path.join(HERE, "..", "scripts", "build_variant.sh")
    srcs = [os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith((".hip", ".h"))]
    srcs += [os.path.join(HERE, "csrc", "lab", f) for f in os.listdir(os.path.join(HERE, "csrc", "lab"))]
    srcs.append(os.path.join(HERE, "..", "include", "snappier_hip.h"))
    if not os.path.exists(LAB_PATH) or any(os.path.getmtime(f) > os.path.getmtime(LAB_PATH) for f in srcs):
        r = subprocess.run(["bash", script, "lab"], env=dict(os.environ, LAB="1"), capture_output=True, text=True)
        if r.returncode != 0:
            raise ImportError("building the lab library failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])
    return LAB_PATH


def lib() -> C.CDLL:
    """The product library -- or, while any SNAPPIER_HIP_* knob is set in the environment (tests and A/B scripts that vary a kernel
    layout), the LAB library, which is the only one that reads them.  A Context remembers the library it was created from."""
    global _lib, _lab
    if not os.environ.get("SNAPPIER_HIP_LIB") and debug_knobs_set():
        if _lab is None:
            _lab = _load(lab_path())
        return _lab
    if _lib is None:
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
