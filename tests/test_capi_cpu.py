"""CPU-side checks of the drop-in boundary: libsnappier_hip.so loads, exports every symbol include/snappier_hip.h
declares, its host-only arithmetic matches the reference KATs, and it FAILS LOUDLY without a HIP device
(no CPU fallback).  No compute calls here."""
import ctypes as C
import os

import numpy as np
import pytest

import kats
from conftest import ROOT, read_testdata


def test_library_built_and_exports_every_declared_symbol():
    from snappier_amd import _native as N
    assert os.path.exists(N.LIB_PATH), "run python snappier_amd/build.py"
    declared = N.declared_symbols()
    assert len(declared) >= 20 and "snp_compress_batch" in declared and "snp_decompress_batch" in declared
    L = C.CDLL(N.LIB_PATH)
    for s in declared:
        assert hasattr(L, s), s


def test_every_exported_snp_symbol_is_declared_in_a_header():
    """Nothing is exported behind the headers' back: every snp_* function the .so exports is declared either in the product
    header or in include/snappier_hip_debug.h (test hooks), apart from the launchers the translation units call among themselves."""
    import re
    import subprocess
    from snappier_amd import _native as N
    out = subprocess.run(["nm", "-D", "--defined-only", N.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {m.group(1) for m in re.finditer(r" T (snp_[a-z0-9_]+)$", out, flags=re.M)}
    internal = {e for e in exported if e.startswith(("snp_launch_", "snp_probe_", "snp_compress_lanes_workspace", "snp_compress_win_table_bytes", "snp_frame_scan_workspace", "snp_tag_index_"))}
    dbg = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "snappier_hip_debug.h")).read(), flags=re.S)
    debug_declared = set(re.findall(r"\b(snp_debug_[a-z0-9_]+)\s*\(", dbg))
    assert debug_declared == {e for e in exported if e.startswith("snp_debug_")}, (debug_declared, exported)
    assert exported - internal - debug_declared == set(N.declared_symbols()), sorted(exported - internal - debug_declared - set(N.declared_symbols()))


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "snappier_amd")
    for dirpath, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in text and "from oracle" not in text and "libsnappy_oracle" not in text, f


def test_host_only_arithmetic():
    import snappier_amd as S
    L = S.lib()
    assert L.snp_max_compressed_length(65536) == 76496 and L.snp_max_fragment_compressed_length(65536) == 76491
    assert L.snp_max_compressed_length(0) == 38 and L.snp_max_compressed_length(-1) == -1
    assert L.snp_frame_max_encoded_length(0) == 10 and L.snp_frame_max_encoded_length(65537) == 10 + 16 + 65537
    for value, enc in kats.VARINT:
        if value <= 0x7FFFFFFF or True:
            assert S.Snappy.GetUncompressedLength(enc + b"\xff" * 3) == value
    for enc in kats.VARINT_INCOMPLETE + [kats.VARINT_BAD, b""]:
        with pytest.raises(S.InvalidDataException):
            S.Snappy.GetUncompressedLength(enc)
    for name, declared in (("baddata1.snappy", 128082), ("baddata2.snappy", 128059), ("baddata3.snappy", 130378)):
        assert S.Snappy.GetUncompressedLength(read_testdata(name)) == declared


def test_frame_decoded_length_on_goldens():
    import snappier_amd as S
    for name, n in (("html_x_4.snappy", 409600), ("alice29.snappy", 152089)):
        d = np.frombuffer(read_testdata(name), dtype=np.uint8)
        v = C.c_uint64(0)
        assert S.lib().snp_frame_decoded_length(C.c_void_p(d.ctypes.data), d.size, C.byref(v)) == 0
        assert v.value == n
    bad = np.frombuffer(read_testdata("html_x_4.snappy")[:10] + bytes([0x02, 1, 0, 0, 0]), dtype=np.uint8)
    v = C.c_uint64(0)
    assert S.lib().snp_frame_decoded_length(C.c_void_p(bad.ctypes.data), bad.size, C.byref(v)) == S._native.ERR_CHUNK_TYPE


def test_fails_loudly_without_a_device():
    import torch
    import snappier_amd as S
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(S.InvalidOperationException):
        S.Context()
    with pytest.raises(S.InvalidOperationException):
        S.Snappy.CompressToArray(b"hello")          # no CPU fallback behind the API


def test_status_strings_match_reference_messages():
    import snappier_amd as S
    N = S._native
    assert S.status_string(N.ERR_BAD_OFFSET) == "Invalid copy offset"                 # SnappyDecompressor.cs:600
    assert S.status_string(N.ERR_TOO_LONG) == "Data too long"                         # :572,605
    assert S.status_string(N.ERR_INCOMPLETE) == "Incomplete Snappy block."            # ThrowHelper.cs
    assert S.status_string(N.ERR_BAD_LENGTH) == "Invalid stream length"               # VarIntEncoding.Read.cs:20
    assert S.status_string(N.ERR_CRC_MISMATCH) == "Chunk CRC mismatch."               # SnappyStreamDecompressor.cs:130
    assert S.status_string(N.ERR_OUTPUT_TOO_SMALL) == "Output buffer is too small."   # ThrowHelper.cs:18-19


def test_frame_chunk_cannot_declare_more_than_it_can_produce():
    """A ~20-byte stream whose chunk declares 2 GiB must not size any allocation: no tag expands more than 3 bytes -> 64,
    so the chunk can only end "Incomplete Snappy block." (host header walk, no device needed; the device walk applies the
    same bound -- tests/test_gpu_parity.py)."""
    from snappier_amd import _native as N
    L = C.CDLL(N.LIB_PATH)
    L.snp_frame_decoded_length.restype = C.c_int
    L.snp_frame_decoded_length.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
    head = bytes([0xFF, 0x06, 0x00, 0x00, 0x73, 0x4E, 0x61, 0x50, 0x70, 0x59])
    body = bytes([0xFF, 0xFF, 0xFF, 0xFF, 0x07]) + bytes([0x00, 0x41])            # varint 2^31-1, then a 1-byte literal
    chunk = bytes([0x00]) + (4 + len(body)).to_bytes(3, "little") + bytes(4) + body
    total = C.c_uint64(123)
    assert L.snp_frame_decoded_length(head + chunk, len(head + chunk), C.byref(total)) == 4      # SNP_ERR_INCOMPLETE
    assert total.value == 0
    # a chunk at the bound is still accepted by the walk (it is the decoder's business from there on)
    body = bytes([64]) + bytes([0xFE, 0x01, 0x00])                                 # declares 64, one copy-2 tag
    chunk = bytes([0x00]) + (4 + len(body)).to_bytes(3, "little") + bytes(4) + body
    assert L.snp_frame_decoded_length(head + chunk, len(head + chunk), C.byref(total)) == 0 and total.value == 64
