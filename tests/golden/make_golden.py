#!/usr/bin/env python3
"""Generate tests/golden/libsnappy_mul_goldens.json with Google libsnappy 1.1.8 (BUILD CONTAINER ONLY).

libsnappy 1.1.8 (/opt/conda/lib/libsnappy.so.1, C API snappy-c.h) is an independent implementation of the block
format.  SURVEY.md section 8(c): for fragments whose hash table has the full 16384 entries (fragment >= 16384 B)
its output is byte-identical to Snappier's multiplicative-hash path (HashTable.cs:121-122), so it mass-produces
goldens for hash_variant = MUL at 64 KiB; it is also an independent DECODER for any compressor output.

The JSON holds, for every 65536-byte window of every corpus file in tests/golden/testdata, the compressed length
and SHA-256 of libsnappy's output.  Only the vectors are committed; libsnappy is not needed on the GPU box.

    python tests/golden/make_golden.py
"""
import ctypes as C
import hashlib
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
CORPUS = ["alice29.txt", "asyoulik.txt", "fireworks.jpeg", "geo.protodata", "html", "kppkn.gtb", "lcet10.txt",
          "paper-100k.pdf", "plrabn12.txt", "urls.10K"]


def main():
    lib = C.CDLL("/opt/conda/lib/libsnappy.so.1")
    lib.snappy_max_compressed_length.restype = C.c_size_t
    lib.snappy_max_compressed_length.argtypes = [C.c_size_t]
    lib.snappy_compress.restype = C.c_int
    lib.snappy_compress.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_size_t)]
    out = {"generator": "libsnappy 1.1.8 snappy_compress", "window": 65536, "files": {}}
    for name in CORPUS:
        data = open(os.path.join(HERE, "testdata", name), "rb").read()
        rows = []
        for start in range(0, len(data), 65536):
            w = data[start:start + 65536]
            if len(w) < 16384:      # smaller tables hash differently in libsnappy 1.1.8
                continue
            cap = lib.snappy_max_compressed_length(len(w))
            buf = C.create_string_buffer(cap)
            n = C.c_size_t(cap)
            assert lib.snappy_compress(w, len(w), buf, C.byref(n)) == 0
            rows.append({"start": start, "len": len(w), "clen": n.value,
                         "sha256": hashlib.sha256(buf.raw[:n.value]).hexdigest()})
        out["files"][name] = rows
    # whole-file (multi-fragment) goldens
    whole = {}
    for name in CORPUS:
        data = open(os.path.join(HERE, "testdata", name), "rb").read()
        if len(data) % 65536 and len(data) % 65536 < 16384:
            continue
        cap = lib.snappy_max_compressed_length(len(data))
        buf = C.create_string_buffer(cap)
        n = C.c_size_t(cap)
        assert lib.snappy_compress(data, len(data), buf, C.byref(n)) == 0
        whole[name] = {"len": len(data), "clen": n.value, "sha256": hashlib.sha256(buf.raw[:n.value]).hexdigest()}
    out["whole_files"] = whole
    with open(os.path.join(HERE, "libsnappy_mul_goldens.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("windows:", sum(len(v) for v in out["files"].values()), "whole files:", len(whole))


if __name__ == "__main__":
    main()
