"""include/snappier.hpp (the C++ mirror of Snappy.*) compiles against the C-ABI and behaves: host-only arithmetic works,
and without a GPU constructing a Context throws instead of falling back to anything."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

SRC = r'''
#include <cstdio>
#include <cstring>
#include <vector>
#include "snappier.hpp"
using namespace Snappier;
int main() {
    if (Snappy::GetMaxCompressedLength(65536) != 76496) return 2;
    const unsigned char v[] = {0x80, 0x80, 0x04, 0xff};
    if (Snappy::GetUncompressedLength(v, 4) != 65536) return 3;
    try { const unsigned char bad[] = {0xff,0xff,0xff,0xff,0xff,0xff}; Snappy::GetUncompressedLength(bad, 6); return 4; }
    catch (const InvalidDataException& e) { if (e.status != SNP_ERR_BAD_LENGTH) return 5; }
    try {
        Context ctx;                                     // needs a HIP device
        std::vector<unsigned char> in(200000);
        for (size_t i = 0; i < in.size(); ++i) in[i] = (unsigned char)((i * 7) % 251 < 40 ? 'a' : (i >> 5));
        auto c = Snappy::CompressToArray(ctx, in.data(), in.size());
        auto d = Snappy::DecompressToArray(ctx, c.data(), c.size());
        if (d != in) return 6;
        size_t w = 0;
        std::vector<unsigned char> small(10);
        if (Snappy::TryCompress(ctx, in.data(), in.size(), small.data(), small.size(), w)) return 7;
        std::printf("gpu roundtrip ok %zu -> %zu\n", in.size(), c.size());
    } catch (const InvalidOperationException& e) {
        std::printf("no device: %s\n", e.what());
        return 10;
    }
    return 0;
}
'''


def _build(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = tmp_path / "t"
    lib_dir = os.path.join(ROOT, "snappier_amd")
    subprocess.run(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", lib_dir,
                    "-l:libsnappier_hip.so", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib",
                    "-Wl,--allow-shlib-undefined"], check=True)
    return str(exe)


def test_cpp_mirror_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 10, (r.returncode, r.stdout, r.stderr)      # Context() threw: no CPU fallback
    assert "no device" in r.stdout


@pytest.mark.gpu
def test_cpp_mirror_roundtrip_on_gpu(tmp_path):
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "gpu roundtrip ok" in r.stdout
