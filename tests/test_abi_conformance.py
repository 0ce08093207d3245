"""The drop-in boundary from the foreign side: tests/abi/abi_conformance.c (plain C, dlopen) makes the calls the C# shim
csharp/Snappier.Gpu makes -- same symbols, argument widths, enum values, call sequences and the status codes it branches
on.  Host-only part here on CPU; the full sequences with -m gpu.  Also checks that every DllImport in NativeMethods.cs names
a function the header declares (and vice versa)."""
import os
import re
import subprocess

import pytest
import torch

from conftest import ROOT, TESTDATA

SRC = os.path.join(ROOT, "tests", "abi", "abi_conformance.c")
EXE = os.path.join(ROOT, "tests", "abi", "abi_conformance")
LIB = os.path.join(ROOT, "snappier_amd", "libsnappier_hip.so")


def build():
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(SRC), os.path.getmtime(os.path.join(ROOT, "include", "snappier_hip.h"))):
        subprocess.run(["gcc", "-O1", "-std=c11", "-Wall", "-Wextra", "-o", EXE, SRC, "-ldl", "-lpthread"], check=True)
    return EXE


def test_dllimports_match_the_header():
    from snappier_amd import _native as N
    cs = open(os.path.join(ROOT, "csharp", "Snappier.Gpu", "NativeMethods.cs")).read()
    imports = set(re.findall(r"static extern [A-Za-z]+\*? (snp_[a-z0-9_]+)\(", cs))
    declared = set(N.declared_symbols())
    assert imports == declared, (sorted(imports - declared), sorted(declared - imports))
    c_names = set(re.findall(r'"(snp_[a-z0-9_]+)"', open(SRC).read()))
    assert declared <= c_names


def test_abi_conformance_host_part():
    r = subprocess.run([build(), LIB, "host"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "conforming" in r.stdout and "FAIL" not in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_abi_conformance_device_part():
    assert torch.cuda.is_available()
    r = subprocess.run([build(), LIB, "device", os.path.join(TESTDATA, "lcet10.txt")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "conforming (device part)" in r.stdout and "FAIL" not in r.stdout, r.stdout + r.stderr
    assert "same bytes as one context" in r.stdout, r.stdout        # the device-list sequence (MultiDeviceChunkCodec) ran
