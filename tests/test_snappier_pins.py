"""Pins of the oracle's COMPRESSOR bytes against real Snappier, consumed when a maintainer has produced them.

The reference is C# and this image has no .NET runtime, so what real Snappier emits with its CRC-32C TableEntry hash
(HashTable.cs:109-117 -- the variant every x64 / ARM64 .NET 8 host uses, and the one bench.py runs) cannot be generated
here: DESIGN.md 2 calls those bytes "parity unpinned".  csharp/PinVectors is the 60-line console project that emits them
(length + SHA-256 of Snappy.CompressToArray for html[0:65536], html[0:102400], every 64 KiB corpus window, every corpus file):

    dotnet run -c Release --project csharp/PinVectors -- tests/golden/testdata > tests/golden/snappier_pins.json
    DOTNET_EnableHWIntrinsic=0 dotnet run -c Release --project csharp/PinVectors -- tests/golden/testdata > tests/golden/snappier_pins_nointrinsics.json

With either file present this test checks EVERY vector against oracle/snappy_oracle.c; without them it skips, loudly.
The model's own known answers are asserted either way, so that a pin file that disagrees points at the model."""
import hashlib
import json
import os

import pytest

import oracle as O
from conftest import GOLDEN, read_testdata

PIN_FILES = ["snappier_pins.json", "snappier_pins_nointrinsics.json"]
KNOWN = {   # (hash, name, length) -> (compressed length, sha256 prefix) from the probe model
    ("crc32c", "html", 65536): (16446, "822945612f80e8d49f087640acbe415b9b11c59f686e17b8bfe833fe4768251e"),
    ("mul", "html", 65536): (16533, "2f8a1e2979f6b2cb256046dab000c012c0bbca0846be628a26911220beac16f4"),
    ("crc32c", "html", 102400): (22774, "82d47590"),
    ("mul", "html", 102400): (22843, "c7c94425"),
}


def _variant(name: str) -> int:
    return O.HASH_CRC32C if name == "crc32c" else O.HASH_MUL


def test_model_known_answers_hold():
    html = read_testdata("html")
    for (h, name, n), (clen, sha) in KNOWN.items():
        z = O.compress(html[:n], _variant(h))
        assert len(z) == clen and hashlib.sha256(z).hexdigest().startswith(sha), (h, n, len(z))


@pytest.mark.parametrize("pin_file", PIN_FILES)
def test_oracle_compressor_bytes_equal_real_snappier(pin_file):
    path = os.path.join(GOLDEN, pin_file)
    if not os.path.exists(path):
        pytest.skip(f"PARITY UNPINNED: tests/golden/{pin_file} is absent -- no .NET host has run csharp/PinVectors yet; "
                    "the crc32c-hash compressor bytes are checked against the oracle and an independent Python model only")
    doc = json.load(open(path))
    variant = _variant(doc["hash"])
    assert doc["vectors"], "empty pin file"
    checked = 0
    for v in doc["vectors"]:
        data = read_testdata(v["name"])[v["offset"]: v["offset"] + v["length"]]
        assert len(data) == v["length"], v
        z = O.compress(data, variant)
        assert len(z) == v["compressed_length"], (doc["hash"], v, len(z))
        assert hashlib.sha256(z).hexdigest() == v["sha256"], (doc["hash"], v)
        checked += 1
    assert checked >= 30
