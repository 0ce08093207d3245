import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
TESTDATA = os.path.join(GOLDEN, "testdata")

CORPUS = ["alice29.txt", "asyoulik.txt", "fireworks.jpeg", "geo.protodata", "html", "html_x_4", "kppkn.gtb",
          "lcet10.txt", "paper-100k.pdf", "plrabn12.txt", "urls.10K"]   # SnappyTests.cs:8-19


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def read_testdata(name: str) -> bytes:
    if name == "html_x_4":      # identical to html repeated four times (checked when the fixtures were harvested)
        return read_testdata("html") * 4
    with open(os.path.join(TESTDATA, name), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def testdata():
    return read_testdata
