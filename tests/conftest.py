import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def opy():
    from oracle import bn254_py
    return bn254_py


@pytest.fixture(scope="session")
def orc():
    from oracle import cpu
    cpu.build()
    return cpu


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="session")
def ctx(pkg):
    """A live device context.  Fails (does not skip) when the HIP library or the GPU is missing: -m gpu tests
    must never pass on a fallback."""
    c = pkg.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def vectors():
    return json.load(open(os.path.join(GOLDEN, "oracle_vectors.json")))


@pytest.fixture(scope="session")
def refconst():
    return json.load(open(os.path.join(GOLDEN, "reference_constants.json")))


class Conv:
    """hex/int fixtures <-> gnark memory images (uint64 Montgomery limbs)."""

    def __init__(self, o):
        self.o = o

    def fr(self, vals):
        o = self.o
        return np.array([o.limbs(o.to_mont(int(v, 16) if isinstance(v, str) else v, o.R)) for v in vals], dtype=np.uint64).reshape(-1, 4)

    def fr_back(self, arr):
        o = self.o
        return [o.from_mont(o.from_limbs(row), o.R) for row in arr]

    def g1(self, pts):
        o = self.o
        rows = []
        for p in pts:
            if p is None:
                rows.append([0] * 8)
            else:
                x, y = (int(c, 16) if isinstance(c, str) else c for c in p)
                rows.append(o.g1_to_limbs((x, y)))
        return np.array(rows, dtype=np.uint64).reshape(-1, 8)

    def g2(self, pts):
        o = self.o
        rows = []
        for p in pts:
            if p is None:
                rows.append([0] * 16)
            else:
                c = [int(v, 16) if isinstance(v, str) else v for v in p]
                rows.append(o.g2_to_limbs(((c[0], c[1]), (c[2], c[3]))))
        return np.array(rows, dtype=np.uint64).reshape(-1, 16)


@pytest.fixture(scope="session")
def conv(opy):
    return Conv(opy)
