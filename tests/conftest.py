import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (reference build; build container only)")


@pytest.fixture(scope="session")
def po():
    """The CPU oracle (test infrastructure)."""
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    d = os.path.join(ROOT, "tests", "golden")

    class G:
        def __getitem__(self, name):
            return np.load(os.path.join(d, name + ".npz"), allow_pickle=False)
    return G()


@pytest.fixture(scope="session")
def scenes(golden):
    """Scenes as loaded by the REFERENCE loader (golden fixture)."""
    z = golden["scenes"]
    out = {}
    for k in z.files:
        name, field = k.split("__")
        out.setdefault(name, {})[field] = z[k]
    for s in out.values():
        s["depth"] = int(s["depth"])
        s["iterations"] = int(s["iterations"])
    return out
