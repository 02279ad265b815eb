"""What the -m gpu parity modules share: the library fixture, the two launch plans every test runs under, bit-level
comparisons.  (Imported by name into each module: the autouse `launch_plan` fixture then applies there.)"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

@pytest.fixture(scope="module")
def pt():
    p = ge.load_package()
    p.library()
    yield p
    p.pathtraceFree()


@pytest.fixture(autouse=True, params=["one launch per bounce", "small batches in one launch"])
def launch_plan(request, monkeypatch):
    """Every test runs under both launch plans: a kernel per bounce for every batch (PTMI355_WHOLE_MAX=0), and
    the default, where batches of up to 3 M paths run all their bounces in one launch (k_iteration)."""
    if request.param == "one launch per bounce":
        monkeypatch.setenv("PTMI355_WHOLE_MAX", "0")
    else:
        monkeypatch.delenv("PTMI355_WHOLE_MAX", raising=False)
    return request.param


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def rel_l2(a, b):
    return float(np.sqrt(((a.astype(np.float64) - b) ** 2).sum()) / max(1e-30, np.sqrt((b.astype(np.float64) ** 2).sum())))


def assert_paths_equal(got, want, n):
    for f in ("origin", "direction", "color"):
        assert (bits(got[f][:n]) == bits(want[f][:n])).all(), f
    assert (got["pixelIndex"][:n] == want["pixelIndex"][:n]).all()
    assert (got["remainingBounces"][:n] == want["remainingBounces"][:n]).all()


def _resized(cam, w, h):
    """The reference camera at another resolution: pixelLength follows scene.cpp:131-135 (2 * tan(fov) / resolution)."""
    c = cam.copy()
    c["resolution"][0] = (w, h)
    yscaled = np.tan(np.float32(c["fov"][0][1]) * np.float32(np.pi / 180))
    xscaled = np.float32(yscaled * np.float32(w) / np.float32(h))
    c["pixelLength"][0] = (np.float32(2 * xscaled / np.float32(w)), np.float32(2 * yscaled / np.float32(h)))
    return c


def _after(snaps, d, ref):
    """Oracle path array after bounce d (snapshots hold copies made in the callback,
    which runs after shade + compaction of that bounce)."""
    return snaps[d]["paths"]
