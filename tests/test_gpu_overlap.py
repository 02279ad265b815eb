"""GPU parity, launch plans that overlap or replay: consecutive asynchronous batches on the lanes (csrc/ptmi355.hip:
enqueue_batch_direct), the final-colour stamps they rely on, hipGraph replay -- each equal to the serial plan and to the oracle."""
import hashlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import __graft_entry__ as ge  # noqa: E402,F401
from gpu_common import pt, launch_plan, bits, rel_l2, assert_paths_equal, _resized, _after  # noqa: E402,F401

pytestmark = pytest.mark.gpu


def test_c2_one_iteration_per_call_overlapped(pt, scenes, monkeypatch):
    """C2 at full size through the reference's call pattern, enqueued back to back: every call is ONE k_iteration launch on a
    PARTIAL grid (csrc/pt_h_enqueue.hpp: iter_grid_for) overlapping its neighbours on the lanes.  Image, ray count and per-bounce
    live counts equal those of the same calls waited for one by one (whole grid, in-launch finalGather), which
    test_c2_full_iteration pins to the oracle and the golden image."""
    s = scenes["cornell"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    n = scene.resolution[0] * scene.resolution[1]

    def run(overlapped, env=()):
        for k, v in env:
            monkeypatch.setenv(k, v)
        pt.pathtraceInit(scene, flags=pt.PT_COMPACT, max_batch=1)
        for it in range(1, 14):
            if overlapped:
                pt.trace_batch_async(it, 1)
            else:
                pt.pathtrace(None, 0, it)
        pt.synchronize()
        out = (pt.get_image(n).tobytes(), tuple(int(v) for v in pt.counters()))
        pt.pathtraceFree()
        for k, _ in env:
            monkeypatch.delenv(k)
        return out

    serial = run(False)
    assert run(True) == serial
    if pt.has_experiments():          # (grid-size experiments: a -DPT_EXPERIMENTS build only)
        assert run(True, (("PTMI355_ITER_TPW", "1"), ("PTMI355_ITER_WGS_ALL", "2"))) == serial      # a quarter of a workgroup per CU
        assert run(True, (("PTMI355_ITER_TPW", "0"),)) == serial                                     # the whole grid


def test_overlapped_small_batches(pt, po, scenes, monkeypatch):
    """Consecutive small batches whose caller does not wait (pt_trace_batch_async, PT_ASYNC_IMAGE) overlap on lanes
    that share two launch streams (csrc/pt_h_enqueue.hpp: enqueue_batch_direct).  The image after every call, the ray counters and the
    per-bounce statistics equal the serial plan's and the oracle's: batch sizes mixed with larger (serial) batches,
    the camera moved and the trace depth changed in between, synchronous calls in between, a second session."""
    s = scenes["cornell_64"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    n = scene.resolution[0] * scene.resolution[1]
    cam2 = scene.camera.copy()
    cam2["position"][0][0] += 0.75
    plan = [("a", 1, 1), ("a", 2, 1), ("a", 3, 2), ("a", 5, 1), ("a", 6, 1), ("a", 7, 1), ("s", 8, 1), ("a", 9, 1), ("a", 10, 3),
            ("cam", cam2, s["depth"] - 3), ("a", 13, 1), ("a", 14, 1), ("a", 15, 16), ("a", 31, 1), ("a", 32, 1), ("cam", scene.camera.copy(), s["depth"]),
            ("a", 33, 1), ("a", 34, 2), ("a", 36, 1)]

    def run(overlap, serial0=None):
        monkeypatch.setenv("PTMI355_OVERLAP", str(overlap))
        if serial0 is None:
            monkeypatch.delenv("PTMI355_FIN_SERIAL", raising=False)
        else:
            monkeypatch.setenv("PTMI355_FIN_SERIAL", serial0)     # the final-colour stamp wraps in the middle of the plan
        out = []
        for session in range(2):
            pt.pathtraceInit(scene, flags=pt.PT_COMPACT, max_batch=16)
            for step in plan:
                if step[0] == "cam":
                    pt.set_camera(step[1], step[2])
                elif step[0] == "s":
                    out.append(pt.pathtrace(None, 0, step[1]).tobytes())
                else:
                    pt.trace_batch_async(step[1], step[2])
            pt.synchronize()
            out.append(pt.get_image(n).tobytes())
            out.append(tuple(int(v) for v in pt.counters()))
            # single calls, each waited for, between overlapped ones
            pt.trace_batch_async(40, 1)
            pt.synchronize()
            img = pt.get_image(n).copy()
            pt.trace_batch_async(41, 1)
            pt.trace_batch_async(42, 1)
            out.append(pt.get_image(n).tobytes())
            out.append(img.tobytes())
            pt.pathtraceFree()
        return out

    serial, overlapped = run(0), run(1)
    assert serial == overlapped
    assert run(2) == serial and run(4) == serial                  # two / four lanes
    assert run(3) == serial and run(5) == serial and run(8) == serial     # lanes that do not divide the two streams evenly
    if pt.has_experiments():          # a -DPT_EXPERIMENTS build: the stamp's wrap, stream layouts, grid sizes, stream priority
        assert run(3, "0xfffffff8") == serial
        # the lanes share two launch streams by default; one stream for all, one per lane, lanes that do not divide evenly, and
        # k_iteration's grid under the lanes (whole grid / a tile per wave) change nothing either
        for lanes, streams, tpw, wgs in ((3, 1, "0", "15"), (6, 6, "8", "15"), (5, 3, "1", "4"), (8, 2, "2", "40")):
            monkeypatch.setenv("PTMI355_LANE_STREAMS", str(streams))
            monkeypatch.setenv("PTMI355_ITER_TPW", tpw)
            monkeypatch.setenv("PTMI355_ITER_WGS_ALL", wgs)
            assert run(lanes) == serial, (lanes, streams, tpw, wgs)
        for k in ("PTMI355_LANE_STREAMS", "PTMI355_ITER_TPW", "PTMI355_ITER_WGS_ALL"):
            monkeypatch.delenv(k)
        monkeypatch.setenv("PTMI355_MAIN_PRIO", "0")                      # the library's own launch stream at default priority
        assert run(4) == serial
        monkeypatch.delenv("PTMI355_MAIN_PRIO")
    monkeypatch.setenv("PTMI355_OVERLAP_GB", "0.0001")                # the lanes' buffers do not fit the budget: the launch stream alone
    assert run(4) == serial
    monkeypatch.delenv("PTMI355_OVERLAP_GB")
    assert serial[:len(serial) // 2] == serial[len(serial) // 2:]
    # and the oracle: iterations 1..7 with the first camera
    tr = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"])
    for it in range(1, 9):
        tr.iterate(it)
    assert tr.image.tobytes() == overlapped[0]


@pytest.mark.parametrize("variant", ["jitter_lens", "sort", "no_compaction", "glass_sorted"])
def test_overlapped_batches_other_pipelines(pt, scenes, monkeypatch, variant):
    """The lanes under the other fused pipelines: stochastic antialiasing + thin lens (no bounce-0 masks, the lens set
    between batches), the fused material sort (pools K times as long per lane), no compaction, the glass scene sorted.
    Overlapped == one stream, call for call."""
    s = scenes["cornell_glass_64" if variant == "glass_sorted" else "cornell_64"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    n = scene.resolution[0] * scene.resolution[1]
    flags = {"jitter_lens": pt.PT_COMPACT | pt.PT_AA_JITTER, "sort": pt.PT_COMPACT | pt.PT_SORT_MATERIAL,
             "no_compaction": 0, "glass_sorted": pt.PT_COMPACT | pt.PT_SORT_MATERIAL}[variant]

    def run(overlap):
        monkeypatch.setenv("PTMI355_OVERLAP", str(overlap))
        pt.pathtraceInit(scene, flags=flags, max_batch=4)
        out = []
        it = 1
        for k, cnt in enumerate((1, 1, 2, 1, 4, 1, 1, 3, 1, 1)):
            if variant == "jitter_lens" and k in (3, 7):
                pt.set_lens(0.25 if k == 3 else 0.0, 9.0 if k == 3 else 0.0)
            pt.trace_batch_async(it, cnt)
            it += cnt
            if k in (4, 9):
                out.append(pt.get_image(n).tobytes())
        out.append(tuple(int(v) for v in pt.counters()))
        pt.pathtraceFree()
        return out

    assert run(0) == run(4) == run(2)


def test_overlapped_async_image(pt, scenes, monkeypatch):
    """PT_ASYNC_IMAGE + one iteration per call (the shim's asynchronous variant) interleaved with overlapped batches:
    every buffer still holds exactly the sum after its own call."""
    s = scenes["cornell_64"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    n = scene.resolution[0] * scene.resolution[1]
    monkeypatch.setenv("PTMI355_OVERLAP", "0")
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT)
    sums = [pt.pathtrace(None, 0, it).copy() for it in range(1, 12)]
    pt.pathtraceFree()
    monkeypatch.setenv("PTMI355_OVERLAP", "1")
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_ASYNC_IMAGE)
    bufs = [np.zeros((n, 3), dtype=np.float32) for _ in range(3)]
    L = pt.library()
    last = None                                   # (buffer, iteration) of the previous call that took a host image
    calls = 0
    for it in range(1, 12):
        if it in (3, 4, 7, 10):
            pt.trace_batch_async(it, 1)           # overlapped on the lanes, between the image calls
            continue
        buf = bufs[calls % 3]
        calls += 1
        assert L.pt_trace(None, 0, it, buf.ctypes.data) == 0
        if last is not None:                      # the buffer of the previous image call is complete when this one returns
            assert last[0].tobytes() == sums[last[1] - 1].tobytes()
        last = (buf, it)
    pt.synchronize()
    assert last[0].tobytes() == sums[last[1] - 1].tobytes()
    assert pt.get_image(n).tobytes() == sums[10].tobytes()
    assert pt.counters()[2] == 11
    pt.pathtraceFree()


@pytest.mark.parametrize("graph", [False, True])
def test_final_colour_stamps(pt, po, scenes, monkeypatch, graph):
    """(Run with direct launches and under hipGraph replay, PTMI355_GRAPH=1: the stamp then travels through
    Control::keep[0] because kernel arguments are frozen at capture.)
    Paths that end with colour 0 write nothing; k_gather tells this batch's entries from stale ones by the batch's
    stamp (a per-session serial number in the entry's fourth component).  The same iteration traced again after
    clear_image, batches of different sizes over the same entries, and the serial's wrap-around at 2^32 (the buffer
    is cleared and the serial restarts) all give the oracle's sums."""
    s = scenes["cornell_64"]
    n = 64 * 64
    if graph:
        monkeypatch.setenv("PTMI355_GRAPH", "1")
    for start in (None, "0xfffffffd") if pt.has_experiments() else (None,):    # the second run wraps after three batches (test hook of a -DPT_EXPERIMENTS build)
        if start:
            monkeypatch.setenv("PTMI355_FIN_SERIAL", start)
        scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
        pt.pathtraceInit(scene, flags=pt.PT_COMPACT, max_batch=4)
        ref = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"], flags=po.F_COMPACT, trig=po.TRIG_SHARED)
        img = np.zeros((n, 3), dtype=np.float32)
        for iter0, count in ((1, 4), (5, 1), (6, 3), (9, 4), (13, 2), (15, 1)):
            pt.trace_batch(iter0, count, img)
            for it in range(iter0, iter0 + count):
                ref.iterate(it)
            assert img.tobytes() == ref.image.tobytes(), (start, iter0)
        pt.clear_image()
        ref.image[:] = 0
        pt.trace_batch(1, 4, img)                          # the same iterations again: new stamps, same colours
        for it in range(1, 5):
            ref.iterate(it)
        assert img.tobytes() == ref.image.tobytes()
        pt.pathtraceFree()
    monkeypatch.delenv("PTMI355_FIN_SERIAL", raising=False)


@pytest.mark.parametrize("sort", [False, True])
def test_graph_replay_equals_direct_launches(pt, scenes, monkeypatch, sort):
    """PTMI355_GRAPH=1: a batch captured once and replayed with hipGraphLaunch (iteration number through
    Control::iter0) gives the same image as direct launches, across batch sizes and a camera change -- fused, and with
    the material sort (whose bounce-0 kernels generate the camera rays themselves)."""
    s = scenes["cornell_glass_64"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    n = scene.resolution[0] * scene.resolution[1]

    def run():
        img = np.zeros((n, 3), dtype=np.float32)
        pt.pathtraceInit(scene, flags=pt.PT_COMPACT | (pt.PT_SORT_MATERIAL if sort else 0), max_batch=4)
        for it in (1, 2, 3):
            pt.pathtrace(None, 0, it)                 # batch size 1, three replays
        pt.trace_batch(4, 4, img)                     # batch size 4
        pt.trace_batch(8, 4, img)
        pt.trace_batch(12, 3, img)                    # a third size
        rays = pt.get_stats().total_rays
        cam = scene.camera.copy()
        cam["position"][0][0] += 0.5                  # frozen launch arguments change: graphs are re-captured
        pt.set_camera(cam, s["depth"])
        pt.trace_batch(15, 4, img)
        pt.pathtraceFree()
        return img, rays

    monkeypatch.delenv("PTMI355_GRAPH", raising=False)
    direct = run()
    monkeypatch.setenv("PTMI355_GRAPH", "1")
    replay = run()
    assert direct[1] == replay[1]
    assert direct[0].tobytes() == replay[0].tobytes()
