"""GPU parity, PT_LOOKAHEAD: pt_trace traces windows of iterations ahead of its caller (include/ptmi355.h; csrc/pt_h_api.hpp:
la_trace) and every call only gathers its own sample.  What the reference's host sees must not change: state.image, the
device's accumulation buffer and the PBO after EVERY call equal the oracle's running sum through that iteration, bit for
bit -- across window boundaries, a skipped iteration number, a camera move, a traceDepth change, a second host buffer,
batches in between and a free / re-init (src/main.cpp:102-140 does all of these) -- under both launch plans."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import __graft_entry__ as ge  # noqa: E402,F401
from gpu_common import pt, launch_plan, bits, _resized  # noqa: E402,F401

pytestmark = pytest.mark.gpu

K = 8                # max_batch: windows of 4, then 8 iterations


def bookkeeping(pt):
    """(windows enqueued, calls that had to trace their own window first, windows discarded, size of the window being consumed)"""
    import ctypes as C
    out = (C.c_uint64 * 4)()
    assert pt.library().ptdbg_lookahead(out) == 0
    return tuple(int(v) for v in out)


class Oracle:
    """The oracle's running sum, carried across camera / depth changes (the accumulation buffer survives them in the
    library too: only the host resets it, by re-initialising)."""

    def __init__(self, po, s, cam, depth):
        self.po, self.s = po, s
        self.image = None
        self.rays = 0
        self.retarget(cam, depth)

    def retarget(self, cam, depth):
        t = self.po.Tracer(self.s["geoms"], self.s["materials"], cam, depth, flags=self.po.F_COMPACT, trig=self.po.TRIG_SHARED)
        if self.image is not None:
            t.image[:] = self.image
        self.t, self.image = t, t.image

    def iterate(self, it):
        st = self.t.iterate(it, threads=8)
        self.rays += st.rays
        return self.image


@pytest.mark.parametrize("host_flags", ["pin+sparse", "pin", "pageable"])
def test_host_image_after_every_call_equals_the_oracle(pt, po, scenes, launch_plan, host_flags):
    s = scenes["cornell"]
    w, h = 400, 300                                        # 1.44 MB of image: above the 1 MiB from which PT_PIN_IMAGE page-locks
    cam = _resized(s["camera"], w, h)
    n = w * h
    depth = s["depth"]
    scene = pt.Scene(s["geoms"], s["materials"], cam, depth)
    L = pt.library()
    extra = {"pin+sparse": pt.PT_PIN_IMAGE | pt.PT_HOST_SPARSE, "pin": pt.PT_PIN_IMAGE, "pageable": 0}[host_flags]
    flags = pt.PT_COMPACT | pt.PT_LOOKAHEAD | extra
    a = np.full((n, 3), -7.0, dtype=np.float32)
    b = np.full((n, 3), -9.0, dtype=np.float32)
    cam2 = cam.copy()
    cam2["position"][0][1] += 0.5

    pt.pathtraceInit(scene, flags=flags, max_batch=K, pin_image=False)
    ref = Oracle(po, s, cam, depth)
    served = []

    def call(buf, it, camera=cam, d=depth):
        pt.set_camera(camera, d)                           # the shim forwards both on every pathtrace() (pathtrace.cu:285-286)
        assert L.pt_trace(None, 0, it, buf.ctypes.data) == 0, L.pt_last_error()
        want = ref.iterate(it)
        assert (bits(buf) == bits(want)).all(), "host image after iteration %d" % it
        assert (bits(pt.get_image(n)) == bits(want)).all(), "device image after iteration %d" % it
        served.append(pt.get_stats().rays)

    for it in range(1, 3 * K - 3):                         # 1..20: windows [1,4] [5,12] [13,20], the last consumed to its end
        call(a, it)
    assert sum(served) == ref.rays                         # statistics are additive: every window reported once, whole
    assert bookkeeping(pt) == (6, 1, 0, K)                 # [1,4] traced by call 1; [5,12] .. [37,44] ahead, three at a time; nothing thrown away
    assert pt.counters()[0] >= ref.rays                    # (the device has traced, or is tracing, [21, 44] ahead as well)
    if host_flags == "pin+sparse":                         # such calls take the CU-masked streams (256 compute units in 8 XCDs) ...
        assert masked_bookkeeping(pt) in ((6, 19), (0, 0)) # ... every window, and every call but the first (a whole-frame copy)
    else:
        assert masked_bookkeeping(pt) == (0, 0)
    call(a, 22)                                            # 21 is skipped: the window traced ahead starts at 21 and is void
    assert bookkeeping(pt) == (10, 2, 3, 4)                # [21,28] [29,36] [37,44] thrown away; [22,25] by this call, [26,33] [34,41] [42,49] ahead
    call(a, 23); call(b, 24); call(a, 25); call(a, 26)     # (24: a second host buffer; the 4-iteration window ends at 25)
    call(a, 26)                                            # the same number again: not consecutive either
    call(a, 27)
    ref.retarget(cam2, depth)                              # the camera moves mid-window (main.cpp:102-120)
    call(a, 28, cam2); call(a, 29, cam2); call(a, 30, cam2)
    ref.retarget(cam2, depth - 3)                          # traceDepth is re-read on every call (pathtrace.cu:286)
    for it in range(31, 31 + K + 2):
        call(a, it, cam2, depth - 3)
    pt.trace_batch(100, 2, None)                           # a batch in between (another entry point uses the same pools)
    ref.iterate(100); ref.iterate(101)
    call(a, 102, cam2, depth - 3); call(a, 103, cam2, depth - 3)
    pt.clear_image()
    ref.image[:] = 0
    call(a, 104, cam2, depth - 3); call(a, 105, cam2, depth - 3)
    pt.pathtraceFree()                                     # pathtraceFree + pathtraceInit: what every camera move does (main.cpp:125-128)

    pt.pathtraceInit(scene, flags=flags, max_batch=K, pin_image=False)
    ref = Oracle(po, s, cam, depth)
    served.clear()
    for it in range(1, K):
        call(a, it)
    pt.pathtraceFree()


def masked_bookkeeping(pt):
    """(windows traced on the lanes' CU-masked streams, calls whose gather ran on the compute units set aside for it)"""
    import ctypes as C
    out = (C.c_uint64 * 2)()
    assert pt.library().ptdbg_lookahead_masked(out) == 0
    return tuple(int(v) for v in out)


def test_calls_of_every_kind_on_windows_of_the_masked_streams(pt, po, scenes, launch_plan):
    """Calls that write a page-locked host image take windows traced on CU-masked streams and gather on compute units of
    their own (csrc/pt_h_enqueue.hpp: ensure_la_masks).  One chain of such windows serves calls with the host image, without
    one, with a PBO, with another buffer; asynchronous batches (the lanes' plain streams, the same buffers) come in between."""
    import torch
    s = scenes["cornell"]
    w, h = 400, 300
    cam = _resized(s["camera"], w, h)
    n = w * h
    depth = s["depth"]
    scene = pt.Scene(s["geoms"], s["materials"], cam, depth)
    L = pt.library()
    a = np.full((n, 3), -7.0, dtype=np.float32)
    b = np.full((n, 3), -9.0, dtype=np.float32)
    pbo = torch.zeros(n * 4, dtype=torch.uint8, device="cuda:0")
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_LOOKAHEAD | pt.PT_PIN_IMAGE | pt.PT_HOST_SPARSE, max_batch=K, pin_image=False)
    ref = Oracle(po, s, cam, depth)

    def call(it, buf=None, with_pbo=False):
        assert L.pt_trace(pbo.data_ptr() if with_pbo else None, 0, it, buf.ctypes.data if buf is not None else None) == 0, L.pt_last_error()
        want = ref.iterate(it)
        if buf is not None:
            assert (bits(buf) == bits(want)).all(), "host image after iteration %d" % it
        assert (bits(pt.get_image(n)) == bits(want)).all(), "device image after iteration %d" % it
        if with_pbo:
            assert pbo.cpu().numpy().tobytes() == pt.tonemap(n, it).tobytes(), it

    it = 1
    for kind in ("a", "a", "a", None, "a", "a", "pbo+a", "a", "b", "b", "a", "a", "pbo", "a", "a", None, None, "a", "a", "a"):
        call(it, {"a": a, "b": b, "pbo+a": a}.get(kind), with_pbo=kind in ("pbo", "pbo+a"))
        it += 1
    windows, misses, discards, _ = bookkeeping(pt)
    masked_windows, masked_calls = masked_bookkeeping(pt)
    assert (misses, discards) == (1, 0)
    if masked_windows:                                     # (a device that is not 256 compute units in 8 XCDs keeps the plain streams)
        assert masked_windows == windows                   # one chain: every window the kind its first call asked for
        assert masked_calls == 10                          # the calls that found the buffer they write current: a a | a a | a | b | a | a | a a
    for rounds in range(2):                                # batches on the lanes' plain streams, then calls again: twice over
        for k in range(3):
            pt.trace_batch_async(100 * (rounds + 1) + 4 * k, 4)
            for j in range(4):
                ref.iterate(100 * (rounds + 1) + 4 * k + j)
        for j in range(K + 2):
            call(200 * (rounds + 1) + j, a)
    assert masked_bookkeeping(pt)[0] in (0, bookkeeping(pt)[0])
    pt.pathtraceFree()


def test_pbo_and_image_without_a_host_buffer(pt, po, scenes, launch_plan):
    """The reference's GL host hands over a PBO every call (main.cpp:131-137): tonemapped by the same launch that gathers
    the sample.  sendImageToPBO's arithmetic (pathtrace.cu:48-68: float divide, double multiply, truncation) restated in numpy."""
    import torch
    s = scenes["cornell"]
    w, h = 160, 120
    cam = _resized(s["camera"], w, h)
    n = w * h
    scene = pt.Scene(s["geoms"], s["materials"], cam, s["depth"])
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_LOOKAHEAD, max_batch=K)
    ref = Oracle(po, s, cam, s["depth"])
    pbo = torch.zeros(n * 4, dtype=torch.uint8, device="cuda:0")
    for it in range(1, 2 * K + 3):
        img = pt.pathtrace(pbo.data_ptr(), 0, it)
        want = ref.iterate(it)
        assert (bits(img) == bits(want)).all(), it
        v = (want / np.float32(it)).astype(np.float64) * 255.0
        rgb = np.clip(v.astype(np.int64), 0, 255).astype(np.uint8)
        got = pbo.cpu().numpy().reshape(n, 4)
        assert (got[:, :3] == rgb).all() and (got[:, 3] == 0).all(), it
        assert got.tobytes() == pt.tonemap(n, it).tobytes()
    pt.pathtraceFree()


def test_lookahead_equals_the_plain_calls_at_full_size(pt, scenes, monkeypatch):
    """BASELINE configs[1] as the reference's host drives it: 800x800, one pathtrace() per iteration, windows of up to 64
    iterations.  The image after every call against the same calls without the flag (themselves held against the oracle by
    tests/test_gpu_parity.py), through three window sizes and into the steady state."""
    monkeypatch.delenv("PTMI355_WHOLE_MAX", raising=False)
    s = scenes["cornell"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    n = 800 * 800
    L = pt.library()
    calls = 4 + 16 + 64 + 64 + 5

    def run(extra, max_batch):
        pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_PIN_IMAGE | pt.PT_HOST_SPARSE | extra, max_batch=max_batch, pin_image=False)
        host = np.zeros((n, 3), dtype=np.float32)
        out = []
        for it in range(1, calls + 1):
            assert L.pt_trace(None, 0, it, host.ctypes.data) == 0
            out.append(host.copy())
        dev = pt.get_image(n).copy()
        if extra:
            book.append(bookkeeping(pt))
        pt.pathtraceFree()
        return out, dev

    book = []

    want, dev0 = run(0, 1)
    got, dev1 = run(pt.PT_LOOKAHEAD, 64)
    assert book == [(8, 1, 0, 64)]                          # [1,4] [5,20] [21,84] [85,148] [149,212] (+ three ahead, up to [341,404]): only call 1 traced its own window
    for it, (g, wnt) in enumerate(zip(got, want), 1):
        assert (bits(g) == bits(wnt)).all(), it
    assert (bits(dev0) == bits(dev1)).all()


def test_4k_frame_in_windows_of_four(pt, scenes, monkeypatch):
    """C5's frame (3840x2160, 99.5 MB of host image) as the shim initialises it: windows of at most four iterations (33 M paths).
    The host image after every call against the plain calls'."""
    monkeypatch.delenv("PTMI355_WHOLE_MAX", raising=False)
    s = scenes["cornell_4k"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    n = 3840 * 2160
    L = pt.library()
    import hashlib

    def run(extra, max_batch):
        pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_PIN_IMAGE | pt.PT_HOST_SPARSE | extra, max_batch=max_batch, pin_image=False)
        host = np.zeros((n, 3), dtype=np.float32)
        out = []
        for it in range(1, 12):
            assert L.pt_trace(None, 0, it, host.ctypes.data) == 0
            out.append(hashlib.md5(host.tobytes()).hexdigest())
        pt.pathtraceFree()
        return out

    assert run(pt.PT_LOOKAHEAD, 4) == run(0, 1)


def test_jittered_camera_and_a_lens_change_mid_window(pt, scenes, launch_plan):
    """PT_AA_JITTER and the thin lens draw from the engine of (iteration, pixel): still functions of the iteration number alone, so
    windows stay valid -- and a pt_set_lens in the middle of one voids it (the window was traced with the old lens).  Against
    the plain calls, image by image."""
    s = scenes["cornell_64"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    n = 64 * 64

    def run(extra, max_batch):
        pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_AA_JITTER | extra, max_batch=max_batch, lens=(0.05, 9.0))
        out = []
        for it in range(1, 20):
            if it == 7:
                pt.set_lens(0.2, 7.5)
            if it == 15:
                pt.set_lens(0.0, 0.0)
            out.append(pt.pathtrace(None, 0, it).copy())
        pt.pathtraceFree()
        return out

    want = run(0, 1)
    got = run(pt.PT_LOOKAHEAD, K)
    for it, (g, w) in enumerate(zip(got, want), 1):
        assert (bits(g) == bits(w)).all(), it
    assert not (bits(want[5]) == bits(want[7])).all()


@pytest.mark.parametrize("pipeline", ["sorted glass", "mesh hierarchy", "mesh loop"])
def test_other_pipelines_behind_the_windows(pt, scenes, launch_plan, pipeline):
    """The windows go through whatever batched pipeline the session runs: the material sort folded into the compaction (C3's
    glass ball, depth 16), the mesh pre-pass + hierarchy, the loop over every triangle.  Image after every call against the
    plain calls (each of those pipelines is held against the oracle by its own tests)."""
    if pipeline == "sorted glass":
        s = scenes["cornell_glass"]
        cam = _resized(s["camera"], 96, 54)
        scene, flags, depth = pt.Scene(s["geoms"], s["materials"], cam, s["depth"]), pt.PT_COMPACT | pt.PT_SORT_MATERIAL, s["depth"]
    else:
        s = scenes["cornell_64"]
        tris = pt.meshes.uv_sphere(n_lat=12, n_lon=24)
        geoms, tris, meshes = pt.meshes.add_mesh(s["geoms"], tris, material_id=1)
        scene = pt.Scene(geoms, s["materials"], s["camera"], s["depth"], triangles=tris, meshes=meshes)
        flags, depth = pt.PT_COMPACT | (pt.PT_MESH_BVH if pipeline == "mesh hierarchy" else 0), s["depth"]

    def run(extra, max_batch):
        pt.pathtraceInit(scene, flags=flags | extra, max_batch=max_batch)
        out = [pt.pathtrace(None, 0, it).copy() for it in range(1, 2 * K + 4)]
        pt.pathtraceFree()
        return out

    want = run(0, 1)
    got = run(pt.PT_LOOKAHEAD, K)
    for it, (g, w) in enumerate(zip(got, want), 1):
        assert (bits(g) == bits(w)).all(), (pipeline, it)
    assert want[-1].sum() > 0


@pytest.mark.parametrize("pipeline", ["sorted glass + jitter + lens", "mesh loop", "mesh hierarchy"])
def test_other_pipelines_on_the_masked_streams(pt, scenes, launch_plan, pipeline):
    """The same with a page-locked host image of more than 1 MiB (PT_PIN_IMAGE | PT_HOST_SPARSE): such chains of windows go to
    the lanes' CU-masked streams with a smaller persistent grid, their calls gather on compute units of their own -- except
    under PT_MESH_BVH (k_mesh's grid is sized for the whole chip: plain streams).  The host image after every call against
    the plain calls'."""
    w, h = 416, 304
    n = w * h
    if pipeline.startswith("sorted glass"):
        s = scenes["cornell_glass"]
        scene = pt.Scene(s["geoms"], s["materials"], _resized(s["camera"], w, h), s["depth"])
        flags, lens = pt.PT_COMPACT | pt.PT_SORT_MATERIAL | pt.PT_AA_JITTER, (0.15, 8.0)
    else:
        s = scenes["cornell"]
        tris = pt.meshes.uv_sphere(n_lat=12, n_lon=24)
        geoms, tris, meshes = pt.meshes.add_mesh(s["geoms"], tris, material_id=1)
        scene = pt.Scene(geoms, s["materials"], _resized(s["camera"], w, h), s["depth"], triangles=tris, meshes=meshes)
        flags, lens = pt.PT_COMPACT | (pt.PT_MESH_BVH if pipeline == "mesh hierarchy" else 0), None
    L = pt.library()

    def run(extra, max_batch):
        pt.pathtraceInit(scene, flags=flags | pt.PT_PIN_IMAGE | pt.PT_HOST_SPARSE | extra, max_batch=max_batch, pin_image=False)
        if lens:
            pt.set_lens(*lens)
        host = np.full((n, 3), -3.0, dtype=np.float32)
        out = []
        for it in range(1, 2 * K + 4):
            assert L.pt_trace(None, 0, it, host.ctypes.data) == 0, L.pt_last_error()
            out.append(host.copy())
        masked = masked_bookkeeping(pt) if extra else None
        windows = bookkeeping(pt)[0] if extra else None
        pt.pathtraceFree()
        return out, masked, windows

    want, _, _ = run(0, 1)
    got, masked, windows = run(pt.PT_LOOKAHEAD, K)
    for it, (g, v) in enumerate(zip(got, want), 1):
        assert (bits(g) == bits(v)).all(), (pipeline, it)
    assert want[-1].sum() > 0 and windows >= 4
    if pipeline == "mesh hierarchy":
        assert masked == (0, 0)
    else:
        assert masked in ((windows, 2 * K + 2), (0, 0))    # every window, every call but the first (256 compute units in 8 XCDs; else none)


def test_flag_is_ignored_where_it_cannot_apply(pt, scenes, launch_plan):
    """max_batch = 1, the fake shader, the unfused pipeline: pt_trace takes its plain path and the image is the usual one."""
    s = scenes["cornell_64"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    n = 64 * 64

    def run(flags, max_batch):
        pt.pathtraceInit(scene, flags=flags, max_batch=max_batch)
        for it in range(1, 7):
            img = pt.pathtrace(None, 0, it).copy()
        live = list(pt.get_stats().live[:s["depth"]])
        pt.pathtraceFree()
        return img, live

    for base, mb in ((pt.PT_COMPACT, 1), (pt.PT_FAKE_SHADER, 4), (pt.PT_COMPACT | pt.PT_UNFUSED, 4)):
        a, la = run(base, mb)
        b, lb = run(base | pt.PT_LOOKAHEAD, mb)
        assert (bits(a) == bits(b)).all() and la == lb and sum(la) > 0
