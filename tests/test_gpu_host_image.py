"""GPU parity, the host image: what pt_trace / pt_trace_batch hand back in host memory -- pageable copies, the page-locked
image the launch writes itself (PT_PIN_IMAGE, PT_HOST_SPARSE, PT_ASYNC_IMAGE, PT_SHARED_IMAGE), resumed accumulations, the PBO --
against the device's running sum and the oracle, bit for bit, under both launch plans (tests/gpu_common.py)."""
import hashlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import __graft_entry__ as ge  # noqa: E402,F401
from gpu_common import pt, launch_plan, bits, rel_l2, assert_paths_equal, _resized, _after  # noqa: E402,F401

pytestmark = pytest.mark.gpu


def test_resumed_accumulation_equals_the_uninterrupted_run(pt, po, scenes, tmp_path):
    """pt_set_image + ptbench --save-sum / --resume (C5's 5000 spp across GPU leases): the running sum is the whole
    state the reference carries between iterations (dev_image, pathtrace.cu:71,84,389), so 2 x N/2 iterations with the
    sum taken through host memory (and a PFM file) in between == N iterations, bit for bit -- per call and batched,
    one device and three contexts, and through the headless host."""
    import subprocess
    s = scenes["cornell_64"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    npix = 64 * 64
    ref = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"])
    for it in range(1, 13):
        ref.iterate(it)
    for devices in (None, [0, 0, 0]):
        pt.pathtraceInit(scene, max_batch=4, devices=devices)
        for it in range(1, 7):
            pt.pathtrace(None, 0, it)
        half = pt.get_image(npix)
        pt.pathtraceFree()
        path = str(tmp_path / "half.6samp.sum.pfm")
        pt.save_pfm(path, half, 64, 64, 1.0)
        pt.pathtraceInit(scene, max_batch=4, devices=devices)           # a new session: nothing survives but the file
        pt.set_image(pt.load_pfm(path, 64, 64))
        pt.trace_batch(7, 4)
        pt.pathtrace(None, 0, 11)
        img = pt.pathtrace(None, 0, 12).copy()
        pt.pathtraceFree()
        assert img.tobytes() == ref.image.tobytes(), devices
    with pytest.raises(pt.PtError):
        pt.set_image(half)                                              # no session
    # the headless host: 5 + 7 iterations in two processes == 12 in one
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "scenes", "cornell.txt")).read().replace("RES         800 800", "RES         64 64")
    scene_file = tmp_path / "cornell64.txt"
    scene_file.write_text(txt)
    exe = pt.build_ptbench()
    run = lambda *a: subprocess.run([exe, str(scene_file)] + list(a), capture_output=True, text=True, timeout=300)
    p = run("--iters", "5", "--batch", "2", "--out", str(tmp_path / "a"), "--save-sum")
    assert p.returncode == 0, p.stdout + p.stderr
    p = run("--iters", "12", "--batch", "3", "--out", str(tmp_path / "a"), "--resume", str(tmp_path / "a.5samp.sum.pfm"), "--save-sum")
    assert p.returncode == 0 and "resumed" in p.stdout, p.stdout + p.stderr
    got = pt.load_pfm(str(tmp_path / "a.12samp.sum.pfm"), 64, 64)
    assert got.tobytes() == ref.image.tobytes()
    p = run("--iters", "3", "--resume", str(tmp_path / "a.5samp.sum.pfm"))
    assert p.returncode != 0                                            # 5 iterations done, 3 wanted


def test_async_image_mode(pt, scenes):
    """PT_ASYNC_IMAGE: pathtrace() returns without waiting for its own copy; the running sum reaches the host while
    the next call traces.  When call i+1 returns the buffer of call i is complete; pt_synchronize completes the last
    one; every sum equals the synchronous mode's."""
    s = scenes["cornell_64"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    n = scene.resolution[0] * scene.resolution[1]
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT)
    sums = [pt.pathtrace(None, 0, it).copy() for it in range(1, 7)]
    pt.pathtraceFree()
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_ASYNC_IMAGE)
    bufs = [np.zeros((n, 3), dtype=np.float32) for _ in range(3)]
    L = pt.library()
    for it in range(1, 7):
        assert L.pt_trace(None, 0, it, bufs[it % 3].ctypes.data) == 0
        if it >= 2:                               # the buffer of the previous call is complete when this one returns
            assert bufs[(it - 1) % 3].tobytes() == sums[it - 2].tobytes()
    pt.synchronize()
    assert bufs[6 % 3].tobytes() == sums[5].tobytes()
    assert pt.get_image(n).tobytes() == sums[5].tobytes()
    pt.pathtraceFree()


def test_host_image_freed_and_reallocated_between_calls(pt, po, scenes):
    """pathtrace() copies the running sum into WHATEVER buffer it is handed (pathtrace.cu:389-390 is a plain
    cudaMemcpy): without PT_PIN_IMAGE the library keeps no claim on a buffer after the call returns -- the host may
    free it, and a new allocation (often at the same address) is just another buffer.  1200 x 900 x 12 B: above the
    1 MiB from which PT_PIN_IMAGE would page-lock.  With the flag the one long-lived buffer gives the same sums."""
    s = scenes["cornell"]
    cam = _resized(s["camera"], 1200, 900)
    scene = pt.Scene(s["geoms"], s["materials"], cam, 3)
    n = 1200 * 900
    L = pt.library()
    ref = po.Tracer(s["geoms"], s["materials"], cam, 3)
    want = []
    for it in range(1, 7):
        ref.iterate(it, threads=8)
        want.append(ref.image.copy())
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT, pin_image=False)
    for it in range(1, 7):
        buf = np.empty((n, 3), dtype=np.float32)          # a fresh buffer per call ...
        buf[:] = -1.0
        assert L.pt_trace(None, 0, it, buf.ctypes.data) == 0
        assert buf.tobytes() == want[it - 1].tobytes(), it
        del buf                                           # ... freed before the next
    pt.pathtraceFree()
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_PIN_IMAGE, pin_image=False)
    keep = np.zeros((n, 3), dtype=np.float32)
    for it in range(1, 7):
        assert L.pt_trace(None, 0, it, keep.ctypes.data) == 0
        assert keep.tobytes() == want[it - 1].tobytes(), it
    pt.pathtraceFree()


def test_host_image_kept_current_incrementally(pt, scenes, monkeypatch):
    """pathtrace() per call with a long-lived page-locked host image (PT_PIN_IMAGE | PT_HOST_SPARSE): from the second call
    on, the launch adds every ending path's colour to its pixel itself and writes only those pixels to the host
    (BounceArgs::epi_direct; the others still hold their sums).  After every call the host buffer IS the device's running sum, whatever else
    happened in between: overlapped batches (k_gather wrote the buffer), pt_clear_image, pt_set_image, a second host
    buffer, a camera move; and the whole sequence equals the one with every pixel written every call."""
    s = scenes["cornell"]
    cam = _resized(s["camera"], 400, 300)                  # 1.44 MB of image: above the 1 MiB from which PT_PIN_IMAGE page-locks
    scene = pt.Scene(s["geoms"], s["materials"], cam, s["depth"])
    n = 400 * 300
    L = pt.library()
    cam2 = cam.copy()
    cam2["position"][0][1] += 0.5

    def run(env, extra=0):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_PIN_IMAGE | extra, pin_image=False)
        a = np.full((n, 3), -7.0, dtype=np.float32)
        b = np.full((n, 3), -9.0, dtype=np.float32)
        out = []

        def call(buf, it):
            assert L.pt_trace(None, 0, it, buf.ctypes.data) == 0
            assert buf.tobytes() == pt.get_image(n).tobytes(), it
            out.append(buf.tobytes())

        for it in (1, 2, 3):
            call(a, it)
        pt.trace_batch_async(4, 1); pt.trace_batch_async(5, 1)            # the buffer changes behind the host's copy
        call(a, 6); call(a, 7)
        pt.clear_image()
        call(a, 8); call(a, 9)
        call(b, 10); call(a, 11); call(a, 12); call(b, 13)               # two host buffers in turn
        pt.set_image(np.ascontiguousarray(np.frombuffer(out[2], dtype=np.float32).reshape(n, 3)))
        call(a, 14); call(a, 15)
        pt.set_camera(cam2, s["depth"] - 2)
        call(a, 16); call(a, 17)
        pt.trace_batch(18, 1, None)                                      # a synchronous batch without a host image
        call(a, 19)
        pt.pathtraceFree()
        for k in env:
            monkeypatch.delenv(k)
        return out

    ref = run({})                                         # PT_PIN_IMAGE alone: every pixel, every call
    assert run({}, pt.PT_HOST_SPARSE) == ref
    if pt.has_experiments():                              # (a -DPT_EXPERIMENTS build: the plans the default replaced)
        assert run({"PTMI355_EPI_DIRECT": "0"}, pt.PT_HOST_SPARSE) == ref
        assert run({"PTMI355_HOST_EPILOGUE": "0"}, pt.PT_HOST_SPARSE) == ref


def test_host_writes_between_calls(pt, scenes, launch_plan):
    """what a host's own writes into the image do.  PT_PIN_IMAGE alone keeps the reference's semantics -- every
    call hands back the WHOLE running sum (pathtrace.cu:389-390), so scribbles are overwritten; under PT_HOST_SPARSE the
    host has promised to only read, the binding returns a read-only view, and a scribble through the raw buffer survives
    exactly on pixels whose sum did not change (documented in include/ptmi355.h) while every other pixel is current."""
    s = scenes["cornell"]
    cam = _resized(s["camera"], 400, 300)
    scene = pt.Scene(s["geoms"], s["materials"], cam, s["depth"])
    n = 400 * 300
    L = pt.library()
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_PIN_IMAGE, pin_image=False)
    buf = np.zeros((n, 3), dtype=np.float32)
    for it in (1, 2, 3, 4):
        assert L.pt_trace(None, 0, it, buf.ctypes.data) == 0
        assert buf.tobytes() == pt.get_image(n).tobytes(), it
        buf /= float(it)                                  # the host normalises in place ...
        buf[::7] = -1.0                                   # ... and scribbles
    pt.pathtraceFree()

    pt.pathtraceInit(scene, flags=pt.PT_COMPACT, host_sparse=True)       # (pin_image=True: scene.image)
    img = pt.pathtrace(None, 0, 1)
    assert not img.flags.writeable and img.tobytes() == pt.get_image(n).tobytes()
    with pytest.raises(ValueError):
        img[0, 0] = 1.0
    img = pt.pathtrace(None, 0, 2)
    before = pt.get_image(n).copy()
    scene.image[::5] = -3.0                               # behind the binding's back: the promise broken
    img = pt.pathtrace(None, 0, 3)
    dev = pt.get_image(n)
    changed = (dev.view(np.uint32) != before.view(np.uint32)).any(axis=1)
    assert changed.any() and not changed.all()
    assert np.asarray(img)[changed].tobytes() == dev[changed].tobytes()         # every pixel whose sum changed is current
    if launch_plan == "one launch per bounce":            # the flag is a permission: this plan copies the whole image anyway
        assert np.asarray(img).tobytes() == dev.tobytes()
        pt.pathtraceFree()
        return
    stale = ~changed
    stale[np.arange(n) % 5 != 0] = False
    assert stale.any() and (np.asarray(img)[stale] == -3.0).all()               # the rest is as the host left it
    pt.pathtraceFree()


def test_async_image_written_by_the_launch(pt, scenes, monkeypatch):
    """PT_ASYNC_IMAGE with a page-lockable image (400 x 300 x 12 B > 1 MiB): pt_trace returns without waiting and the
    launch writes the host buffer itself (only the pixels that changed when the buffer is the one it wrote last).  The
    contract of the flag holds call for call: a buffer is complete when the NEXT call returns (or after pt_synchronize)
    and then holds exactly the sum after its own call -- one buffer reused, two buffers in turn, a batch call (snapshot
    + copy engine) in between -- and equals the copy-engine-only plan and the synchronous sums."""
    s = scenes["cornell"]
    cam = _resized(s["camera"], 400, 300)
    scene = pt.Scene(s["geoms"], s["materials"], cam, s["depth"])
    n = 400 * 300
    L = pt.library()
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT, pin_image=False)
    want = {}
    for it in range(1, 15):
        want[it] = pt.pathtrace(None, 0, it).tobytes()
    pt.pathtraceFree()

    def run(env, extra=0):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_ASYNC_IMAGE | extra, pin_image=False)
        a = np.full((n, 3), -1.0, dtype=np.float32)
        b = np.full((n, 3), -2.0, dtype=np.float32)
        c = np.full((n, 3), -3.0, dtype=np.float32)
        ok = []
        assert L.pt_trace(None, 0, 1, a.ctypes.data) == 0
        assert L.pt_trace(None, 0, 2, b.ctypes.data) == 0
        ok.append(a.tobytes() == want[1])                         # complete when the next call has returned
        assert L.pt_trace(None, 0, 3, a.ctypes.data) == 0
        ok.append(b.tobytes() == want[2])
        assert L.pt_trace(None, 0, 4, a.ctypes.data) == 0         # the same buffer again: only what changed is written
        assert L.pt_trace(None, 0, 5, a.ctypes.data) == 0
        assert L.pt_trace_batch(6, 1, c.ctypes.data) == 0         # batch entry point: snapshot + copy engine
        ok.append(a.tobytes() == want[5])
        assert L.pt_trace(None, 0, 7, a.ctypes.data) == 0
        ok.append(c.tobytes() == want[6])
        assert L.pt_trace(None, 0, 8, c.ctypes.data) == 0         # the buffer the copy engine wrote, now written by the launch
        ok.append(a.tobytes() == want[7])
        assert L.pt_trace(None, 0, 9, c.ctypes.data) == 0
        pt.synchronize()
        ok.append(c.tobytes() == want[9])
        pt.clear_image()
        for it in (1, 2, 3):
            assert L.pt_trace(None, 0, it, c.ctypes.data) == 0
        pt.synchronize()
        ok.append(c.tobytes() == want[3])
        pt.pathtraceFree()
        for k in env:
            monkeypatch.delenv(k)
        return ok

    assert all(run({}, pt.PT_HOST_SPARSE)), "launch-written, the pixels that changed"
    assert all(run({})), "launch-written, every pixel"
    if pt.has_experiments():
        assert all(run({"PTMI355_ASYNC_DIRECT": "0"})), "copy engine"


def test_shared_host_frame_assembled_by_the_tiles(pt, scenes, launch_plan):
    """PT_SHARED_IMAGE: the ranks of a tiled frame hand pt_trace ONE host frame and each writes only the pixels of its own
    tile into it (first call: all of them; later calls: the ones whose sum changed) -- the frame is assembled in host memory
    with no exchange.  Here the "ranks" are sessions of this process, one after the other, interleaved call by call on two
    frames: tiles of 2 and of 3 ranks (strips of 8 and of 5 rows: the last strip short) give the 1-session sums, pixel
    for pixel, and a session never touches a pixel outside its tile."""
    s = scenes["cornell"]
    cam = _resized(s["camera"], 400, 300)
    scene = pt.Scene(s["geoms"], s["materials"], cam, s["depth"])
    n = 400 * 300
    L = pt.library()
    if launch_plan == "one launch per bounce":                   # the flag needs one-launch iterations: refused under this plan
        pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_SHARED_IMAGE, tile=(0, 2, 8), pin_image=False)
        frame = np.zeros((n, 3), dtype=np.float32)
        assert L.pt_trace(None, 0, 1, frame.ctypes.data) < 0 and not frame.any()
        pt.pathtraceFree()
        return
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT, pin_image=False)
    want = {}
    for it in (1, 2, 3, 4):
        want[it] = pt.pathtrace(None, 0, it).reshape(n, 3).copy()
    pt.pathtraceFree()
    for ranks, strip in ((2, 8), (3, 5)):
        frame = np.full((n, 3), -5.0, dtype=np.float32)
        owned = []
        for r in range(ranks):
            rows = pt.sharding.owned_rows(r, ranks, strip, 300)
            owned.append(np.repeat(rows, 400))
        for r in range(ranks):                                   # rank r: its four iterations into the shared frame
            before = frame.copy()
            pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_SHARED_IMAGE, tile=(r, ranks, strip), pin_image=False)
            for it in (1, 2, 3, 4):
                assert L.pt_trace(None, 0, it, frame.ctypes.data) == 0
                assert frame[owned[r]].tobytes() == want[it][owned[r]].tobytes(), (ranks, r, it)
            pt.pathtraceFree()
            assert frame[~owned[r]].tobytes() == before[~owned[r]].tobytes(), (ranks, r)       # nobody else's pixels
        assert frame.tobytes() == want[4].tobytes(), ranks
    # what cannot run as one launch is refused, not copied over the other ranks' pixels
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_SORT_MATERIAL | pt.PT_SHARED_IMAGE, tile=(0, 2, 8), pin_image=False)
    frame = np.zeros((n, 3), dtype=np.float32)
    assert L.pt_trace(None, 0, 1, frame.ctypes.data) < 0
    pt.pathtraceFree()


def test_4k_one_iteration_per_call_into_the_host_image(pt, scenes, monkeypatch):
    """C5's frame (3840 x 2160 = 8.3 M paths) through pathtrace() per call with a page-locked host image: one launch per
    iteration although the frame is above the 6 M paths up to which batches run as one launch (the launch hides the PCIe
    transfer).  Host image == device sum after every call == the kernel-per-bounce plan with a copy per call."""
    s = scenes["cornell"]
    cam = _resized(s["camera"], 3840, 2160)
    scene = pt.Scene(s["geoms"], s["materials"], cam, s["depth"])
    n = 3840 * 2160
    L = pt.library()

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_PIN_IMAGE, pin_image=False)
        host = np.full((n, 3), -3.0, dtype=np.float32)
        out = []
        for it in (1, 2, 3):
            assert L.pt_trace(None, 0, it, host.ctypes.data) == 0
            out.append(hashlib.md5(host.tobytes()).hexdigest())
        assert host.tobytes() == pt.get_image(n).tobytes()
        out.append(tuple(int(v) for v in pt.counters()))
        pt.pathtraceFree()
        for k in env:
            monkeypatch.delenv(k)
        return out

    assert run({}) == run({"PTMI355_WHOLE_MAX_HOST": "0"})


def test_pbo_device_pointer(pt, scenes, golden):
    """pathtrace() writes the tonemapped RGBA8 into a device buffer (the mapped PBO)."""
    import torch
    s = scenes["cornell_64"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    pt.pathtraceInit(scene, flags=pt.PT_FAKE_SHADER)
    pbo = torch.zeros(64 * 64 * 4, dtype=torch.uint8, device="cuda:0")
    for it in (1, 2, 3):
        pt.pathtrace(pbo.data_ptr(), 0, it)
    torch.cuda.synchronize()
    assert pbo.cpu().numpy().tobytes() == golden["fakeshade"]["pbo64"].tobytes()
    pt.pathtraceFree()
