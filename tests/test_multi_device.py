"""Several GPUs behind ONE session of the C-ABI (include/ptmi355.h: pt_scene_desc::devices, csrc/pt_multi.hpp):
the frame device 0 assembles from the devices' tiles is the single-device image, bit for bit -- per call
(pathtrace()), per batch, with asynchronous batches, with every pipeline.

On a 1-GPU box the contexts share the one device (tiles then travel by device-to-device copies: RCCL refuses two
ranks on one device) and every RCCL call of the exchange is executed with a communicator of ONE device
(PTMI355_XCHG=rccl); with >= 2 devices the same tests run over RCCL send / recv."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pt():
    p = ge.load_package()
    p.library()
    yield p
    p.pathtraceFree()


def gpu_count():
    import torch
    return torch.cuda.device_count()            # counting does not initialise the GPU


def device_lists():
    """Context -> device assignments worth testing on this box."""
    n = gpu_count()
    out = [[0, 0], [0, 0, 0]]
    if n >= 2:
        out += [[0, 1], list(range(min(n, 8)))]
    return out


def run_calls(pt, scene, n, **kw):
    """The calling patterns of a host: pathtrace() per iteration, a synchronous batch, asynchronous batches."""
    pt.pathtraceInit(scene, max_batch=4, **kw)
    out = []
    for it in (1, 2, 3):
        out.append(pt.pathtrace(None, 0, it).copy())                 # state.image after every call
    live = list(pt.get_stats().live[:scene.traceDepth])
    img = np.zeros((n, 3), dtype=np.float32)
    pt.trace_batch(4, 4, img)
    out.append(img.copy())
    pt.trace_batch_async(8, 3)
    pt.trace_batch_async(11, 4)
    pt.trace_batch_async(15, 1)
    pt.synchronize()
    out.append(pt.get_image(n).copy())
    rays, first, iters = pt.counters()
    rgba = pt.tonemap(n, 15).copy()
    ndev, transport = pt.num_devices(), pt.exchange_transport()
    pt.pathtraceFree()
    return out, live, (rays, first, iters), rgba, ndev, transport


@pytest.mark.parametrize("pipeline", ["fused", "sort", "nocompact", "bvh+aa+lens"])
def test_contexts_equal_single_device(pt, scenes, pipeline, monkeypatch):
    s = scenes["cornell_glass_64" if pipeline != "fused" else "cornell_64"]
    kw = {"fused": dict(flags=pt.PT_COMPACT), "sort": dict(flags=pt.PT_COMPACT | pt.PT_SORT_MATERIAL),
          "nocompact": dict(flags=0),
          "bvh+aa+lens": dict(flags=pt.PT_COMPACT | pt.PT_MESH_BVH | pt.PT_AA_JITTER, lens=(0.2, 9.0))}[pipeline]
    if pipeline == "bvh+aa+lens":
        tris = pt.meshes.uv_sphere(center=(1.5, 3.0, 1.0), radius=1.5, n_lat=16, n_lon=32)
        geoms, tris, meshes = pt.meshes.add_mesh(s["geoms"], tris, material_id=2)
        scene = pt.Scene(geoms, s["materials"], s["camera"], s["depth"], triangles=tris, meshes=meshes)
    else:
        scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    n = scene.resolution[0] * scene.resolution[1]
    monkeypatch.delenv("PTMI355_XCHG", raising=False)
    monkeypatch.delenv("PTMI355_DEVICES", raising=False)
    want, live, counters, rgba, ndev, transport = run_calls(pt, scene, n, **kw)
    assert ndev == 1 and transport == "none"
    assert np.isfinite(want[-1]).all() and want[-1].max() > 0
    for devs in device_lists():
        for strip in (8, 5):                                     # 64 rows: 8 strips of 8 (even), 13 of 5 (uneven)
            # (a -DPT_EXPERIMENTS build: the even runs issue the asynchronous batches' exchanges from the caller's thread as
            # rounds 1-3 did; the shipped library always uses its exchange thread -- pt_multi.hpp, Exchanger)
            monkeypatch.setenv("PTMI355_XCHG_THREAD", "1" if strip == 5 else "0")
            got, glive, gcounters, grgba, gdev, gtransport = run_calls(pt, scene, n, devices=devs, tile=(0, 1, strip), **kw)
            assert gdev == len(devs)
            assert gtransport == ("rccl" if len(set(devs)) == len(devs) else "peer")
            for a, b in zip(got, want):
                assert a.tobytes() == b.tobytes(), (devs, strip)
            assert glive == live and gcounters == counters, (devs, strip)       # the tiles' ray counts add up to the frame's
            assert grgba.tobytes() == rgba.tobytes()


def test_c2_frame_over_two_contexts(pt, scenes):
    """The benchmark configuration: 800 x 800, depth 8, a batch of 8 + one pathtrace() with the host image."""
    s = scenes["cornell"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    n = 800 * 800

    def run(**kw):
        pt.pathtraceInit(scene, flags=pt.PT_COMPACT, max_batch=8, **kw)
        img = np.zeros((n, 3), dtype=np.float32)
        pt.trace_batch(1, 8, img)
        one = pt.pathtrace(None, 0, 9).copy()
        rays = pt.total_rays()
        pt.pathtraceFree()
        return img, one, rays

    want = run()
    devs = [0, 1] if gpu_count() >= 2 else [0, 0]
    got = run(devices=devs)
    assert got[2] == want[2], "rays traced: %d over two contexts, %d over one" % (got[2], want[2])
    for name, a, b in (("host image after the batch of 8", got[0], want[0]), ("device image after the single call", got[1], want[1])):
        bad = np.flatnonzero((a.view(np.uint32) != b.view(np.uint32)).reshape(n, 3).any(axis=1))
        assert bad.size == 0, "%s: %d pixels differ, rows %d..%d (first: pixel %d, %r against %r)" % (
            name, bad.size, bad[0] // 800, bad[-1] // 800, bad[0], a.reshape(n, 3)[bad[0]], b.reshape(n, 3)[bad[0]])


def test_pathtrace_per_call_several_contexts_no_exchange(pt, scenes, monkeypatch):
    """pathtrace() with the page-locked host image over several contexts: every context's launch writes its own tile's
    pixels into the caller's image (registered once for all devices) -- no pack, no exchange, no copy of the frame.  The
    host image after every call, the device frame whenever it is asked for (it is brought up to date by one exchange),
    and everything in between -- batches that do exchange, pt_clear_image, pt_set_image, a second host buffer -- equal
    the single-context run and the run with the direct path switched off."""
    s = scenes["cornell"]
    cam = s["camera"].copy()
    cam["resolution"][0] = (400, 300)
    yscaled = np.tan(np.float32(cam["fov"][0][1]) * np.float32(np.pi / 180))
    xscaled = np.float32(yscaled * np.float32(400) / np.float32(300))
    cam["pixelLength"][0] = (np.float32(2 * xscaled / np.float32(400)), np.float32(2 * yscaled / np.float32(300)))
    scene = pt.Scene(s["geoms"], s["materials"], cam, s["depth"])
    n = 400 * 300
    L = pt.library()

    def run(env=(), **kw):
        for k, v in env:
            monkeypatch.setenv(k, v)
        pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_PIN_IMAGE, max_batch=4, pin_image=False, **kw)
        a = np.full((n, 3), -4.0, dtype=np.float32)
        b = np.full((n, 3), -6.0, dtype=np.float32)
        out = []

        def call(buf, it, check_device=False):
            assert L.pt_trace(None, 0, it, buf.ctypes.data) == 0
            out.append(buf.tobytes())
            if check_device:
                assert pt.get_image(n).tobytes() == buf.tobytes(), it

        call(a, 1); call(a, 2); call(a, 3, True)
        pt.trace_batch_async(4, 4)                                   # exchanges
        call(a, 8); call(a, 9, True)
        img = np.zeros((n, 3), dtype=np.float32)
        pt.trace_batch(10, 2, img)                                   # synchronous batch: frame copied to the host
        out.append(img.tobytes())
        call(b, 12); call(a, 13); call(b, 14, True)
        pt.clear_image()
        call(a, 1); call(a, 2, True)
        pt.set_image(np.frombuffer(out[2], dtype=np.float32).reshape(n, 3).copy())
        call(a, 4); call(a, 5, True)
        out.append(tuple(int(v) for v in pt.counters()))
        out.append(pt.tonemap(n, 5).tobytes())
        pt.pathtraceFree()
        for k, _ in env:
            monkeypatch.delenv(k)
        return out

    monkeypatch.delenv("PTMI355_XCHG", raising=False)
    monkeypatch.delenv("PTMI355_DEVICES", raising=False)
    want = run()
    for devs in ([0, 0], [0, 0, 0]) if gpu_count() < 2 else ([0, 1], [0, 0], [0, 1, 0]):
        for strip in (8, 7):
            assert run(devices=devs, tile=(0, 1, strip)) == want, (devs, strip)
    if pt.has_experiments():          # (a -DPT_EXPERIMENTS build: pack + exchange + copy instead of the launch-written host frame)
        assert run((("PTMI355_MULTI_DIRECT", "0"),), devices=[0, 0], tile=(0, 1, 8)) == want


def test_rccl_calls_with_a_communicator_of_one(pt, scenes, monkeypatch):
    """ncclCommInitAll / ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd / ncclCommDestroy executed from the
    library on this box's one GPU: the one context sends its packed tile to itself and unpacks it into the frame."""
    s = scenes["cornell_64"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    n = 64 * 64
    monkeypatch.delenv("PTMI355_XCHG", raising=False)
    want = run_calls(pt, scene, n, flags=pt.PT_COMPACT)
    monkeypatch.setenv("PTMI355_XCHG", "rccl")
    got = run_calls(pt, scene, n, flags=pt.PT_COMPACT, devices=[0])
    assert got[4] == 1 and got[5] == "rccl"
    for a, b in zip(got[0], want[0]):
        assert a.tobytes() == b.tobytes()
    assert got[1] == want[1] and got[2] == want[2]
    # two contexts on one device cannot use RCCL: refused with a message, not a crash
    L = pt.library()
    with pytest.raises(pt.PtError, match="own device"):
        pt.pathtraceInit(scene, devices=[0, 0])
    assert L.pt_num_devices() == 0


def test_rccl_library_named_by_the_environment(tmp_path):
    """PTMI355_RCCL_LIB names the RCCL library the in-library exchange dlopens (a deployment whose librccl is not on the
    loader path).  Fresh processes (the library is loaded once per process): the variable pointing at this image's
    library, and at a file that does not exist (the default names are tried next) -- the exchange runs over RCCL both times
    and the frame is the plain single-device frame."""
    code = """
import os, sys, hashlib
import numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as ge
pt = ge.load_package()
z = np.load(os.path.join(%r, "tests", "golden", "scenes.npz"))
g = lambda k: z["cornell_64__" + k]
scene = pt.Scene(g("geoms"), g("materials"), g("camera"), int(g("depth")))
devs = [0] if os.environ.get("PTMI355_XCHG") else None
pt.pathtraceInit(scene, flags=pt.PT_COMPACT, devices=devs, max_batch=2)
pt.trace_batch(1, 2, None)
img = pt.pathtrace(None, 0, 3)
print("RESULT", pt.exchange_transport(), hashlib.md5(img.tobytes()).hexdigest())
pt.pathtraceFree()
""" % (ROOT, ROOT)

    def run(env):
        e = {k: v for k, v in os.environ.items() if k not in ("PTMI355_XCHG", "PTMI355_RCCL_LIB", "PTMI355_DEVICES")}
        e.update(env)
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=e)
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
        return [l for l in p.stdout.splitlines() if l.startswith("RESULT")][-1].split()[1:]

    plain = run({})
    assert plain[0] == "none"
    real = os.path.realpath("/opt/rocm/lib/librccl.so.1")
    assert os.path.exists(real)
    assert run({"PTMI355_XCHG": "rccl", "PTMI355_RCCL_LIB": real}) == ["rccl", plain[1]]
    assert run({"PTMI355_XCHG": "rccl", "PTMI355_RCCL_LIB": str(tmp_path / "no_such_librccl.so")}) == ["rccl", plain[1]]


@pytest.mark.skipif(gpu_count() < 2, reason="needs two GPUs")
def test_two_gpus_over_rccl(pt, scenes):
    s = scenes["cornell_64"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    n = 64 * 64
    want = run_calls(pt, scene, n, flags=pt.PT_COMPACT)
    for devs in ([0, 1], [1, 0], list(range(gpu_count()))):
        got = run_calls(pt, scene, n, flags=pt.PT_COMPACT, devices=devs)
        assert got[5] == "rccl"
        for a, b in zip(got[0], want[0]):
            assert a.tobytes() == b.tobytes(), devs


def test_devices_from_the_environment_and_the_reference_host(pt, po, scenes, tmp_path, monkeypatch):
    """PTMI355_DEVICES spreads a host that knows one device (the reference's: cudaGLSetGLDevice(0), preview.cpp:107)
    over several: the reference's own host through the shim (oracle/_ref/refhost) and ptbench write the PNG of the
    single-device run."""
    s = scenes["cornell_64"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    monkeypatch.setenv("PTMI355_DEVICES", "0,0,0")
    pt.pathtraceInit(scene)
    assert pt.num_devices() == 3
    img = pt.pathtrace(None, 0, 1).copy()
    pt.pathtraceFree()
    monkeypatch.delenv("PTMI355_DEVICES")
    pt.pathtraceInit(scene)
    assert pt.num_devices() == 1
    assert pt.pathtrace(None, 0, 1).tobytes() == img.tobytes()
    pt.pathtraceFree()
    # a session that is itself one tile of K processes keeps its one device
    monkeypatch.setenv("PTMI355_DEVICES", "0,0")
    pt.pathtraceInit(scene, tile=(1, 2, 8))
    assert pt.num_devices() == 1
    pt.pathtraceFree()
    monkeypatch.setenv("PTMI355_DEVICES", "0,x")
    with pytest.raises(pt.PtError, match="device list"):
        pt.pathtraceInit(scene)
    monkeypatch.delenv("PTMI355_DEVICES")

    txt = open(os.path.join(ROOT, "scenes", "cornell.txt")).read().replace("RES         800 800", "RES         64 64")
    txt = re.sub(r"(?m)^ITERATIONS\s+\d+", "ITERATIONS  5", txt)
    from PIL import Image
    ref = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"])
    for it in range(1, 6):
        ref.iterate(it)
    want = pt.image_to_rgb8(ref.image, 64, 64, 5.0).tobytes()
    bench = pt.build_ptbench()
    scene_file = tmp_path / "c64.txt"
    scene_file.write_text(txt)
    for extra in (["--gpus", "1"], ["--devices", "0,0"], ["--devices", "0,0,0", "--strip-rows", "3", "--batch", "2"]):
        p = subprocess.run([bench, str(scene_file), "--out", str(tmp_path / "ptb")] + extra, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        assert ("on %d device(s)" % len(extra[1].split(",")) in p.stdout) or extra[0] == "--gpus"
        got = np.asarray(Image.open(str(tmp_path / "ptb.5samp.png")).convert("RGB"), dtype=np.uint8)
        assert got.tobytes() == want, extra
    exe = os.path.join(ROOT, "oracle", "_ref", "refhost")
    if os.path.exists(exe):
        txt2 = re.sub(r"(?m)^FILE\s+\S+", "FILE        %s" % str(tmp_path / "refhost"), txt)
        scene_file.write_text(txt2)
        env = dict(os.environ, PTMI355_DEVICES="0,0")
        p = subprocess.run([exe, str(scene_file), "T0"], capture_output=True, text=True, timeout=300, env=env)
        assert p.returncode == 0, p.stdout + p.stderr
        got = np.asarray(Image.open(str(tmp_path / "refhost.T0.5samp.png")).convert("RGB"), dtype=np.uint8)
        assert got.tobytes() == want


def test_what_a_multi_device_session_refuses(pt, scenes):
    s = scenes["cornell_64"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    with pytest.raises(pt.PtError, match="tile_count"):
        pt.pathtraceInit(scene, devices=[0, 0], tile=(0, 2, 8))
    with pytest.raises(pt.PtError, match="pt_init: device"):
        pt.pathtraceInit(scene, devices=[0, 7 if gpu_count() <= 7 else 99])
    # more contexts than strips: some tile owns no rows
    with pytest.raises(pt.PtError, match="no rows"):
        pt.pathtraceInit(scene, devices=[0] * 3, tile=(0, 1, 32))
    pt.pathtraceInit(scene, devices=[0, 0])
    with pytest.raises(pt.PtError, match="stepping interface"):
        pt.trace_begin(1, 1)
    # camera re-read per call (pathtrace.cu:285-286) reaches every context; depth may change
    scene.traceDepth = 3
    a = pt.pathtrace(None, 0, 1).copy()
    pt.pathtraceFree()
    pt.pathtraceInit(scene)
    assert pt.pathtrace(None, 0, 1).tobytes() == a.tobytes()
    pt.pathtraceFree()
    pt.pathtraceFree()


def test_inproc_bench_line(pt):
    """bench.py --gpus 2 --inproc (one process, the library's own tiling) prints the contract's JSON line and the
    image of the single-device run."""
    import json
    args = ["--steps", "2", "--warmup", "1", "--batch", "2", "--no-roofline", "--no-cpu-baseline", "--digest", "--sub-iters", "8"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + ["--batch", "4"], capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    a = json.loads(one.stdout.strip().splitlines()[-1])
    same = [] if gpu_count() >= 2 else ["--same-device"]
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--inproc"] + same + args,
                         capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    b = json.loads(two.stdout.strip().splitlines()[-1])
    assert b["n_gpus"] == 2 and b["scaling"] == "weak" and b["metric"] == a["metric"] and b["unit"] == "Mrays/s"
    assert b["config"]["exchanges_per_step"] == 1 and "in the library" in b["config"]["sharding"]
    # weak scaling: 2 x 2 iterations per step on two tiles == 4 per step on the whole frame
    assert b["config"]["rays_per_step"] == a["config"]["rays_per_step"]
    assert b["image_md5"] == a["image_md5"]
    # the sub-measurements of every N > 1 line: an exchange after every iteration (the library's exchange thread) and strong scaling
    assert b["config"]["per_iteration_exchange"]["mrays_per_s"] > 10 and b["config"]["strong"]["mrays_per_s"] > 10
