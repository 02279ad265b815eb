"""GPU parity through the hosts: ptbench (the headless host) and the REFERENCE'S OWN host code through the shim
(oracle/_ref/refhost), images compared with the oracle's pixel for pixel."""
import hashlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import __graft_entry__ as ge  # noqa: E402,F401
from gpu_common import pt, launch_plan, bits, rel_l2, assert_paths_equal, _resized, _after  # noqa: E402,F401

pytestmark = pytest.mark.gpu


def test_ptbench_headless_host(pt, po, scenes, tmp_path):
    """The C++ headless host (host/ptbench.cpp = main.cpp/runCuda without GLFW): scene file in, PNG out;
    the PNG equals the oracle's image pushed through the same saveImage pipeline."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "scenes", "cornell.txt")).read().replace("RES         800 800", "RES         64 64")
    scene_file = tmp_path / "cornell64.txt"
    scene_file.write_text(txt)
    exe = pt.build_ptbench()
    p = subprocess.run([exe, str(scene_file), "--iters", "5", "--batch", "2", "--out", str(tmp_path / "r")],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "Mrays/s" in p.stdout
    from PIL import Image
    got = np.asarray(Image.open(str(tmp_path / "r.5samp.png")).convert("RGB"), dtype=np.uint8)
    s = scenes["cornell_64"]
    ref = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"])
    for it in range(1, 6):
        ref.iterate(it)
    want = pt.image_to_rgb8(ref.image, 64, 64, 5.0)
    assert got.tobytes() == want.tobytes()


def test_reference_host_through_the_shim(pt, po, scenes, tmp_path):
    """The REFERENCE host -- its own scene.cpp / utilities.cpp / image.cpp / stb.cpp and the runCuda sequence of
    main.cpp:101-147 (free before init, per-call camera re-read, scene->state.image refreshed by every pathtrace()) --
    linked against host/pathtrace_shim.cpp + libptmi355.so (oracle/_ref/refhost, built in the build container by
    oracle/Makefile, shipped with the snapshot): the PNG its saveImage() writes decodes to the pixels of ptbench's PNG
    and of the oracle's image pushed through the same pipeline."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "oracle", "_ref", "refhost")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/refhost is built where /root/reference exists")
    txt = open(os.path.join(root, "scenes", "cornell.txt")).read().replace("RES         800 800", "RES         64 64")
    import re
    txt = re.sub(r"(?m)^ITERATIONS\s+\d+", "ITERATIONS  5", txt)
    txt = re.sub(r"(?m)^FILE\s+\S+", "FILE        %s" % str(tmp_path / "refhost"), txt)
    scene_file = tmp_path / "cornell64.txt"
    scene_file.write_text(txt)
    p = subprocess.run([exe, str(scene_file), "T0"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    from PIL import Image
    got = np.asarray(Image.open(str(tmp_path / "refhost.T0.5samp.png")).convert("RGB"), dtype=np.uint8)
    bench = pt.build_ptbench()
    p = subprocess.run([bench, str(scene_file), "--out", str(tmp_path / "ptb")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    mine = np.asarray(Image.open(str(tmp_path / "ptb.5samp.png")).convert("RGB"), dtype=np.uint8)
    assert got.tobytes() == mine.tobytes()
    s = scenes["cornell_64"]
    ref = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"])
    for it in range(1, 6):
        ref.iterate(it)
    assert got.tobytes() == pt.image_to_rgb8(ref.image, 64, 64, 5.0).tobytes()


def _read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = [int(v) for v in f.readline().split()]
        scale = float(f.readline())
        data = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4")
    return data.reshape(h, w, 3)


def test_ptbench_tiles(pt, tmp_path):
    """`ptbench --tile R/K`: K host processes (one per GPU in production) render one frame between them; their raw
    sums add up -- exactly, a sum with zeros -- to the single-process image."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "scenes", "cornell.txt")).read().replace("RES         800 800", "RES         96 80")
    scene_file = tmp_path / "c.txt"
    scene_file.write_text(txt)
    exe = pt.build_ptbench()

    def render(name, *extra):
        p = subprocess.run([exe, str(scene_file), "--iters", "3", "--batch", "2", "--pfm", "--out", str(tmp_path / name)] + list(extra),
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        return _read_pfm(str(tmp_path / (name + ".3samp.pfm")))

    whole = render("whole")
    parts = [render("t%d" % k, "--tile", "%d/3" % k, "--strip-rows", "8") for k in range(3)]
    assert (parts[0] + parts[1] + parts[2]).tobytes() == whole.tobytes()
    assert all((p != 0).any() and (p == 0).any() for p in parts)


def test_ptbench_mesh_scene_hierarchy_and_camera_options(pt, tmp_path):
    """ptbench on a scene file with a `mesh file.obj` object: --bvh gives the PNG of the loop over every triangle,
    byte for byte; --aa / --lens render (and change the image)."""
    import os
    import subprocess
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tris = pt.meshes.uv_sphere(center=(0.0, 0.0, 0.0), radius=1.0, n_lat=24, n_lon=48)
    with open(tmp_path / "ball.obj", "w") as f:
        for t in tris:
            for k in ("v0", "v1", "v2"):
                f.write("v %.9g %.9g %.9g\n" % tuple(t[k]))
        for i in range(len(tris)):
            f.write("f %d %d %d\n" % (3 * i + 1, 3 * i + 2, 3 * i + 3))
    txt = open(os.path.join(root, "scenes", "cornell.txt")).read().replace("RES         800 800", "RES         96 96")
    n_obj = sum(1 for line in txt.splitlines() if line.startswith("OBJECT "))
    txt = txt.rstrip("\n") + "\n\nOBJECT %d\nmesh ball.obj\nmaterial 2\nTRANS 2 3 1\nROTAT 0 30 0\nSCALE 1.5 1.5 1.5\n" % n_obj
    scene_file = tmp_path / "cornell_mesh.txt"
    scene_file.write_text(txt)
    exe = pt.build_ptbench()

    def render(tag, *opts):
        p = subprocess.run([exe, str(scene_file), "--iters", "4", "--batch", "2", "--out", str(tmp_path / tag)] + list(opts),
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        assert "%d triangles" % len(tris) in p.stdout
        return np.asarray(Image.open(str(tmp_path / (tag + ".4samp.png"))).convert("RGB"), dtype=np.uint8)

    loop, bvh = render("loop"), render("bvh", "--bvh")
    assert loop.tobytes() == bvh.tobytes()
    assert (loop[30:70, 55:90] != loop[0, 0]).any()
    dof = render("dof", "--bvh", "--aa", "--lens", "0.3", "9")
    assert dof.shape == loop.shape and (dof != loop).any()
