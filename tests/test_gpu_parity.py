"""GPU parity: the HIP path, called through the C-ABI, against the CPU oracle on
the same inputs.  Everything here is bit-exact (float radiance included: the
library is built without FMA contraction and shares the oracle's sin/cos
definition), which is stricter than the north star's 1e-4 relative L2."""
import hashlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import __graft_entry__ as ge  # noqa: E402,F401
from gpu_common import pt, launch_plan, bits, rel_l2, assert_paths_equal, _resized, _after  # noqa: E402,F401

pytestmark = pytest.mark.gpu


def test_raygen(pt, po, scenes, golden):
    for name in ("cornell_64", "cornell"):
        s = scenes[name]
        scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
        pt.pathtraceInit(scene)
        pt.trace_begin(1, 1)
        n = scene.resolution[0] * scene.resolution[1]
        paths, live = pt.export_paths(n)
        assert live == n
        want = po.generate_rays(s["camera"], s["depth"])
        assert paths.tobytes() == want.tobytes()
        pt.pathtraceFree()
    assert hashlib.md5(paths.tobytes()).hexdigest() == str(golden["raygen"]["md5_800"])


def test_intersect_kernel_vs_golden_rays(pt, po, scenes, golden):
    """computeIntersections on the adversarial ray sets of tests/golden/geomtests.npz
    (inside-origin, grazing, axis-parallel +-inf slabs, un-normalised directions)."""
    z = golden["geomtests"]
    s = scenes["cornell"]
    rays = np.concatenate([z["rays_%d" % i] for i in range(7)])
    paths = np.zeros(len(rays), dtype=pt.PATH_DT)
    paths["origin"], paths["direction"] = rays[:, :3], rays[:, 3:]
    paths["color"] = 1.0
    paths["pixelIndex"] = np.arange(len(rays))
    paths["remainingBounces"] = 8
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    pt.pathtraceInit(scene)
    got, got_out = pt.intersect_once(paths)
    want, want_out = po.compute_intersections(paths.view(po.PATH_DT), s["geoms"])
    assert (bits(got["t"]) == bits(want["t"])).all()
    assert (bits(got["normal"]) == bits(want["normal"])).all()
    assert (got["materialId"] == want["materialId"]).all()
    hit = want["t"] > 0
    assert hit.sum() > 1000
    assert (got_out[hit] == want_out[hit]).all()
    # each geom alone, including the rotated / non-uniformly scaled extras
    for i, gi in enumerate(z["geom_index"]):
        geom = s["geoms"][gi:gi + 1] if gi < 100 else z["extra_geoms"][gi - 100:gi - 99]
        geom = geom.copy()
        geom["materialid"] = 0
        sc1 = pt.Scene(geom, s["materials"], s["camera"], s["depth"])
        pt.pathtraceInit(sc1)
        r = z["rays_%d" % i]
        p = np.zeros(len(r), dtype=pt.PATH_DT)
        p["origin"], p["direction"] = r[:, :3], r[:, 3:]
        got, _ = pt.intersect_once(p)
        ref_t = z["out_%d" % i][:, 0]
        ref_n = z["out_%d" % i][:, 4:7]
        hit = ref_t > 0
        assert (bits(got["t"][hit]) == bits(ref_t[hit])).all(), "geom %d" % gi
        assert (bits(got["normal"][hit]) == bits(np.ascontiguousarray(ref_n[hit]))).all(), "geom %d" % gi
        assert (got["t"][~hit] == -1.0).all()
    pt.pathtraceFree()


def test_first_bounce_intersections_c2(pt, po, scenes, golden):
    """SURVEY section 7 minimum slice: ShadeableIntersection[N] of Cornell 800x800 bounce 0."""
    s = scenes["cornell"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_UNFUSED)
    pt.trace_begin(1, 1)
    pt.trace_bounce(0)
    isx, _ = pt.export_intersections(640000)
    assert hashlib.md5(isx.tobytes()).hexdigest() == str(golden["fakeshade"]["md5_isect800"])
    pt.pathtraceFree()


@pytest.mark.parametrize("flags_name", ["fused", "unfused", "nocompact", "sort", "sort2", "cache"])
@pytest.mark.parametrize("scene_name", ["cornell_64", "cornell_glass_64", "cornell_diffuse_64"])
def test_bounce_by_bounce(pt, po, scenes, scene_name, flags_name):
    """Every bounce: live count, compacted pixelIndex sequence and full path state bit-exact."""
    s = scenes[scene_name]
    flags = {"fused": pt.PT_COMPACT, "unfused": pt.PT_COMPACT | pt.PT_UNFUSED, "nocompact": 0,
             "sort": pt.PT_COMPACT | pt.PT_SORT_MATERIAL, "cache": pt.PT_COMPACT | pt.PT_CACHE_FIRST,
             # sort: survivors placed by material inside the fused kernel; sort2: the two-kernel form (intersections
             # materialised, k_sort_hist + k_shade_sorted_w)
             "sort2": pt.PT_COMPACT | pt.PT_SORT_MATERIAL | pt.PT_UNFUSED}[flags_name]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    n = scene.resolution[0] * scene.resolution[1]
    pt.pathtraceInit(scene, flags=flags)
    oflags = (po.F_COMPACT if flags & pt.PT_COMPACT else 0) | (po.F_SORT if flags & pt.PT_SORT_MATERIAL else 0)
    ref = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"], flags=oflags, trig=po.TRIG_SHARED)
    for it in (1, 2, 3):
        snaps = []
        st = ref.iterate(it, snapshots=snaps)
        pt.trace_begin(it, 1)
        for snap in snaps:
            d = snap["depth"]
            n_live = pt.trace_bounce(d)
            paths, live = pt.export_paths(n)
            if flags & pt.PT_COMPACT:
                assert n_live == snap["n_live"] == live
                want = ref.paths if d == len(snaps) - 1 else None
                # oracle snapshot is taken after compaction: its live prefix is the pool
                assert_paths_equal(paths, _after(snaps, d, ref), live)
            else:
                alive = paths["pixelIndex"] >= 0
                wp = _after(snaps, d, ref)
                walive = wp["remainingBounces"] > 0
                assert (alive == walive[:len(alive)]).all()
                assert_paths_equal(paths[alive], wp[:len(alive)][alive], int(alive.sum()))
        for d in range(len(snaps), s["depth"]):
            pt.trace_bounce(d)
        pt.trace_end()
        gs = pt.get_stats()
        assert list(gs.live[:s["depth"]]) == list(st.live[:s["depth"]])
        assert gs.rays == st.rays
        img = pt.get_image(n)
        assert img.tobytes() == ref.image.tobytes()
        assert rel_l2(img, ref.image) <= 1e-4          # the north star's stated tolerance
    pt.pathtraceFree()


@pytest.mark.parametrize("flags_name", ["fused", "unfused", "nocompact", "sort", "sort2", "sort_nocompact", "cache"])
def test_c2_full_iteration(pt, po, scenes, golden, flags_name):
    """Config C2 (800x800, depth 8): image, live counts and compaction order vs golden + oracle."""
    z = golden["completion"]
    s = scenes["cornell"]
    flags = {"fused": pt.PT_COMPACT, "unfused": pt.PT_COMPACT | pt.PT_UNFUSED, "nocompact": 0,
             "sort": pt.PT_COMPACT | pt.PT_SORT_MATERIAL, "sort_nocompact": pt.PT_SORT_MATERIAL,
             "sort2": pt.PT_COMPACT | pt.PT_SORT_MATERIAL | pt.PT_UNFUSED,
             "cache": pt.PT_COMPACT | pt.PT_CACHE_FIRST}[flags_name]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    pt.pathtraceInit(scene, flags=flags)
    for it in (1, 2):
        img = pt.pathtrace(None, 0, it)
        gs = pt.get_stats()
        assert (np.array(gs.live[:8]) == z["shared__cornell__live"][it - 1]).all()
        assert gs.rays == z["shared__cornell__rays"][it - 1]
        assert hashlib.md5(img.tobytes()).hexdigest() == str(z["shared__cornell__img_md5"][it - 1])
    pt.pathtraceFree()


def test_c2_forty_iterations_in_batches(pt, po, scenes):
    """The bench workload at length: 40 iterations of C2 traced as batches of 16 + 16 + 8 paths pools against the
    oracle's 40 sequential iterations (all host threads): total rays and every pixel of the running sum."""
    import os
    s = scenes["cornell"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    n = scene.resolution[0] * scene.resolution[1]
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT, max_batch=16)
    img = np.zeros((n, 3), dtype=np.float32)
    pt.trace_batch(1, 16, None)
    pt.trace_batch(17, 16, None)
    pt.trace_batch(33, 8, img)
    rays = pt.get_stats().total_rays
    pt.pathtraceFree()
    ref = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"], trig=po.TRIG_SHARED)
    want = 0
    for it in range(1, 41):
        want += ref.iterate(it, threads=os.cpu_count() or 8).rays
    assert rays == want
    assert img.tobytes() == ref.image.tobytes()


def test_c2_compaction_order_hash(pt, po, scenes, golden):
    z = golden["completion"]
    s = scenes["cornell"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT)
    for it in (1, 2):                                             # the golden fixture holds both iterations
        pt.trace_begin(it, 1)
        for d in range(8):
            live = pt.trace_bounce(d)
            paths, n_live = pt.export_paths(640000)
            assert n_live == live
            seq = np.ascontiguousarray(paths["pixelIndex"])       # keep alive across the C call
            h = po.lib().pto_fnv1a_i32(seq.ctypes.data, 4, live)
            assert h == int(z["shared__cornell__seq_hash"][it - 1][d]), "iteration %d bounce %d" % (it, d)
        pt.trace_end()
    pt.pathtraceFree()


@pytest.mark.parametrize("flags_name", ["compact", "sort", "sort2"])
def test_c3_full_size(pt, po, scenes, flags_name):
    """Config C3 at its full size -- glass ball, 1280x720, depth 16, material sort on / off: live counts, the
    compacted pixelIndex sequence after every bounce (its hash) and the image of two iterations equal the oracle's
    (threaded: one bounce of 921 600 paths at a time on 8 threads)."""
    s = scenes["cornell_glass"]
    cam = s["camera"]
    assert tuple(cam[0]["resolution"]) == (1280, 720) and s["depth"] == 16
    flags = pt.PT_COMPACT | {"compact": 0, "sort": pt.PT_SORT_MATERIAL, "sort2": pt.PT_SORT_MATERIAL | pt.PT_UNFUSED}[flags_name]
    oflags = po.F_COMPACT | (po.F_SORT if flags_name != "compact" else 0)
    n = 1280 * 720
    scene = pt.Scene(s["geoms"], s["materials"], cam, s["depth"])
    ref = po.Tracer(s["geoms"], s["materials"], cam, s["depth"], flags=oflags, trig=po.TRIG_SHARED)
    pt.pathtraceInit(scene, flags=flags)
    for it in (1, 2):
        st = ref.iterate(it) if flags_name != "compact" else ref.iterate(it, threads=8)
        pt.trace_begin(it, 1)
        for d in range(s["depth"]):
            live = pt.trace_bounce(d)
            if d < st.bounces:
                paths, n_live = pt.export_paths(n)
                assert n_live == live
                seq = np.ascontiguousarray(paths["pixelIndex"])
                assert po.lib().pto_fnv1a_i32(seq.ctypes.data, 4, live) == st.seq_hash[d], "iteration %d bounce %d" % (it, d)
        pt.trace_end()
        gs = pt.get_stats()
        assert list(gs.live[:16]) == list(st.live[:16]) and gs.rays == st.rays
    assert pt.get_image(n).tobytes() == ref.image.tobytes()
    pt.pathtraceFree()


def test_statistical_tier_256spp(pt, golden, scenes):
    """SURVEY section 4 / BASELINE section 5: 800x800 Cornell at 256 spp against the reference's only rendered
    artefact (img/REFERENCE_cornell.5000samp.png, committed as 16x16-pooled means): x-flipped, clamped, quantised as
    savePNG does, the ball and its reflection / shadow masked (the PNG's ball is matte), relative L2 <= 0.05."""
    s = scenes["cornell"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    n = 800 * 800
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT, max_batch=64)
    img = np.zeros((n, 3), dtype=np.float32)
    for k in range(4):
        pt.trace_batch(1 + 64 * k, 64, img)
    pt.pathtraceFree()
    img = (img / 256.0).reshape(800, 800, 3)[:, ::-1, :]               # saveImage x-flip (main.cpp:87)
    img = np.floor(np.clip(img, 0, 1) * 255.0) / 255.0                 # savePNG (image.cpp:22-39)
    pooled = img.reshape(50, 16, 50, 16, 3).mean(axis=(1, 3))
    want = golden["png_stat"]["pooled"]
    mask = np.ones((50, 50), dtype=bool)
    mask[22:40, 12:32] = False
    num = np.sqrt(((pooled - want)[mask] ** 2).sum())
    den = np.sqrt((want[mask] ** 2).sum())
    assert num / den <= 0.05, num / den


def test_batch_equals_sequential(pt, scenes):
    """pt_trace_batch(iter0, count) == count sequential pathtrace calls, bit for bit."""
    s = scenes["cornell_glass_64"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    n = scene.resolution[0] * scene.resolution[1]
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT, max_batch=1)
    for it in range(1, 8):
        seq = pt.pathtrace(None, 0, it).copy()
    rays_seq = pt.get_stats().total_rays
    pt.pathtraceFree()
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT, max_batch=4)
    img = np.zeros((n, 3), dtype=np.float32)
    pt.trace_batch(1, 4, img)
    pt.trace_batch(5, 3, img)
    assert pt.get_stats().total_rays == rays_seq
    assert img.tobytes() == seq.tobytes()
    pt.pathtraceFree()


def test_fake_shader_as_is(pt, scenes, golden):
    """The reference exactly as shipped: one bounce + shadeFakeMaterial + sendImageToPBO."""
    z = golden["fakeshade"]
    s = scenes["cornell_64"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    pt.pathtraceInit(scene, flags=pt.PT_FAKE_SHADER)
    for it in (1, 2, 3):
        img = pt.pathtrace(None, 0, it)
    assert img.tobytes() == z["img64"].tobytes()
    assert pt.tonemap(64 * 64, 3).tobytes() == z["pbo64"].tobytes()
    pt.pathtraceFree()
    s = scenes["cornell"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    pt.pathtraceInit(scene, flags=pt.PT_FAKE_SHADER)
    for it in (1, 2):
        img = pt.pathtrace(None, 0, it)
    assert hashlib.md5(img.tobytes()).hexdigest() == str(z["md5_img800"])
    assert hashlib.md5(pt.tonemap(640000, 2).tobytes()).hexdigest() == str(z["md5_pbo800"])
    pt.pathtraceFree()


def test_lifecycle_and_errors(pt, scenes):
    s = scenes["cornell_64"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    pt.pathtraceFree()                       # before init (main.cpp:126)
    pt.pathtraceInit(scene)
    pt.pathtraceInit(scene)                  # re-init on camera move (main.cpp:125-128)
    a = pt.pathtrace(None, 0, 1).copy()
    pt.pathtraceFree()
    pt.pathtraceFree()
    pt.pathtraceInit(scene)
    b = pt.pathtrace(None, 0, 1).copy()
    assert a.tobytes() == b.tobytes()
    with pytest.raises(pt.PtError):
        pt.trace_batch(1, 2)                 # count > max_batch
    with pytest.raises(pt.PtError):
        pt.trace_bounce(0)                   # stepping without begin
    with pytest.raises(pt.PtError):
        pt.export_intersections(10)          # not materialised in fused mode
    pt.pathtraceFree()
    # inconsistent mesh tables are refused, not traced: a range past the triangle array (also when first + count
    # overflows 32 bits), two meshes on one geom (in both mesh modes)
    tris = pt.meshes.uv_sphere(n_lat=4, n_lon=6)
    geoms, tris, meshes = pt.meshes.add_mesh(s["geoms"], tris, material_id=1)
    for bad in ("range", "overflow", "duplicate"):
        m = np.concatenate([meshes, meshes]) if bad == "duplicate" else meshes.copy()
        if bad == "range":
            m["triangle_count"][0] = len(tris) + 1
        if bad == "overflow":
            m["first_triangle"][0], m["triangle_count"][0] = 2 ** 31 - 4, 8
        for flags in (pt.PT_COMPACT, pt.PT_COMPACT | pt.PT_MESH_BVH):
            with pytest.raises(pt.PtError, match="mesh"):
                pt.pathtraceInit(pt.Scene(geoms, s["materials"], s["camera"], s["depth"], triangles=tris, meshes=m), flags=flags)


def test_c1_as_stated(pt, po, scenes, golden):
    """BASELINE configs[0]: scenes/cornell_diffuse.txt as the reference loader reads it -- the Cornell box with the
    sphere made diffuse, 400 x 400, 1 spp, depth 4 -- on the GPU against the oracle's single-thread loop and against
    the image the REFERENCE'S OWN headers produce for it (tests/golden/c1.npz; tests/test_oracle_golden.py holds the
    oracle against the same fixture on the CPU)."""
    s = scenes["cornell_diffuse"]
    assert tuple(s["camera"][0]["resolution"]) == (400, 400) and s["depth"] == 4
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    pt.pathtraceInit(scene)
    img = pt.pathtrace(None, 0, 1).copy()
    live = list(pt.get_stats().live[:4])
    pt.pathtraceFree()
    ref = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"])
    st = ref.iterate(1)
    z = golden["c1"]
    assert live == list(st.live[:4]) == list(z["live"][0]) and live[0] == 160000
    assert img.tobytes() == ref.image.tobytes()
    assert hashlib.md5(img.tobytes()).hexdigest() == str(z["img_md5"])


@pytest.mark.parametrize("shape", [(1, 1), (3, 5), (65, 1), (63, 2), (1, 130)])
def test_ragged_frames(pt, po, scenes, shape):
    """Frames that are not a multiple of the 64-path tile (down to one pixel), depth 1 and the full depth,
    every pipeline: live counts and image equal the oracle's."""
    s = scenes["cornell_glass_64"]
    cam = _resized(s["camera"], *shape)
    for depth in (1, s["depth"]):
        scene = pt.Scene(s["geoms"], s["materials"], cam, depth)
        for flags, oflags in ((pt.PT_COMPACT, po.F_COMPACT), (0, 0), (pt.PT_COMPACT | pt.PT_SORT_MATERIAL, po.F_COMPACT | po.F_SORT),
                              (pt.PT_COMPACT | pt.PT_CACHE_FIRST, po.F_COMPACT)):
            ref = po.Tracer(s["geoms"], s["materials"], cam, depth, flags=oflags, trig=po.TRIG_SHARED)
            pt.pathtraceInit(scene, flags=flags, max_batch=3)
            n = shape[0] * shape[1]
            img = np.zeros((n, 3), dtype=np.float32)
            for it in (1, 2):
                img = pt.pathtrace(None, 0, it)
                st = ref.iterate(it)
                assert list(pt.get_stats().live[:depth]) == list(st.live[:depth])
            pt.trace_batch(3, 3, img)
            for it in (3, 4, 5):
                ref.iterate(it)
            assert img.tobytes() == ref.image.tobytes()
            pt.pathtraceFree()


def test_degenerate_scenes(pt, po, scenes):
    """No geometry at all (every ray misses), a lone light, and a scene whose only object is behind the camera."""
    s = scenes["cornell_64"]
    light = s["geoms"][:1].copy()
    behind = s["geoms"][6:7].copy()
    behind["translation"][0] = (0, 5, 30)
    behind["transform"][0][3][:3] = (0, 5, 30)
    inv = np.linalg.inv(behind["transform"][0].T.astype(np.float64))
    behind["inverseTransform"][0] = inv.T.astype(np.float32)
    behind["invTranspose"][0] = inv.astype(np.float32)
    for geoms in (s["geoms"][:0], light, behind):
        scene = pt.Scene(geoms, s["materials"], s["camera"], s["depth"])
        ref = po.Tracer(np.ascontiguousarray(geoms).view(po.GEOM_DT), s["materials"], s["camera"], s["depth"], trig=po.TRIG_SHARED)
        pt.pathtraceInit(scene, flags=pt.PT_COMPACT)
        for it in (1, 2):
            img = pt.pathtrace(None, 0, it)
            st = ref.iterate(it)
            assert list(pt.get_stats().live[:s["depth"]]) == list(st.live[:s["depth"]])
            assert img.tobytes() == ref.image.tobytes()
        pt.pathtraceFree()


@pytest.mark.parametrize("seed", list(range(1, 17)))
def test_random_scenes(pt, po, scenes, seed):
    """Randomised scenes: 3-24 cubes / spheres with random rotations, non-uniform scales (thin slabs to large
    enclosing shells, so rays start inside primitives too), random diffuse / mirror / glass / emissive materials;
    matrices built by the C++ host (bit-exact with the reference's utilities.cpp).  Live counts and the image of
    three iterations equal the oracle's, with and without compaction and with the material sort."""
    rng = np.random.default_rng(1000 + seed)
    s = scenes["cornell_64"]
    H = pt.host_binding.host_library()
    nm = int(rng.integers(3, 9))
    mats = np.zeros(nm, dtype=pt.MATERIAL_DT)
    for m in mats:
        m["color"] = rng.uniform(0.1, 1.0, 3)
        m["spec_color"] = rng.uniform(0.5, 1.0, 3)
        kind = rng.integers(4)
        m["hasReflective"], m["hasRefractive"] = (1.0, 0.0) if kind == 1 else ((0.0, 1.0) if kind == 2 else (0.0, 0.0))
        m["indexOfRefraction"] = rng.uniform(1.1, 2.0)
        m["emittance"] = rng.uniform(1, 6) if kind == 3 else 0.0
    mats[0]["emittance"] = 4.0                                        # at least one light
    ng = int(rng.integers(3, 25))
    geoms = np.zeros(ng, dtype=pt.GEOM_DT)
    for k, g in enumerate(geoms):
        g["type"] = rng.integers(2)
        g["materialid"] = rng.integers(nm)
        g["translation"] = rng.uniform(-4, 4, 3) + (0, 5, 0)
        g["rotation"] = rng.uniform(-180, 180, 3) * (rng.random() < 0.7)
        sc = rng.uniform(0.3, 3.0, 3)
        if k % 5 == 0:
            sc[rng.integers(3)] = 0.02                                # a thin slab
        if k == 1:
            sc = np.array([25.0, 25.0, 25.0]); g["translation"] = (0, 5, 0)   # everything happens inside this one
        g["scale"] = sc
    for k in range(ng):
        H.pth_build_geom_matrices(geoms.ctypes.data + k * pt.GEOM_DT.itemsize)
    depth = int(rng.integers(1, 9))
    scene = pt.Scene(geoms, mats, s["camera"], depth)
    for flags, oflags in ((pt.PT_COMPACT, po.F_COMPACT), (0, 0), (pt.PT_COMPACT | pt.PT_SORT_MATERIAL, po.F_COMPACT | po.F_SORT)):
        ref = po.Tracer(geoms.view(po.GEOM_DT), mats.view(po.MATERIAL_DT), s["camera"], depth, flags=oflags, trig=po.TRIG_SHARED)
        pt.pathtraceInit(scene, flags=flags)
        for it in (1, 2, 3):
            img = pt.pathtrace(None, 0, it)
            st = ref.iterate(it)
            assert list(pt.get_stats().live[:depth]) == list(st.live[:depth]), (flags, it)
        assert img.tobytes() == ref.image.tobytes(), flags
        pt.pathtraceFree()
    assert np.isfinite(ref.image).all() and ref.image.max() > 0


def test_trace_depth_reread_every_call(pt, po, scenes):
    """pathtrace() re-reads traceDepth from the scene on every call (pathtrace.cu:286): a smaller OR LARGER depth set
    after init takes effect at once (and back), like the camera."""
    s = scenes["cornell_glass_64"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT)
    total = np.zeros((scene.resolution[0] * scene.resolution[1], 3), dtype=np.float32)
    for it, depth in ((1, s["depth"]), (2, 3), (3, 1), (4, s["depth"]), (5, s["depth"] + 7), (6, 2), (7, 40)):
        scene.traceDepth = depth
        img = pt.pathtrace(None, 0, it)
        ref = po.Tracer(s["geoms"], s["materials"], s["camera"], depth, trig=po.TRIG_SHARED)
        st = ref.iterate(it)
        assert list(pt.get_stats().live[:depth]) == list(st.live[:depth])
        total += ref.image                                   # the running sum adds each iteration's contribution
        assert np.array_equal(img, total)
    with pytest.raises(pt.PtError):
        scene.traceDepth = 65                                # the control block holds 64 bounces
        pt.pathtrace(None, 0, 8)
    pt.pathtraceFree()


@pytest.mark.parametrize("ng,nm", [(150, 5), (1000, 300)])
def test_many_primitives(pt, po, scenes, ng, nm):
    """150 cubes / spheres (gather records staged in LDS) and 1000 with 300 materials (too large for LDS: matrices and
    materials are gathered through the vector cache; the reference's loop has no limit, pathtrace.cu:176): live counts
    and image equal the oracle's, fused, with the material sort (> 255 materials) and without compaction."""
    rng = np.random.default_rng(4242 + ng)
    s = scenes["cornell_64"]
    H = pt.host_binding.host_library()
    mats = np.zeros(nm, dtype=pt.MATERIAL_DT)
    mats[:len(s["materials"])] = s["materials"]
    for m in mats[len(s["materials"]):]:
        m["color"] = rng.uniform(0.2, 1.0, 3)
        m["spec_color"] = rng.uniform(0.5, 1.0, 3)
        kind = rng.integers(5)
        m["hasReflective"], m["hasRefractive"] = (1.0, 0.0) if kind == 1 else ((0.0, 1.0) if kind == 2 else (0.0, 0.0))
        m["indexOfRefraction"] = rng.uniform(1.1, 2.0)
    geoms = np.zeros(ng, dtype=pt.GEOM_DT)
    for g in geoms:
        g["type"] = rng.integers(2)
        g["materialid"] = rng.integers(nm)
        g["translation"] = rng.uniform(-4.5, 4.5, 3) + (0, 5, 0)
        g["rotation"] = rng.uniform(-180, 180, 3)
        g["scale"] = rng.uniform(0.2, 1.2, 3) * (0.5 if ng > 500 else 1.0)
    geoms[0] = s["geoms"][0]                                         # the light
    for k in range(1, ng):
        H.pth_build_geom_matrices(geoms.ctypes.data + k * pt.GEOM_DT.itemsize)
    scene = pt.Scene(geoms, mats, s["camera"], 4)
    for flags, oflags in ((pt.PT_COMPACT, po.F_COMPACT), (pt.PT_COMPACT | pt.PT_SORT_MATERIAL, po.F_COMPACT | po.F_SORT), (0, 0)):
        ref = po.Tracer(geoms.view(po.GEOM_DT), mats.view(po.MATERIAL_DT), s["camera"], 4, flags=oflags, trig=po.TRIG_SHARED)
        pt.pathtraceInit(scene, flags=flags)
        for it in (1, 2):
            img = pt.pathtrace(None, 0, it)
            st = ref.iterate(it)
            assert list(pt.get_stats().live[:4]) == list(st.live[:4]), (flags, it)
        assert img.tobytes() == ref.image.tobytes(), flags
        pt.pathtraceFree()


def test_scene_gathers_from_global_memory(pt, po, scenes, monkeypatch):
    """The Cornell scenes with the LDS staging of matrices / materials switched off (PTMI355_SCENE_LDS=0, the path
    large scenes take): same bits."""
    monkeypatch.setenv("PTMI355_SCENE_LDS", "0")
    for name in ("cornell_64", "cornell_glass_64"):
        s = scenes[name]
        scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
        ref = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"], trig=po.TRIG_SHARED)
        pt.pathtraceInit(scene, flags=pt.PT_COMPACT)
        for it in (1, 2, 3):
            img = pt.pathtrace(None, 0, it)
            st = ref.iterate(it)
            assert list(pt.get_stats().live[:s["depth"]]) == list(st.live[:s["depth"]])
        assert img.tobytes() == ref.image.tobytes()
        pt.pathtraceFree()


def test_cull_box_gates(pt, po, scenes):
    """The cull stage (csrc/pt_cull.hpp, pt_kernels.hpp stage 1) never changes an intersection: rays aimed at the
    faces / edges / corners of every primitive from 1e-7 to 1 object units off the surface, from on the surface, from
    inside, from beyond the origin bound (`wild`), with axis-parallel, zero, un-normalised, NaN / inf components --
    t, normal, material and the outside flag equal the oracle's loop over every primitive, bit for bit; for the
    Cornell box and for rotated / thin / huge / tiny / singular primitives."""
    import cull_model
    H = pt.host_binding.host_library()
    s = scenes["cornell"]
    rng = np.random.default_rng(99)
    extra = np.zeros(8, dtype=pt.GEOM_DT)
    scales = [(2.0, 0.004, 3.0), (0.5, 0.5, 0.5), (25.0, 25.0, 25.0), (1e-3, 1e-3, 1e-3), (0.0, 1.0, 1.0), (3.0, 0.3, 0.03),
              (1.0, 1.0, 1.0), (0.7, 2.0, 0.7)]
    for k, g in enumerate(extra):
        g["type"] = k % 2
        g["materialid"] = 1 + k % 4
        g["translation"] = rng.uniform(-3, 3, 3) + (0, 5, 0)
        g["rotation"] = rng.uniform(-180, 180, 3)
        g["scale"] = scales[k]
        H.pth_build_geom_matrices(extra.ctypes.data + k * pt.GEOM_DT.itemsize)
    for geoms in (s["geoms"], np.concatenate([s["geoms"], extra])):
        rays = cull_model.stress_rays(geoms, rng, per_geom=3000)
        paths = np.zeros(len(rays), dtype=pt.PATH_DT)
        paths["origin"], paths["direction"] = rays[:, :3], rays[:, 3:]
        scene = pt.Scene(geoms, s["materials"], s["camera"], s["depth"])
        pt.pathtraceInit(scene, max_batch=1 + len(rays) // (800 * 800))
        got, got_out = pt.intersect_once(paths)
        pt.pathtraceFree()
        want, want_out = po.compute_intersections(paths.view(po.PATH_DT), geoms.view(po.GEOM_DT))
        assert (bits(got["t"]) == bits(want["t"])).all()
        assert (bits(got["normal"]) == bits(want["normal"])).all()
        assert (got["materialId"] == want["materialId"]).all()
        hit = want["t"] > 0
        assert hit.sum() > 5000 and (~hit).sum() > 1000
        assert (got_out[hit] == want_out[hit]).all()


def test_singular_and_extreme_transforms(pt, po, scenes):
    """Geoms whose matrices hold inf / NaN (zero scale -> singular inverse), denormal and 1e18 scales: whatever the
    reference arithmetic makes of them (mostly misses, some NaN distances), the GPU makes the same of them."""
    s = scenes["cornell_64"]
    H = pt.host_binding.host_library()
    geoms = s["geoms"].copy()
    extra = np.zeros(6, dtype=pt.GEOM_DT)
    scales = [(0.0, 1.0, 1.0), (1e-30, 1e-30, 1e-30), (1e18, 1e18, 1e18), (1.0, 0.0, 0.0), (1e-20, 2.0, 2.0), (3e10, 1e-10, 1.0)]
    for k, g in enumerate(extra):
        g["type"] = k % 2
        g["materialid"] = 1 + k % 4
        g["translation"] = (0.5 * k - 1, 5, 0)
        g["rotation"] = (10 * k, 20, 0)
        g["scale"] = scales[k]
        H.pth_build_geom_matrices(extra.ctypes.data + k * pt.GEOM_DT.itemsize)
    geoms = np.concatenate([geoms, extra])
    assert not np.isfinite(geoms["inverseTransform"]).all()              # the singular ones really are inf / NaN
    scene = pt.Scene(geoms, s["materials"], s["camera"], s["depth"])
    rays = po.generate_rays(s["camera"], s["depth"])
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_UNFUSED)
    got, got_out = pt.intersect_once(rays.view(pt.PATH_DT))
    pt.pathtraceFree()
    want, want_out = po.compute_intersections(rays, geoms.view(po.GEOM_DT))
    assert got.tobytes() == want.tobytes()
    ref = po.Tracer(geoms.view(po.GEOM_DT), s["materials"], s["camera"], s["depth"], trig=po.TRIG_SHARED)
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT)
    for it in (1, 2):
        img = pt.pathtrace(None, 0, it)
        st = ref.iterate(it)
        assert list(pt.get_stats().live[:s["depth"]]) == list(st.live[:s["depth"]])
    assert img.tobytes() == ref.image.tobytes()
    pt.pathtraceFree()


def test_nan_camera(pt, po, scenes):
    """Scene::loadCamera leaves camera.right NaN (scene.cpp:138) until runCuda recomputes it; a host that skips the
    recompute traces NaN rays.  They take the reference's paths through the tests (every comparison false) and so
    does the GPU: same live counts, same (NaN-laden) image bits."""
    s = scenes["cornell_64"]
    cam = s["camera"].copy()
    cam["right"][0] = np.nan
    scene = pt.Scene(s["geoms"], s["materials"], cam, s["depth"])
    ref = po.Tracer(s["geoms"], s["materials"], cam, s["depth"], trig=po.TRIG_SHARED)
    for flags in (pt.PT_COMPACT, pt.PT_COMPACT | pt.PT_UNFUSED):
        pt.pathtraceInit(scene, flags=flags)
        img = pt.pathtrace(None, 0, 1)
        if flags == pt.PT_COMPACT:
            st = ref.iterate(1)
        assert list(pt.get_stats().live[:s["depth"]]) == list(st.live[:s["depth"]])
        assert img.tobytes() == ref.image.tobytes()
        pt.pathtraceFree()


def test_large_iteration_numbers(pt, po, scenes):
    """Iteration numbers past 2^22 spill into the depth bits of the seed word (pathtrace.cu:41-45); the reference
    keeps going with colliding streams and so do the oracle and the GPU, identically.  Iteration 0 too."""
    s = scenes["cornell_glass_64"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    n = scene.resolution[0] * scene.resolution[1]
    ref = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"], trig=po.TRIG_SHARED)
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT, max_batch=3)
    img = np.zeros((n, 3), dtype=np.float32)
    for it in (0, (1 << 22) - 1, 1 << 22, 5000000, (1 << 31) - 3):
        img = pt.pathtrace(None, 0, it)
        st = ref.iterate(it)
        assert list(pt.get_stats().live[:s["depth"]]) == list(st.live[:s["depth"]]), it
    pt.trace_batch((1 << 22) - 1, 3, img)                    # a batch that straddles the boundary
    for it in range((1 << 22) - 1, (1 << 22) + 2):
        ref.iterate(it)
    assert img.tobytes() == ref.image.tobytes()
    with pytest.raises(pt.PtError):
        pt.trace_batch((1 << 31) - 2, 3, img)                # would overflow int
    pt.pathtraceFree()


def test_division_fast_path_gates(pt, po, scenes, golden):
    """The rescale-free divide / sqrt sequences (csrc/pt_device.hpp: div_by_rcp, sqrt_normal_range) are
    gated per wave; rays on both sides of every gate (direction components around 2^-40, origins around
    2^54, exact zeros, denormals, huge magnitudes, NaN/inf) must still equal the oracle's IEEE results."""
    rng = np.random.default_rng(2026)
    s = scenes["cornell"]
    geoms = np.concatenate([s["geoms"], golden["geomtests"]["extra_geoms"]])
    geoms = geoms.copy()
    geoms["materialid"] = np.minimum(geoms["materialid"], len(s["materials"]) - 1)
    n = 1 << 16
    o = rng.uniform(-6, 11, (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    k = np.arange(n) % 16
    # tiny direction components straddling the 2^-40 gate, and far smaller
    m = (k == 1) | (k == 2)
    d[m, rng.integers(0, 3, m.sum())] = rng.choice([-1, 1], m.sum()) * 2.0 ** rng.uniform(-44, -36, m.sum())
    m = k == 3
    d[m, rng.integers(0, 3, m.sum())] = rng.choice([-1, 1], m.sum()) * 2.0 ** rng.uniform(-140, -60, m.sum())
    m = k == 4                                                   # exact zeros (axis-parallel)
    d[m, rng.integers(0, 3, m.sum())] = 0.0
    m = k == 5                                                   # un-normalised, enormous / minute directions
    d[m] *= (10.0 ** rng.uniform(-30, 30, m.sum()))[:, None]
    m = k == 6                                                   # origins around the 2^54 gate (object space is 1/scale larger)
    o[m] = rng.normal(size=(m.sum(), 3)) * 2.0 ** rng.uniform(40, 60, m.sum())[:, None]
    m = k == 7                                                   # origins exactly on cube faces (numerator == 0)
    o[m, 1] = 0.005
    m = k == 8
    d[m, 0] = np.nan
    m = k == 9
    o[m, 2] = np.inf
    paths = np.zeros(n, dtype=pt.PATH_DT)
    paths["origin"], paths["direction"] = o.astype(np.float32), d.astype(np.float32)
    paths["color"] = 1.0
    paths["pixelIndex"] = np.arange(n)
    paths["remainingBounces"] = 8
    scene = pt.Scene(geoms, s["materials"], s["camera"], s["depth"])
    pt.pathtraceInit(scene, max_batch=1)
    # the pool holds 640000 slots: plenty for 65536 imported rays
    got, got_out = pt.intersect_once(paths)
    want, want_out = po.compute_intersections(paths.view(po.PATH_DT), geoms.view(po.GEOM_DT))
    assert (bits(got["t"]) == bits(want["t"])).all()
    hit = want["t"] > 0
    assert hit.sum() > 10000
    assert (bits(got["normal"][hit]) == bits(want["normal"][hit])).all()
    assert (got["materialId"][hit] == want["materialId"][hit]).all()
    assert (got_out[hit] == want_out[hit]).all()
    pt.pathtraceFree()
