"""GPU parity, frames and cameras: tiled frames (the multi-GPU sharding's device side), camera tile masks and bounce-0 candidate
masks, stochastic antialiasing and the thin lens, the first-bounce cache, C5's 4K frame -- against the oracle, bit for bit."""
import hashlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import __graft_entry__ as ge  # noqa: E402,F401
from gpu_common import pt, launch_plan, bits, rel_l2, assert_paths_equal, _resized, _after  # noqa: E402,F401

pytestmark = pytest.mark.gpu


def test_tiles_equal_whole_frame(pt, scenes):
    """Interleaved row-strip tiles (multi-GPU sharding) reproduce the 1-tile image exactly:
    the RNG is keyed by the global pixelIndex."""
    s = scenes["cornell_64"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    n = scene.resolution[0] * scene.resolution[1]
    pt.pathtraceInit(scene)
    for it in (1, 2):
        whole = pt.pathtrace(None, 0, it).copy()
    pt.pathtraceFree()
    for tiles, strip in ((2, 8), (3, 5), (8, 4)):
        acc = np.zeros((n, 3), dtype=np.float32)
        for k in range(tiles):
            pt.pathtraceInit(scene, tile=(k, tiles, strip))
            for it in (1, 2):
                img = pt.pathtrace(None, 0, it)
            # tiles own disjoint pixels: summing zero-padded frames == RCCL reduce(SUM), exact
            assert ((acc != 0) & (img != 0)).sum() == 0
            acc += img
            pt.pathtraceFree()
        assert acc.tobytes() == whole.tobytes(), (tiles, strip)


def test_tiles_equal_whole_frame_with_mesh_jitter_and_lens(pt, scenes):
    """The same for everything that is keyed by pixel or path index: a mesh through the hierarchy (mesh pre-pass
    masks and records), pixel jitter and the lens (random engine keyed by the GLOBAL pixel index), batches."""
    s = scenes["cornell_glass_64"]
    tris = pt.meshes.uv_sphere(center=(1.5, 3.0, 1.0), radius=1.5, n_lat=16, n_lon=32)
    geoms, tris, meshes = pt.meshes.add_mesh(s["geoms"], tris, material_id=2)
    scene = pt.Scene(geoms, s["materials"], s["camera"], s["depth"], triangles=tris, meshes=meshes)
    n = scene.resolution[0] * scene.resolution[1]
    kw = dict(flags=pt.PT_COMPACT | pt.PT_MESH_BVH | pt.PT_AA_JITTER, lens=(0.2, 9.0), max_batch=3)
    whole = np.zeros((n, 3), dtype=np.float32)
    pt.pathtraceInit(scene, **kw)
    pt.trace_batch(1, 3, whole)
    pt.pathtraceFree()
    assert np.isfinite(whole).all() and whole.max() > 0
    for tiles, strip in ((2, 8), (5, 3)):
        acc = np.zeros((n, 3), dtype=np.float32)
        for k in range(tiles):
            img = np.zeros((n, 3), dtype=np.float32)
            pt.pathtraceInit(scene, tile=(k, tiles, strip), **kw)
            pt.trace_batch(1, 3, img)
            pt.pathtraceFree()
            assert ((acc != 0) & (img != 0)).sum() == 0
            acc += img
        assert acc.tobytes() == whole.tobytes(), (tiles, strip)


def test_tile_order_matches_host_sharding(pt, scenes):
    """csrc local_to_pixel == sharding.tile_pixel_indices (what bench.py / RCCL plumbing assume)."""
    s = scenes["cornell_64"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    for rank, world, strip in ((1, 3, 5), (7, 8, 4), (0, 2, 8)):
        pt.pathtraceInit(scene, tile=(rank, world, strip))
        pt.trace_begin(1, 1)
        want = pt.sharding.tile_pixel_indices(rank, world, strip, 64, 64)
        paths, live = pt.export_paths(len(want))
        assert live == len(want)
        assert (paths["pixelIndex"] == want).all()
        pt.pathtraceFree()


@pytest.mark.parametrize("aa", [False, True])
def test_camera_tile_mask(pt, po, scenes, aa):
    """Bounce 0 of the mesh pre-pass skips the 64-pixel tiles that cannot see a mesh (pt_h_scene.hpp: update_cam_mask).
    Same frames as the oracle with the mesh in full view, half off-screen, seen from very close, from INSIDE its box
    (a corner behind the eye: no mask), and as the camera moves between batches (pt_set_camera rebuilds the mask)."""
    s = scenes["cornell_64"]
    tris = pt.meshes.uv_sphere(center=(1.2, 4.0, 0.5), radius=1.4, n_lat=16, n_lon=32)
    geoms, tris, meshes = pt.meshes.add_mesh(s["geoms"][:6], tris, material_id=2)
    og, ot, om = geoms.view(po.GEOM_DT), tris.view(po.TRI_DT), meshes.view(po.MESH_DT)
    gflags = pt.PT_COMPACT | pt.PT_MESH_BVH | (pt.PT_AA_JITTER if aa else 0)
    oflags = po.F_COMPACT | (po.F_AA if aa else 0)
    moves = [(0.0, 0.0, 0.0), (3.2, 0.0, 0.0), (1.0, -1.0, -6.5), (1.2, -1.0, -9.8), (0.0, 0.0, 0.0), (-4.0, 2.0, -3.0)]
    cams = []
    for dx, dy, dz in moves:
        cam = np.array(s["camera"], copy=True).reshape(1)
        cam["position"][0] += np.float32([dx, dy, dz])
        cams.append(cam)
    n = int(cams[0]["resolution"][0][0]) * int(cams[0]["resolution"][0][1])
    # (a) a renderer initialised at each position (mask built by pt_init)
    for k, cam in enumerate(cams):
        scene = pt.Scene(geoms, s["materials"], cam, s["depth"], triangles=tris, meshes=meshes)
        ref = po.Tracer(og, s["materials"], cam, s["depth"], flags=oflags, trig=po.TRIG_SHARED, tris=ot, meshes=om)
        pt.pathtraceInit(scene, flags=gflags, max_batch=2)
        img = np.zeros((n, 3), dtype=np.float32)
        pt.trace_batch(1 + 2 * k, 2, img)
        pt.pathtraceFree()
        ref.iterate(1 + 2 * k); ref.iterate(2 + 2 * k)
        assert img.tobytes() == ref.image.tobytes(), k
        assert (ref.image.sum(axis=1) > 0).any()
    # (b) ONE renderer whose camera moves between batches (mask rebuilt by pt_set_camera); the running sum carries over
    scene = pt.Scene(geoms, s["materials"], cams[0], s["depth"], triangles=tris, meshes=meshes)
    pt.pathtraceInit(scene, flags=gflags, max_batch=2)
    img = np.zeros((n, 3), dtype=np.float32)
    total = np.zeros((n, 3), dtype=np.float32)
    for k, cam in enumerate(cams):
        pt.set_camera(cam, s["depth"])
        pt.trace_batch(1 + 2 * k, 2, img)
        ref = po.Tracer(og, s["materials"], cam, s["depth"], flags=oflags, trig=po.TRIG_SHARED, tris=ot, meshes=om)
        ref.image[:] = total
        ref.iterate(1 + 2 * k); ref.iterate(2 + 2 * k)
        total = ref.image.copy()
        assert img.tobytes() == total.tobytes(), k
    pt.pathtraceFree()


def test_bounce0_candidate_masks(pt, po, scenes, monkeypatch):
    """Bounce 0 of a pinhole camera skips, per 64-pixel camera tile, the cull test of the primitives no ray of the
    tile is a candidate of (k_cull0_mask, rebuilt by pt_set_camera).  Frames equal the oracle's as the camera moves
    between batches -- sideways, far outside the scene (the cull boxes are remade for the new reach), looking
    away from it -- and equal the frames of a renderer with the masks switched off; a tile of a sharded frame and the
    stepping interface (which loads rays written by k_raygen: no masks) are covered by the other tests."""
    s = scenes["cornell_64"]
    moves = [(0.0, 0.0, 0.0), (2.5, 0.5, 0.0), (0.0, 0.0, -40.0), (30.0, 10.0, 5.0), (0.0, 0.0, 0.0), (-3.0, 2.0, -2.0)]
    cams = []
    for k, (dx, dy, dz) in enumerate(moves):
        cam = np.array(s["camera"], copy=True).reshape(1)
        cam["position"][0] += np.float32([dx, dy, dz])
        if k == 5:                                           # look away from the box: most tiles see nothing at all
            cam["view"][0] = np.float32([0.6, 0.0, 0.8])
            cam["right"][0] = np.float32([-0.8, 0.0, 0.6])
        cams.append(cam)
    n = int(cams[0]["resolution"][0][0]) * int(cams[0]["resolution"][0][1])
    frames = {}
    for masks in ("1", "0"):
        monkeypatch.setenv("PTMI355_CULL0", masks)
        scene = pt.Scene(s["geoms"], s["materials"], cams[0], s["depth"])
        pt.pathtraceInit(scene, flags=pt.PT_COMPACT, max_batch=3)
        img = np.zeros((n, 3), dtype=np.float32)
        total = np.zeros((n, 3), dtype=np.float32)
        for k, cam in enumerate(cams):
            pt.set_camera(cam, s["depth"])
            pt.trace_batch(1 + 3 * k, 3, img)
            if masks == "1":
                ref = po.Tracer(s["geoms"], s["materials"], cam, s["depth"], flags=po.F_COMPACT, trig=po.TRIG_SHARED)
                ref.image[:] = total
                for it in range(3):
                    ref.iterate(1 + 3 * k + it)
                total = ref.image.copy()
                assert img.tobytes() == total.tobytes(), k
            frames[(masks, k)] = img.copy()
        pt.pathtraceFree()
    for k in range(len(cams)):
        assert frames[("1", k)].tobytes() == frames[("0", k)].tobytes(), k
    monkeypatch.delenv("PTMI355_CULL0")


@pytest.mark.parametrize("mode", ["aa", "lens", "aa+lens"])
def test_camera_jitter_and_lens(pt, po, scenes, mode):
    """Stochastic antialiasing and the thin lens (completion spec; pathtrace.cu:134 TODO, INSTRUCTION.md:110-113):
    the camera rays, every iteration's image and the batched path agree with the oracle bit for bit."""
    s = scenes["cornell_glass_64"]
    aa = "aa" in mode
    lens = (0.35, 9.0) if "lens" in mode else (0.0, 0.0)
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    n = scene.resolution[0] * scene.resolution[1]
    gflags = pt.PT_COMPACT | (pt.PT_AA_JITTER if aa else 0)
    oflags = po.F_COMPACT | (po.F_AA if aa else 0)
    # the rays themselves (stepping interface -> k_raygen)
    pt.pathtraceInit(scene, flags=gflags, lens=lens)
    for it in (1, 5):
        pt.trace_begin(it, 1)
        paths, live = pt.export_paths(n)
        want = po.generate_rays_ex(s["camera"], s["depth"], it, aa=aa, lens=lens)
        assert live == n and paths.tobytes() == want.tobytes()
        pt.trace_end()
    pt.pathtraceFree()
    # whole iterations, one at a time (rays generated inside bounce 0), then as batches
    ref = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"], flags=oflags, trig=po.TRIG_SHARED, lens=lens)
    pt.pathtraceInit(scene, flags=gflags, lens=lens)
    for it in range(1, 7):
        img = pt.pathtrace(None, 0, it)
        st = ref.iterate(it)
        assert list(pt.get_stats().live[:s["depth"]]) == list(st.live[:s["depth"]])
        assert img.tobytes() == ref.image.tobytes()
    pt.pathtraceFree()
    pt.pathtraceInit(scene, flags=gflags, lens=lens, max_batch=4)
    img = np.zeros((n, 3), dtype=np.float32)
    pt.trace_batch(1, 4, img)
    pt.trace_batch(5, 2, img)
    assert img.tobytes() == ref.image.tobytes()
    pt.pathtraceFree()
    # sorted batches generate bounce 0 in k_intersect<GEN> / k_shade_sorted_w<GEN> (jitter and lens included)
    ref = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"], flags=oflags | po.F_SORT, trig=po.TRIG_SHARED,
                    lens=lens)
    pt.pathtraceInit(scene, flags=gflags | pt.PT_SORT_MATERIAL, lens=lens)
    for it in (1, 2):
        img = pt.pathtrace(None, 0, it)
        ref.iterate(it)
        assert img.tobytes() == ref.image.tobytes()
    pt.pathtraceFree()


def test_camera_extensions_exclude_the_first_bounce_cache(pt, scenes):
    s = scenes["cornell_64"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    with pytest.raises(pt.PtError, match="PT_CACHE_FIRST"):
        pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_CACHE_FIRST | pt.PT_AA_JITTER)
    with pytest.raises(pt.PtError, match="PT_CACHE_FIRST"):
        pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_CACHE_FIRST, lens=(0.1, 5.0))
    with pytest.raises(pt.PtError, match="focal_distance"):
        pt.pathtraceInit(scene, flags=pt.PT_COMPACT, lens=(0.1, 0.0))
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT)
    a = pt.pathtrace(None, 0, 1).copy()
    pt.set_lens(0.3, 9.0)                        # takes effect from the next iteration on
    pt.clear_image()
    b = pt.pathtrace(None, 0, 1).copy()
    pt.set_lens(0.0, 0.0)
    pt.clear_image()
    c = pt.pathtrace(None, 0, 1).copy()
    assert (a != b).any() and a.tobytes() == c.tobytes()
    pt.pathtraceFree()


def test_first_bounce_cache_follows_camera(pt, po, scenes):
    """PT_CACHE_FIRST (INSTRUCTION.md:87-89): batches reuse the cached bounce-0 intersections; a camera
    change through pathtrace()'s per-call re-read invalidates them."""
    s = scenes["cornell_64"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_CACHE_FIRST, max_batch=3)
    ref = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"])
    img = np.zeros((64 * 64, 3), dtype=np.float32)
    pt.trace_batch(1, 3, img)
    pt.trace_batch(4, 2, img)
    for it in range(1, 6):
        ref.iterate(it)
    assert img.tobytes() == ref.image.tobytes()
    cam2 = s["camera"].copy()
    cam2["position"][0][0] += 0.75                       # move the eye: same resolution, new rays
    scene.camera = cam2
    got = pt.pathtrace(None, 0, 6).copy()
    ref2 = po.Tracer(s["geoms"], s["materials"], cam2, s["depth"])
    ref2.image[:] = ref.image
    ref2.iterate(6)
    assert got.tobytes() == ref2.image.tobytes()
    pt.pathtraceFree()


def test_c5_tile_of_4k_frame(pt, po, scenes):
    """Config C5's sharding at full size: rank 3 of 8 at 3840x2160 (interleaved 8-row strips); its pixels
    must equal the same pixels of the oracle's whole-frame iteration (global pixelIndex keys the RNG)."""
    import os
    s = scenes["cornell_4k"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    W, H = scene.resolution
    assert (W, H) == (3840, 2160)
    pt.pathtraceInit(scene, tile=(3, 8, 8))
    img = pt.pathtrace(None, 0, 1).copy()
    gs = pt.get_stats()
    pt.pathtraceFree()
    ref = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"])
    ref.iterate(1, threads=min(64, os.cpu_count() or 8))
    own = pt.sharding.tile_pixel_indices(3, 8, 8, W, H)
    assert len(own) == gs.live[0] and abs(len(own) - W * H // 8) <= 8 * W
    assert img[own].tobytes() == ref.image[own].tobytes()
    other = np.ones(W * H, dtype=bool)
    other[own] = False
    assert not img[other].any()                           # zero-padded elsewhere: reduce(SUM) is exact


def c5_pooled_statistic(img_sum, samples, golden):
    """BASELINE C5 against the reference's only rendered artefact: the 4K frame has the 800x800 scene's FOVY, so its
    central 2160x2160 square IS that view (2.7 x finer); saveImage's x-flip, clamp and 8-bit quantisation, then the 50x50
    pooled means of tests/golden/png_stat.npz (bins of 43.2 pixels, edges rounded), ball / reflection / shadow masked."""
    W, H = 3840, 2160
    img = (np.asarray(img_sum, dtype=np.float32).reshape(H, W, 3) / np.float32(samples))[:, ::-1, :]
    img = np.floor(np.clip(img, 0, 1) * 255.0) / 255.0
    sq = img[:, (W - H) // 2:(W + H) // 2, :]
    edges = np.round(np.arange(51) * (H / 50.0)).astype(int)
    rows = np.add.reduceat(sq, edges[:-1], axis=0)
    cells = np.add.reduceat(rows, edges[:-1], axis=1)
    area = np.diff(edges)[:, None] * np.diff(edges)[None, :]
    pooled = cells / area[:, :, None]
    want = golden["png_stat"]["pooled"]
    mask = np.ones((50, 50), dtype=bool)
    mask[22:40, 12:32] = False
    return float(np.sqrt(((pooled - want)[mask] ** 2).sum()) / np.sqrt((want[mask] ** 2).sum()))


def test_c5_as_stated_5000spp_through_eight_contexts(pt, golden, tmp_path, launch_plan):
    """BASELINE config C5 AS STATED, on the one GPU of this box: scenes/cornell_4k.txt (3840x2160, depth 8) for its 5000
    iterations through the headless host with the frame tiled over EIGHT contexts (`ptbench --devices 0,0,0,0,0,0,0,0`:
    interleaved 8-row strips, a tile exchange after every batch, as on eight GPUs) -- 1.6 * 10^11 rays.  The image meets
    the reference's 5000-sample PNG (pooled statistic, relative L2 <= 0.05), and a second render of the same frame --
    ONE context, other batch size, interrupted after 2500 iterations and resumed from its saved sum in a new process
    (`--save-sum` / `--resume`) -- gives the same running sum bit for bit."""
    import subprocess
    if launch_plan != "small batches in one launch":
        pytest.skip("one 5000-spp 4K render per suite: the launch plans meet at this batch size")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = pt.build_ptbench()
    scene = os.path.join(root, "scenes", "cornell_4k.txt")
    run = lambda *a: subprocess.run([exe, scene] + list(a), capture_output=True, text=True, timeout=900)
    p = run("--devices", "0,0,0,0,0,0,0,0", "--batch", "4", "--out", str(tmp_path / "m"), "--save-sum")
    assert p.returncode == 0 and "5000 iterations" in p.stdout and "on 8 device(s)" in p.stdout, p.stdout + p.stderr
    full = pt.load_pfm(str(tmp_path / "m.5000samp.sum.pfm"), 3840, 2160)
    stat = c5_pooled_statistic(full, 5000, golden)
    assert stat <= 0.05, stat
    assert np.isfinite(full).all() and full.min() >= 0.0
    p = run("--iters", "2500", "--batch", "5", "--out", str(tmp_path / "s"), "--save-sum")
    assert p.returncode == 0, p.stdout + p.stderr
    p = run("--batch", "5", "--out", str(tmp_path / "s"), "--resume", str(tmp_path / "s.2500samp.sum.pfm"), "--save-sum")
    assert p.returncode == 0 and "resumed" in p.stdout, p.stdout + p.stderr
    again = pt.load_pfm(str(tmp_path / "s.5000samp.sum.pfm"), 3840, 2160)
    assert again.tobytes() == full.tobytes()
