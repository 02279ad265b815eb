"""The plain-C oracle (oracle/ptoracle.c) against the golden vectors generated
from the reference's own code (tests/golden/make_golden.py).  Bit-exact."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_abi(golden):
    # sizes/offsets the reference's compiler assigns
    abi = golden["abi"]["abi"]
    assert list(abi) == [236, 44, 108, 172, 44, 12, 28, 40, 84, 8, 32, 68, 76, 44, 24, 36, 40, 20, 16, 0, 1]


def test_utilhash(po, golden):
    z = golden["rng"]
    got = np.array([po.lib().pto_utilhash(int(x)) for x in z["hash_in"]], dtype=np.uint32)
    assert (got == z["hash_out"]).all()
    assert po.lib().pto_utilhash(7) == 4090419040


def test_lcg_and_u01(po, golden):
    z = golden["rng"]
    for i, s in enumerate(z["seeds"]):
        assert (po.raw_sequence(int(s), 8) == z["raw"][i]).all(), hex(int(s))
        assert (bits(po.u01_sequence(int(s), 8)) == bits(z["u01"][i])).all(), hex(int(s))
    # u01 can return exactly 1.0f: state m-1 -> float(2147483645)/2^31 rounds to 1
    st = C.c_uint32(2147483646)
    # find a predecessor-free check: feed the engine a state whose successor is large
    assert po.lib().pto_lcg_seed(0) == 1 and po.lib().pto_lcg_seed(2147483647) == 1
    assert po.lib().pto_lcg_seed(2147483648) == 1 and po.lib().pto_lcg_seed(0xffffffff) == 1


def test_minstd_rand_published_known_answer(po):
    """A pin nobody here compiled: the C++ standard's check value for minstd_rand ([rand.predef]: the 10 000th
    consecutive invocation of a default-constructed object -- seed 1 -- produces 399268537), and minstd_rand0's
    predecessor constant for contrast (a = 16807 would give 1043618065).  thrust::default_random_engine is
    minstd_rand (thrust/random/linear_congruential_engine.h:274); rocThrust, through which the golden vectors of
    rng.npz were generated, is thereby tied to a published constant."""
    st = C.c_uint32(po.lib().pto_lcg_seed(1))
    assert st.value == 1
    x = 0
    for _ in range(10000):
        x = po.lib().pto_lcg_next(C.byref(st))
    assert x == 399268537
    # the same through the uniform_real_distribution path (u01 advances the engine exactly once per draw)
    st = C.c_uint32(po.lib().pto_lcg_seed(1))
    for _ in range(10000):
        u = po.lib().pto_u01(C.byref(st))
    assert st.value == 399268537
    assert np.float32(u) == np.float32(np.float32(399268536) / np.float32(2147483648.0))
    # closed form: 48271^10000 mod (2^31 - 1)
    assert pow(48271, 10000, 2147483647) == 399268537


def test_make_seeded_engine(po, golden):
    z = golden["rng"]
    for (it, idx, d), want in zip(z["key"], z["first_raw"]):
        st = C.c_uint32(po.lib().pto_make_seeded_engine(int(it), int(idx), int(d)))
        assert po.lib().pto_lcg_next(C.byref(st)) == want


def test_get_point_on_ray_and_multiply_mv(po, golden):
    z = golden["glmfuncs"]
    L = po.lib()
    for r, t, want in zip(z["gp_r"], z["gp_t"], z["gp_o"]):
        v = L.pto_get_point_on_ray(po.ray(r[:3], r[3:]), C.c_float(t))
        assert (bits(np.array([v.x, v.y, v.z], np.float32)) == bits(want)).all()
    for m, v4, want in zip(z["mv_m"], z["mv_v"], z["mv_o"]):
        mm = np.ascontiguousarray(m)
        v = L.pto_multiply_mv(P(mm), po.Vec4(*[float(x) for x in v4]))
        assert (bits(np.array([v.x, v.y, v.z], np.float32)) == bits(want)).all()


def test_reflect_and_triangle(po, golden):
    z = golden["glmfuncs"]
    L = po.lib()
    for I, N, want in zip(z["I"], z["N"], z["reflect"]):
        v = L.pto_reflect(po.vec3(I), po.vec3(N))
        assert (bits(np.array([v.x, v.y, v.z], np.float32)) == bits(want)).all()
    nhit = 0
    for o, d, v9, hit, b in zip(z["tri_o"], z["tri_d"], z["tri_v"], z["tri_hit"], z["tri_b"]):
        out = po.Vec3(-7, -7, -7)
        got = L.pto_ray_triangle(po.vec3(o), po.vec3(d), po.vec3(v9[0:3]), po.vec3(v9[3:6]),
                                 po.vec3(v9[6:9]), C.byref(out))
        assert got == hit
        assert (bits(np.array([out.x, out.y, out.z], np.float32)) == bits(b)).all()
        nhit += hit
    assert nhit > 100


def test_box_and_sphere(po, golden, scenes):
    z = golden["geomtests"]
    cornell = scenes["cornell"]["geoms"]
    extra = z["extra_geoms"]
    hits = 0
    for i, gi in enumerate(z["geom_index"]):
        geom = cornell[gi] if gi < 100 else extra[gi - 100]
        rays, want = z["rays_%d" % i], z["out_%d" % i]
        got = po.geom_test(np.array([geom]), rays, int(geom["type"]))
        # compare bit patterns so NaN payloads and signed zeros count
        assert (bits(got) == bits(want)).all(), "geom %d" % gi
        hits += int((want[:, 0] > 0).sum())
    assert hits > 2000


def test_hemisphere(po, golden):
    z = golden["hemisphere"]
    got_l = po.hemisphere(z["normals"], z["seeds"], po.TRIG_LIBM)
    got_s = po.hemisphere(z["normals"], z["seeds"], po.TRIG_SHARED)
    assert (bits(got_l) == bits(z["libm"])).all()
    assert (bits(got_s) == bits(z["shared"])).all()
    # the shared trig (within 1 ulp of the correctly rounded sin / cos, and equal to it for 98.6 % of the arguments) and
    # this container's libm give the same direction components almost always
    assert (bits(z["libm"]) != bits(z["shared"])).mean() < 0.03


def test_shared_sincos_accuracy(po):
    """The shared sin / cos (binary32, explicit fmas: DESIGN.md section 4) against the correctly rounded values: never more
    than 1 ulp away -- what the libms the reference can bind to deliver (CUDA's sinf / cosf: 1 ulp) -- and correctly
    rounded for at least 98 % of a uniform grid over [0, 2 pi].  (Every float of [0, 2 pi] is covered by the tool
    run quoted in oracle/ptoracle.c and, device against oracle, by tests/test_gpu_pins.py.)"""
    def ulps(a, exact):
        cr = np.float32(exact)
        ia, ib = (int(np.float32(v).view(np.int32)) for v in (a, cr))
        ia = ia if ia >= 0 else -(ia & 0x7fffffff)
        ib = ib if ib >= 0 else -(ib & 0x7fffffff)
        return abs(ia - ib)
    xs = np.concatenate([np.linspace(0, 2 * np.pi, 20001), np.arange(9) * (np.pi / 4), [1e-30, 5.5e-4, 6.2831855]]).astype(np.float32)
    off = worst = 0
    for x in xs:
        s, c = po.sincos(float(x))
        ds, dc = ulps(s, np.sin(np.float64(x))), ulps(c, np.cos(np.float64(x)))
        worst = max(worst, ds, dc)
        off += (ds != 0) + (dc != 0)
    assert worst <= 1
    assert off < 0.02 * 2 * len(xs), off
    # known values: exact at 0, the quadrant logic at multiples of pi/2 (as float arguments)
    assert po.sincos(0.0) == (0.0, 1.0)
    s, c = po.sincos(float(np.float32(np.pi / 2)))
    assert s == 1.0 and abs(c - np.float32(np.cos(np.float64(np.float32(np.pi / 2))))) < 1e-14
    s, c = po.sincos(float(np.float32(np.pi)))
    assert c == -1.0 and np.float32(s) == np.float32(np.sin(np.float64(np.float32(np.pi))))


def test_sincos_sums_are_the_probe_checksums(po):
    """pto_sincos_sums (what the GPU suite holds pt_probe_sincos against) is the plain sum over pto_sincos."""
    first, n = 0x40000000, 257                                       # 2.0 ...
    xs = (np.arange(n, dtype=np.uint32) + np.uint32(first)).view(np.float32)
    ss = sc = 0
    for k, x in enumerate(xs):
        s, c = po.sincos(float(x))
        ss = (ss + int(np.float32(s).view(np.uint32)) * (2 * k + 1)) % (1 << 64)
        sc = (sc + int(np.float32(c).view(np.uint32)) * (2 * k + 1)) % (1 << 64)
    assert po.sincos_sums(first, n) == (ss, sc)


def test_raygen(po, golden, scenes):
    z = golden["raygen"]
    p64 = po.generate_rays(scenes["cornell_64"]["camera"], 8)
    assert p64.tobytes() == z["paths64"].tobytes()
    p800 = po.generate_rays(scenes["cornell"]["camera"], 8)
    assert hashlib.md5(p800.tobytes()).hexdigest() == str(z["md5_800"])
    assert p800.reshape(800, 800)[::13, ::13].tobytes() == z["sub800"].tobytes()


def test_fake_shader_as_is(po, golden, scenes):
    """The only end-to-end behaviour the reference itself computes (one bounce + fake shader)."""
    z = golden["fakeshade"]
    for name, iters, img_key in (("cornell_64", 3, "img64"), ("cornell", 2, None)):
        s = scenes[name]
        tr = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"], flags=po.F_FAKESHADE)
        first_isects = None
        for it in range(1, iters + 1):
            tr.iterate(it)
            if first_isects is None:
                first_isects = tr.isects.copy()
        W, H = s["camera"][0]["resolution"]
        pbo = np.zeros((tr.n, 4), dtype=np.uint8)
        po.lib().pto_send_image_to_pbo(P(pbo), int(W), int(H), iters, P(tr.image))
        if img_key:
            assert tr.image.tobytes() == z["img64"].tobytes()
            assert pbo.tobytes() == z["pbo64"].tobytes()
            assert first_isects.tobytes() == z["isect64"].tobytes()
        else:
            assert hashlib.md5(tr.image.tobytes()).hexdigest() == str(z["md5_img800"])
            assert hashlib.md5(pbo.tobytes()).hexdigest() == str(z["md5_pbo800"])
            assert hashlib.md5(first_isects.tobytes()).hexdigest() == str(z["md5_isect800"])


@pytest.mark.parametrize("trig", ["shared", "libm"])
@pytest.mark.parametrize("scene", ["cornell_64", "cornell_glass_64", "cornell_diffuse_64"])
def test_completion_small(po, golden, scenes, scene, trig):
    """SURVEY 8.0 completion spec: oracle == the same spec driven through the reference's headers."""
    z = golden["completion"]
    s = scenes[scene]
    tr = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"], flags=po.F_COMPACT,
                   trig=po.TRIG_SHARED if trig == "shared" else po.TRIG_LIBM)
    pre = "%s__%s__" % (trig, scene)
    for it in range(1, 5):
        st = tr.iterate(it)
        D = s["depth"]
        live = np.array(st.live[:D])
        assert (live == z[pre + "live"][it - 1]).all()
        assert st.rays == z[pre + "rays"][it - 1]
        h = np.array(st.seq_hash[:st.bounces], dtype=np.uint64)
        assert (h == z[pre + "seq_hash"][it - 1][:st.bounces]).all()
        assert tr.image.tobytes() == z[pre + "images"][it - 1].tobytes()


def test_completion_c2_full(po, golden, scenes):
    """Config C2 (800x800 depth 8): live counts, compaction-order hashes, image digest."""
    z = golden["completion"]
    s = scenes["cornell"]
    tr = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"], flags=po.F_COMPACT, trig=po.TRIG_SHARED)
    for it in (1, 2):
        st = tr.iterate(it)
        assert (np.array(st.live[:8]) == z["shared__cornell__live"][it - 1]).all()
        assert st.rays == z["shared__cornell__rays"][it - 1]
        assert (np.array(st.seq_hash[:8], dtype=np.uint64) == z["shared__cornell__seq_hash"][it - 1]).all()
        assert hashlib.md5(tr.image.tobytes()).hexdigest() == str(z["shared__cornell__img_md5"][it - 1])


def test_compaction_invariance(po, scenes):
    """Image is invariant to compaction on/off and to material sorting (RNG keyed by pixel)."""
    s = scenes["cornell_glass_64"]
    imgs = []
    for flags in (po.F_COMPACT, 0, po.F_COMPACT | po.F_SORT):
        tr = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"], flags=flags)
        for it in (1, 2):
            tr.iterate(it)
        imgs.append(tr.image.copy())
    assert imgs[0].tobytes() == imgs[1].tobytes() == imgs[2].tobytes()


def test_mt_matches_st(po, scenes):
    s = scenes["cornell_64"]
    a = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"])
    b = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"])
    sa = a.iterate(1)
    sb = b.iterate(1, threads=3)
    assert a.image.tobytes() == b.image.tobytes()
    assert list(sa.live[:8]) == list(sb.live[:8]) and list(sa.seq_hash[:8]) == list(sb.seq_hash[:8])


def test_png_statistic(po, golden, scenes):
    """Coarse check against the reference's only rendered artefact:
    x-flipped, clamped, sphere masked (the PNG's ball is matte), 16x16 pooled."""
    s = scenes["cornell"]
    cam = s["camera"].copy()
    tr = po.Tracer(s["geoms"], s["materials"], cam, s["depth"], trig=po.TRIG_SHARED)
    iters = 6
    for it in range(1, iters + 1):
        tr.iterate(it, threads=8)
    img = (tr.image / iters).reshape(800, 800, 3)[:, ::-1, :]          # saveImage x-flip (main.cpp:87)
    img = np.floor(np.clip(img, 0, 1) * 255.0) / 255.0                 # savePNG (image.cpp:22-39)
    pooled = img.reshape(50, 16, 50, 16, 3).mean(axis=(1, 3))
    want = golden["png_stat"]["pooled"]
    mask = np.ones((50, 50), dtype=bool)
    mask[22:40, 12:32] = False                                         # the ball and its reflection/shadow
    num = np.sqrt(((pooled - want)[mask] ** 2).sum())
    den = np.sqrt((want[mask] ** 2).sum())
    assert num / den < 0.12, num / den                                 # 6 spp noise floor ~0.08; 0.05 at >=256 spp


def test_iteration_parallel_driver_equals_sequential(po, scenes):
    """bench.py's CPU baseline runs one whole iteration per thread and adds the images in iteration order: the
    same bits as sequential iterations."""
    s = scenes["cornell_glass_64"]
    a = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"], trig=po.TRIG_SHARED)
    b = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"], trig=po.TRIG_SHARED)
    want = sum(a.iterate(it).rays for it in range(3, 14))
    got = b.iterate_parallel(3, 11, 4)
    assert got == want and a.image.tobytes() == b.image.tobytes()


def test_c1_as_stated(po, scenes, golden):
    """BASELINE configs[0] exactly as stated -- scenes/cornell_diffuse.txt: 400 x 400, 1 spp, depth 4, diffuse only,
    the __host__ intersection / shade path in a SINGLE-THREAD loop: the oracle's iteration 1 against the image, live
    counts and ray count the reference's own headers produce (tests/golden/c1.npz, written by make_golden.py from
    oracle/_ref/libptref_b_shared.so)."""
    import hashlib
    s = scenes["cornell_diffuse"]
    assert tuple(s["camera"][0]["resolution"]) == (400, 400) and s["depth"] == 4
    used = s["materials"][np.unique(s["geoms"]["materialid"])]          # the mirror of cornell.txt is still listed, unused
    assert (used["hasReflective"] == 0).all() and (used["hasRefractive"] == 0).all()
    ref = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"])
    st = ref.iterate(1)                                   # threads = 0: the plain single-thread loop
    z = golden["c1"]
    assert list(st.live[:4]) == list(z["live"][0]) and st.rays == int(z["rays"][0])
    assert hashlib.md5(ref.image.tobytes()).hexdigest() == str(z["img_md5"])
    assert ref.image[::53].tobytes() == z["img_sub"].tobytes()


def test_c4_frame_golden_strip(po, golden, scenes):
    """tests/golden/c4_frame.npz (the oracle's whole-frame C4 iteration, made by tests/golden/make_c4_golden.py) is what
    this checkout's oracle still produces: one 16-row strip of it re-traced here (a few seconds), md5 and sampled pixels."""
    import hashlib
    import __graft_entry__ as ge
    pt = ge.load_package()                       # host-side mesh generator only
    z = golden["c4_frame"]
    s = scenes["cornell"]
    tris = pt.meshes.uv_sphere()
    geoms, tris, meshes = pt.meshes.add_mesh(s["geoms"], tris, material_id=1)
    ref = po.Tracer(geoms, s["materials"], s["camera"], s["depth"], tris=tris.view(po.TRI_DT), meshes=meshes.view(po.MESH_DT))
    r, strip, W = 3, int(z["strip_rows"]), 800
    ref.iterate_rows(1, r * strip, (r + 1) * strip, threads=min(8, os.cpu_count() or 1))
    rows = slice(r * strip * W, (r + 1) * strip * W)
    assert hashlib.md5(ref.image[rows].tobytes()).hexdigest() == str(z["strip_md5"][r])
    idx = z["sample_index"]
    inside = (idx >= rows.start) & (idx < rows.stop)
    assert inside.sum() > 20 and ref.image[idx[inside]].tobytes() == z["sample_value"][inside].tobytes()
