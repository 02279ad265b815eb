#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ from the REFERENCE's
own code (oracle/_ref, built by `make -C oracle ref` from /root/reference).

Runs only in the build container (needs /root/reference).  The outputs are
data -- inputs and the reference's outputs -- never reference source.

    python tests/golden/make_golden.py

Each .npz holds the inputs and the reference's answers; tests/test_oracle_golden.py
replays the inputs through the plain-C oracle and demands bit equality.
"""
import ctypes as C
import hashlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SCENES = os.path.join(ROOT, "scenes")


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def load_scene(A, path):
    geoms = np.zeros(64, dtype=po.GEOM_DT)
    mats = np.zeros(64, dtype=po.MATERIAL_DT)
    cam = np.zeros(1, dtype=po.CAMERA_DT)
    ng, nm, iters, depth = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    name = C.create_string_buffer(64)
    A.ref_load_scene(path.encode(), P(geoms), 64, C.byref(ng), P(mats), 64, C.byref(nm), P(cam),
                     C.byref(iters), C.byref(depth), name)
    cam_loaded = cam.copy()
    A.ref_camera_orbit(P(cam))
    return dict(geoms=geoms[:ng.value].copy(), materials=mats[:nm.value].copy(),
                camera_loaded=cam_loaded, camera=cam.copy(), iterations=iters.value,
                depth=depth.value, name=name.value.decode())


def scene_variant(src, res=None, depth=None):
    """Write a temp copy of a repo scene with RES/DEPTH replaced; return its path."""
    txt = open(os.path.join(SCENES, src)).read().split("\n")
    for i, line in enumerate(txt):
        if res and line.startswith("RES"):
            txt[i] = "RES         %d %d" % res
        if depth and line.startswith("DEPTH"):
            txt[i] = "DEPTH       %d" % depth
    f = tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False)
    f.write("\n".join(txt))
    f.close()
    return f.name


def adversarial_rays(rng, geom, n):
    """Random + adversarial rays around one geom (world space): from inside,
    grazing, axis-parallel (zero direction components -> +-inf slabs), far."""
    T = geom["transform"].astype(np.float64).T          # row-major 4x4
    rays = np.zeros((n, 6), dtype=np.float32)
    for i in range(n):
        k = i % 8
        if k == 0:      # random origin in the scene, random direction
            o = rng.uniform(-6, 11, 3); d = rng.normal(size=3)
        elif k == 1:    # origin inside the object
            o = (T @ np.append(rng.uniform(-.3, .3, 3), 1))[:3]; d = rng.normal(size=3)
        elif k == 2:    # aimed at the object's surface region
            o = rng.uniform(-6, 11, 3); tgt = (T @ np.append(rng.uniform(-.5, .5, 3), 1))[:3]; d = tgt - o
        elif k == 3:    # axis-parallel direction (exact zeros)
            o = rng.uniform(-6, 11, 3); d = np.zeros(3); d[rng.integers(3)] = rng.choice([-1.0, 1.0])
        elif k == 4:    # two zero components, aimed through the object
            ax = rng.integers(3); tgt = (T @ np.append(rng.uniform(-.4, .4, 3), 1))[:3]
            o = tgt.copy(); o[ax] += rng.choice([-7.0, 7.0]); d = np.zeros(3); d[ax] = -np.sign(o[ax] - tgt[ax])
        elif k == 5:    # grazing: aimed at an edge / silhouette
            e = rng.choice([-.5, .5], 3); e[rng.integers(3)] = rng.uniform(-.5, .5)
            tgt = (T @ np.append(e, 1))[:3]; o = rng.uniform(-6, 11, 3); d = tgt - o
        elif k == 6:    # un-normalised / tiny / huge direction magnitudes
            o = rng.uniform(-6, 11, 3); d = rng.normal(size=3) * 10.0 ** rng.uniform(-6, 6)
        else:           # pointing away
            c = T[:3, 3]; o = rng.uniform(-6, 11, 3); d = o - c
        if k not in (3, 4, 6):
            nrm = np.linalg.norm(d)
            d = d / nrm if nrm > 0 else np.array([0, 0, 1.0])
        rays[i, :3] = o; rays[i, 3:] = d
    return rays


def main():
    po.build(ref=True)
    A, BL, BS = po.ref("a"), po.ref("b_libm"), po.ref("b_shared")
    rng = np.random.default_rng(565)

    # ---- ABI -------------------------------------------------------------
    abi = np.zeros(64, dtype=np.int32)
    n = A.ref_abi(P(abi))
    abi_b = np.zeros(64, dtype=np.int32)
    BL.ref_abi(P(abi_b))
    assert (abi == abi_b).all()
    np.savez(os.path.join(OUT, "abi.npz"), abi=abi[:n])

    # ---- utilhash, LCG, u01, makeSeededRandomEngine ------------------------
    hin = np.concatenate([np.arange(64, dtype=np.uint32),
                          np.array([7, 0x7fffffff, 0x80000000, 0xffffffff, 0x7ffffffe], dtype=np.uint32),
                          rng.integers(0, 2 ** 32, 955, dtype=np.uint64).astype(np.uint32)])
    hout = np.array([A.ref_utilhash(int(x)) for x in hin], dtype=np.uint32)
    assert hout[64] == 4090419040                      # SURVEY a12 probe
    BL.ref_utilhash.restype = C.c_uint
    assert (hout == np.array([BL.ref_utilhash(int(x)) for x in hin], dtype=np.uint32)).all(), "TU_A != TU_B: utilhash"
    seeds = np.concatenate([np.array([0, 1, 2147483647, 2147483648, 0xffffffff, 12345, 2147483646],
                                     dtype=np.uint32),
                            rng.integers(0, 2 ** 32, 249, dtype=np.uint64).astype(np.uint32)])
    raw = np.zeros((len(seeds), 8), dtype=np.uint32)
    u01 = np.zeros((len(seeds), 8), dtype=np.float32)
    for i, s in enumerate(seeds):
        BL.ref_rng_sequence(C.c_uint(int(s)), 8, P(raw[i]), P(u01[i]))
    assert abs(u01[5, 0] - 0.277490109) < 1e-9 and abs(u01[5, 1] - 0.725584686) < 1e-9
    key = np.stack([rng.integers(1, 5001, 512), rng.integers(0, 3840 * 2160, 512),
                    rng.integers(0, 17, 512)], axis=1).astype(np.int32)
    key[:4] = [[1, 0, 0], [5000, 8294399, 16], [1, 639999, 7], [2, 1, 1]]
    BL.ref_seeded_first_raw.restype = C.c_uint
    first = np.array([BL.ref_seeded_first_raw(int(a), int(b), int(c)) for a, b, c in key], dtype=np.uint32)
    np.savez(os.path.join(OUT, "rng.npz"), hash_in=hin, hash_out=hout, seeds=seeds, raw=raw, u01=u01,
             key=key, first_raw=first)

    # ---- scenes through the reference loader -------------------------------
    scenes = {}
    for name in ("cornell", "cornell_diffuse", "cornell_glass", "cornell_4k", "lamp_ball"):
        scenes[name] = load_scene(A, os.path.join(SCENES, name + ".txt"))
    # the repo's cornell.txt must load to the same bytes as the reference's own file
    ref_own = load_scene(A, "/root/reference/scenes/cornell.txt")
    for k in ("geoms", "materials", "camera", "camera_loaded"):
        assert scenes["cornell"][k].tobytes() == ref_own[k].tobytes(), k
    small = load_scene(A, scene_variant("cornell.txt", res=(64, 64)))
    scenes["cornell_64"] = small
    scenes["cornell_glass_64"] = load_scene(A, scene_variant("cornell_glass.txt", res=(96, 54)))
    scenes["cornell_diffuse_64"] = load_scene(A, scene_variant("cornell_diffuse.txt", res=(64, 64)))
    flat = {}
    for name, s in scenes.items():
        for k, v in s.items():
            flat["%s__%s" % (name, k)] = v
    np.savez(os.path.join(OUT, "scenes.npz"), **flat)
    cornell = scenes["cornell"]
    c = cornell["camera"][0]
    print("cornell effective camera:", c["position"], c["view"], c["up"], c["right"], c["pixelLength"])

    # ---- ray generation ------------------------------------------------------
    paths800 = np.zeros(800 * 800, dtype=po.PATH_DT)
    BL.ref_generate_rays(P(cornell["camera"]), 8, P(paths800))
    sub = paths800.reshape(800, 800)[::13, ::13].copy()
    paths64 = np.zeros(64 * 64, dtype=po.PATH_DT)
    BL.ref_generate_rays(P(small["camera"]), 8, P(paths64))
    np.savez(os.path.join(OUT, "raygen.npz"), sub800=sub, md5_800=hashlib.md5(paths800.tobytes()).hexdigest(),
             paths64=paths64)

    # ---- box / sphere tests ----------------------------------------------------
    g_in, r_in, k_out = [], [], []
    for gi, geom in enumerate(cornell["geoms"]):
        rays = adversarial_rays(rng, geom, 1024)
        # plus real camera rays and their first-bounce continuation directions
        cam_rays = np.concatenate([paths800["origin"], paths800["direction"]], axis=1)[rng.integers(0, 640000, 256)]
        rays = np.concatenate([rays, cam_rays.astype(np.float32)])
        out = np.full((len(rays), 8), -7.0, dtype=np.float32)
        out_b = out.copy()
        fn = "ref_box" if geom["type"] == po.CUBE else "ref_sphere"
        garr = np.ascontiguousarray(cornell["geoms"][gi:gi + 1])
        getattr(A, fn)(P(garr), P(rays), len(rays), P(out))
        getattr(BL, fn)(P(garr), P(rays), len(rays), P(out_b))
        assert out.tobytes() == out_b.tobytes(), "TU_A != TU_B for geom %d" % gi
        g_in.append(gi); r_in.append(rays); k_out.append(out)
    # a rotated + non-uniformly scaled cube and sphere (exercise the full matrices)
    extra = load_scene(A, _extra_scene())
    for gi, geom in enumerate(extra["geoms"]):
        rays = adversarial_rays(rng, geom, 1024)
        out = np.full((len(rays), 8), -7.0, dtype=np.float32)
        fn = "ref_box" if geom["type"] == po.CUBE else "ref_sphere"
        garr = np.ascontiguousarray(extra["geoms"][gi:gi + 1])
        getattr(A, fn)(P(garr), P(rays), len(rays), P(out))
        g_in.append(100 + gi); r_in.append(rays); k_out.append(out)
    np.savez_compressed(os.path.join(OUT, "geomtests.npz"), geom_index=np.array(g_in),
                        extra_geoms=extra["geoms"], **{"rays_%d" % i: r for i, r in enumerate(r_in)},
                        **{"out_%d" % i: o for i, o in enumerate(k_out)})

    # getPointOnRay / multiplyMV
    gp_r = rng.normal(size=(256, 6)).astype(np.float32); gp_t = rng.uniform(0, 20, 256).astype(np.float32)
    gp_o = np.zeros((256, 3), dtype=np.float32)
    for i in range(256):
        A.ref_get_point_on_ray(P(gp_r[i]), C.c_float(gp_t[i]), P(gp_o[i]))
    mv_m = rng.normal(size=(256, 16)).astype(np.float32); mv_v = rng.normal(size=(256, 4)).astype(np.float32)
    mv_v[::2, 3] = 1.0; mv_v[1::4, 3] = 0.0
    mv_o = np.zeros((256, 3), dtype=np.float32)
    for i in range(256):
        A.ref_multiply_mv(P(mv_m[i]), P(mv_v[i]), P(mv_o[i]))
    # TU_A (g++, the genuine cuda_runtime.h) == TU_B (hipcc --offload-host-only, the forwarding header) on every function both
    # contain, not only the box / sphere tests: what TU_B alone pins (sampler, ray generation, fake shader, the iteration) sits
    # on exactly these
    tmp = np.zeros(3, dtype=np.float32)
    for i in range(256):
        BL.ref_get_point_on_ray(P(gp_r[i]), C.c_float(gp_t[i]), P(tmp))
        assert tmp.tobytes() == gp_o[i].tobytes(), "TU_A != TU_B: getPointOnRay %d" % i
        BL.ref_multiply_mv(P(mv_m[i]), P(mv_v[i]), P(tmp))
        assert tmp.tobytes() == mv_o[i].tobytes(), "TU_A != TU_B: multiplyMV %d" % i
    # glm reflect / refract / intersectRayTriangle
    I = rng.normal(size=(512, 3)).astype(np.float32); I /= np.linalg.norm(I, axis=1, keepdims=True).astype(np.float32)
    Nn = rng.normal(size=(512, 3)).astype(np.float32); Nn /= np.linalg.norm(Nn, axis=1, keepdims=True).astype(np.float32)
    eta = rng.choice(np.array([1.5, 1 / 1.5, 1.33, 2.4], dtype=np.float32), 512)
    refl = np.zeros((512, 3), dtype=np.float32); refr = np.zeros((512, 3), dtype=np.float32)
    for i in range(512):
        A.ref_glm_reflect(P(I[i]), P(Nn[i]), P(refl[i]))
        A.ref_glm_refract(P(I[i]), P(Nn[i]), C.c_float(eta[i]), P(refr[i]))
        BL.ref_glm_reflect(P(I[i]), P(Nn[i]), P(tmp))
        assert tmp.tobytes() == refl[i].tobytes(), "TU_A != TU_B: glm::reflect %d" % i
        BL.ref_glm_refract(P(I[i]), P(Nn[i]), C.c_float(eta[i]), P(tmp))
        assert tmp.tobytes() == refr[i].tobytes(), "TU_A != TU_B: glm::refract %d" % i
    tri_o = rng.uniform(-3, 3, (2048, 3)).astype(np.float32)
    tri_v = rng.uniform(-2, 2, (2048, 9)).astype(np.float32)
    tgt = (tri_v[:, 0:3] * .3 + tri_v[:, 3:6] * .3 + tri_v[:, 6:9] * .4) + rng.normal(scale=.6, size=(2048, 3)).astype(np.float32)
    tri_d = (tgt - tri_o).astype(np.float32); tri_d /= np.linalg.norm(tri_d, axis=1, keepdims=True).astype(np.float32)
    tri_hit = np.zeros(2048, dtype=np.int32); tri_b = np.full((2048, 3), -7, dtype=np.float32)
    for i in range(2048):
        tri_hit[i] = A.ref_glm_ray_triangle(P(tri_o[i]), P(tri_d[i]), P(tri_v[i]), P(tri_b[i]))
    np.savez_compressed(os.path.join(OUT, "glmfuncs.npz"), gp_r=gp_r, gp_t=gp_t, gp_o=gp_o, mv_m=mv_m, mv_v=mv_v,
                        mv_o=mv_o, I=I, N=Nn, eta=eta, reflect=refl, refract=refr, tri_o=tri_o, tri_d=tri_d,
                        tri_v=tri_v, tri_hit=tri_hit, tri_b=tri_b)

    # ---- hemisphere sampler (libm and shared-trig bindings) ----------------------
    axis = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float32)
    rnd = rng.normal(size=(64, 3)).astype(np.float32); rnd /= np.linalg.norm(rnd, axis=1, keepdims=True).astype(np.float32)
    s3 = np.float32(0.57735026)
    edge = np.array([[s3, s3, s3], [-s3, s3, -s3]], dtype=np.float32)
    normals = np.repeat(np.concatenate([axis, rnd, edge]), 16, axis=0)
    hseeds = rng.integers(0, 2 ** 32, len(normals), dtype=np.uint64).astype(np.uint32)
    hseeds[:16] = np.arange(1, 17)
    hseeds[32] = 1
    h_libm = np.zeros((len(normals), 3), dtype=np.float32); h_sh = h_libm.copy()
    BL.ref_hemisphere(P(normals), P(hseeds), len(normals), P(h_libm))
    BS.ref_hemisphere(P(normals), P(hseeds), len(normals), P(h_sh))
    # SURVEY a15 probe: n=(0,1,0), seed 1, libm
    assert np.allclose(h_libm[32], [-0.509211183, 0.00474104099, -0.860628486], atol=1e-8), h_libm[32]
    np.savez(os.path.join(OUT, "hemisphere.npz"), normals=normals, seeds=hseeds, libm=h_libm, shared=h_sh)

    # ---- as-is reference behaviour: one bounce + fake shader ----------------------
    def fake_image(scene, iters):
        cam = scene["camera"]; W, H = cam[0]["resolution"]; N = int(W) * int(H)
        img = np.zeros((N, 3), dtype=np.float32)
        isx0 = None
        for it in range(1, iters + 1):
            paths = np.zeros(N, dtype=po.PATH_DT); isx = np.zeros(N, dtype=po.ISECT_DT)
            BL.ref_generate_rays(P(cam), scene["depth"], P(paths))
            BL.ref_compute_intersections(N, P(paths), P(scene["geoms"]), len(scene["geoms"]), P(isx), None)
            BL.ref_shade_fake(it, N, P(isx), P(paths), P(scene["materials"]))
            img[paths["pixelIndex"]] += paths["color"]
            if isx0 is None:
                isx0 = isx.copy()
        pbo = np.zeros((N, 4), dtype=np.uint8)
        BL.ref_send_image_to_pbo(P(pbo), int(W), int(H), iters, P(img))
        return img, pbo, isx0
    f64, p64, i64 = fake_image(small, 3)
    f800, p800, i800 = fake_image(cornell, 2)
    np.savez_compressed(os.path.join(OUT, "fakeshade.npz"), img64=f64, pbo64=p64, isect64=i64,
                        md5_img800=hashlib.md5(f800.tobytes()).hexdigest(),
                        md5_pbo800=hashlib.md5(p800.tobytes()).hexdigest(),
                        md5_isect800=hashlib.md5(i800.tobytes()).hexdigest(),
                        isect800_sub=i800.reshape(800, 800)[::13, ::13].copy())

    # ---- completion spec through the reference headers ------------------------------
    BL.ref_trace_iteration.restype = C.c_longlong
    BS.ref_trace_iteration.restype = C.c_longlong

    def completion(L, scene, iters, compact=1, keep_order=True):
        cam = scene["camera"]; W, H = cam[0]["resolution"]; N = int(W) * int(H); D = scene["depth"]
        img = np.zeros((N, 3), dtype=np.float32)
        lives, hashes, imgs, rays = [], [], [], []
        for it in range(1, iters + 1):
            live = np.zeros(64, dtype=np.int32)
            order = np.zeros((D, N), dtype=np.int32) if keep_order else None
            r = L.ref_trace_iteration(P(scene["geoms"]), len(scene["geoms"]), P(scene["materials"]), P(cam), D,
                                      it, compact, P(img), P(live), P(order) if keep_order else None)
            lives.append(live[:D].copy()); rays.append(r)
            if keep_order:
                hs = []
                for d in range(D):
                    nl = int((order[d] >= 0).sum())
                    hs.append(po.lib().pto_fnv1a_i32(P(np.ascontiguousarray(order[d])), 4, nl))
                hashes.append(hs)
            imgs.append(img.copy())
        return dict(live=np.array(lives), rays=np.array(rays), seq_hash=np.array(hashes, dtype=np.uint64),
                    images=np.array(imgs))
    comp = {}
    for tag, L in (("shared", BS), ("libm", BL)):
        for sname in ("cornell_64", "cornell_glass_64", "cornell_diffuse_64"):
            r = completion(L, scenes[sname], 4)
            for k, v in r.items():
                comp["%s__%s__%s" % (tag, sname, k)] = v
    # full-size C2: counts, hashes and image digest only (shared trig pins the GPU)
    r = completion(BS, cornell, 2)
    comp["shared__cornell__live"] = r["live"]; comp["shared__cornell__rays"] = r["rays"]
    comp["shared__cornell__seq_hash"] = r["seq_hash"]
    comp["shared__cornell__img_md5"] = np.array([hashlib.md5(x.tobytes()).hexdigest() for x in r["images"]])
    comp["shared__cornell__img_sub"] = r["images"][:, ::97].copy()
    print("C2 live counts iter1:", r["live"][0], "rays", r["rays"])
    np.savez_compressed(os.path.join(OUT, "completion.npz"), **comp)

    # ---- BASELINE configs[0] exactly as stated: cornell_diffuse.txt (400 x 400, depth 4, diffuse only), iteration 1,
    # through the reference's own headers in a single-thread loop ----
    c1s = scenes["cornell_diffuse"]
    assert tuple(c1s["camera"][0]["resolution"]) == (400, 400) and c1s["depth"] == 4
    r = completion(BS, c1s, 1, keep_order=False)
    np.savez_compressed(os.path.join(OUT, "c1.npz"), live=r["live"], rays=r["rays"],
                        img_md5=np.array(hashlib.md5(r["images"][0].tobytes()).hexdigest()),
                        img_sub=r["images"][0][::53].copy())
    print("C1 live counts:", r["live"][0], "rays", r["rays"])

    # ---- image output through the reference's saveImage / image::savePNG (+ stb_image_write) -------
    from PIL import Image as _Image
    img_in = f64.copy()                                  # 3-iteration fake-shader sum, 64x64
    img_in[5] = [-1.0, 7.5, np.nan]                      # clamp edges
    img_in[6] = [3.0, 2.999999, 0.0]
    base = os.path.join(tempfile.mkdtemp(), "out")
    A.ref_save_image(P(np.ascontiguousarray(img_in)), 64, 64, C.c_float(3.0), base.encode())
    rgb = np.asarray(_Image.open(base + ".png").convert("RGB"), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "imageout.npz"), image_sum=img_in, samples=np.float32(3.0), rgb=rgb)

    # ---- golden PNG statistic (the PNG itself stays in /root/reference) -----------
    try:
        from PIL import Image
        png = np.asarray(Image.open("/root/reference/img/REFERENCE_cornell.5000samp.png").convert("RGB"),
                         dtype=np.float32) / 255.0
        pooled = png.reshape(50, 16, 50, 16, 3).mean(axis=(1, 3))
        np.savez(os.path.join(OUT, "png_stat.npz"), pooled=pooled.astype(np.float32))
    except Exception as e:  # pragma: no cover
        print("PNG statistic skipped:", e)
    print("golden vectors written to", OUT)


def _extra_scene():
    txt = """MATERIAL 0
RGB         1 1 1
SPECEX      0
SPECRGB     0 0 0
REFL        0
REFR        0
REFRIOR     0
EMITTANCE   1

CAMERA
RES         32 32
FOVY        30
ITERATIONS  1
DEPTH       2
FILE        extra
EYE         1 2 9
LOOKAT      0 1 0
UP          0 1 0

OBJECT 0
cube
material 0
TRANS       1.5 2.25 -3
ROTAT       30 45 60
SCALE       2 .75 4.5

OBJECT 1
sphere
material 0
TRANS       -2 1 .5
ROTAT       10 -70 25
SCALE       1 2.5 .6
"""
    f = tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False)
    f.write(txt)
    f.close()
    return f.name


if __name__ == "__main__":
    main()
