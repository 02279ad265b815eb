"""ctypes bindings for the CPU oracle (oracle/libptoracle.so) and, when present,
the reference builds under oracle/_ref/.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# numpy mirrors of src/sceneStructs.h (byte-compatible)
GEOM_DT = np.dtype([("type", "<i4"), ("materialid", "<i4"), ("translation", "<f4", 3),
                    ("rotation", "<f4", 3), ("scale", "<f4", 3), ("transform", "<f4", (4, 4)),
                    ("inverseTransform", "<f4", (4, 4)), ("invTranspose", "<f4", (4, 4))])
MATERIAL_DT = np.dtype([("color", "<f4", 3), ("spec_exponent", "<f4"), ("spec_color", "<f4", 3),
                        ("hasReflective", "<f4"), ("hasRefractive", "<f4"),
                        ("indexOfRefraction", "<f4"), ("emittance", "<f4")])
CAMERA_DT = np.dtype([("resolution", "<i4", 2), ("position", "<f4", 3), ("lookAt", "<f4", 3),
                      ("view", "<f4", 3), ("up", "<f4", 3), ("right", "<f4", 3),
                      ("fov", "<f4", 2), ("pixelLength", "<f4", 2)])
PATH_DT = np.dtype([("origin", "<f4", 3), ("direction", "<f4", 3), ("color", "<f4", 3),
                    ("pixelIndex", "<i4"), ("remainingBounces", "<i4")])
ISECT_DT = np.dtype([("t", "<f4"), ("normal", "<f4", 3), ("materialId", "<i4")])
TRI_DT = np.dtype([("v0", "<f4", 3), ("v1", "<f4", 3), ("v2", "<f4", 3)])
MESH_DT = np.dtype([("geom_index", "<i4"), ("first_tri", "<i4"), ("tri_count", "<i4")])
assert GEOM_DT.itemsize == 236 and MATERIAL_DT.itemsize == 44 and CAMERA_DT.itemsize == 84
assert PATH_DT.itemsize == 44 and ISECT_DT.itemsize == 20 and TRI_DT.itemsize == 36

TRIG_LIBM, TRIG_SHARED = 0, 1
F_COMPACT, F_SORT, F_FAKESHADE, F_AA = 1, 2, 4, 8
SPHERE, CUBE, TRIMESH = 0, 1, 2


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Vec4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class Ray(C.Structure):
    _fields_ = [("origin", Vec3), ("direction", Vec3)]


class Camera(C.Structure):
    _fields_ = [("resolution", C.c_int * 2), ("position", Vec3), ("lookAt", Vec3), ("view", Vec3),
                ("up", Vec3), ("right", Vec3), ("fov", C.c_float * 2), ("pixelLength", C.c_float * 2)]


class Scene(C.Structure):
    _fields_ = [("geoms", C.c_void_p), ("ngeoms", C.c_int),
                ("materials", C.c_void_p), ("nmaterials", C.c_int),
                ("tris", C.c_void_p), ("ntris", C.c_int),
                ("meshes", C.c_void_p), ("nmeshes", C.c_int),
                ("camera", Camera), ("traceDepth", C.c_int), ("flags", C.c_int), ("trig", C.c_int),
                ("lensRadius", C.c_float), ("focalDistance", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("bounces", C.c_int), ("rays", C.c_int64), ("live", C.c_int32 * 64),
                ("seq_hash", C.c_uint64 * 64), ("sec_intersect", C.c_double),
                ("sec_shade", C.c_double), ("sec_other", C.c_double)]


BOUNCE_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p)


def build(ref=False):
    """(Re)build the oracle; with ref=True also oracle/_ref (needs /root/reference)."""
    subprocess.run(["make", "-s", "-C", HERE], check=True)
    if ref:
        subprocess.run(["make", "-s", "-C", HERE, "ref"], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "libptoracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.pto_utilhash.restype = C.c_uint32
        L.pto_utilhash.argtypes = [C.c_uint32]
        L.pto_make_seeded_engine.restype = C.c_uint32
        L.pto_make_seeded_engine.argtypes = [C.c_int, C.c_int, C.c_int]
        L.pto_lcg_seed.restype = C.c_uint32
        L.pto_lcg_seed.argtypes = [C.c_uint32]
        L.pto_lcg_next.restype = C.c_uint32
        L.pto_lcg_next.argtypes = [C.POINTER(C.c_uint32)]
        L.pto_u01.restype = C.c_float
        L.pto_u01.argtypes = [C.POINTER(C.c_uint32)]
        L.pto_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.pto_get_point_on_ray.restype = Vec3
        L.pto_get_point_on_ray.argtypes = [Ray, C.c_float]
        L.pto_multiply_mv.restype = Vec3
        L.pto_multiply_mv.argtypes = [C.c_void_p, Vec4]
        for f in (L.pto_box_test, L.pto_sphere_test):
            f.restype = C.c_float
            f.argtypes = [C.c_void_p, Ray, C.POINTER(Vec3), C.POINTER(Vec3), C.POINTER(C.c_int)]
        L.pto_ray_triangle.restype = C.c_int
        L.pto_ray_triangle.argtypes = [Vec3, Vec3, Vec3, Vec3, Vec3, C.POINTER(Vec3)]
        L.pto_mesh_pad.restype = C.c_float
        L.pto_mesh_pad.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.pto_tri_point_ok.restype = C.c_int
        L.pto_tri_point_ok.argtypes = [Vec3, Vec3, C.c_float, C.c_void_p, C.c_float]
        L.pto_mesh_winners.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.pto_hemisphere.restype = Vec3
        L.pto_hemisphere.argtypes = [Vec3, C.POINTER(C.c_uint32), C.c_int]
        L.pto_reflect.restype = Vec3
        L.pto_reflect.argtypes = [Vec3, Vec3]
        L.pto_generate_rays.argtypes = [C.POINTER(Camera), C.c_int, C.c_void_p]
        L.pto_generate_rays_ex.argtypes = [C.POINTER(Scene), C.c_int, C.c_void_p]
        L.pto_compute_intersections.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.pto_shade_fake.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pto_final_gather.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.pto_send_image_to_pbo.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.pto_shade_scatter.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int]
        L.pto_compact.restype = C.c_int
        L.pto_compact.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.pto_sort_by_material.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pto_trace_iteration.argtypes = [C.POINTER(Scene), C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.POINTER(Stats), C.c_void_p, C.c_void_p]
        L.pto_trace_iteration_mt.argtypes = [C.POINTER(Scene), C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.POINTER(Stats), C.c_int]
        L.pto_trace_iterations_parallel.restype = C.c_int64
        L.pto_trace_iterations_parallel.argtypes = [C.POINTER(Scene), C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.pto_trace_rows_mt.argtypes = [C.POINTER(Scene), C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(Stats), C.c_int]
        L.pto_mesh_accepted.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.pto_fnv1a_i32.restype = C.c_uint64
        L.pto_fnv1a_i32.argtypes = [C.c_void_p, C.c_int, C.c_int]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def vec3(v):
    return Vec3(float(v[0]), float(v[1]), float(v[2]))


def ray(o, d):
    return Ray(vec3(o), vec3(d))


def camera_struct(cam_np):
    """CAMERA_DT scalar/0-d array -> ctypes Camera."""
    c = Camera()
    C.memmove(C.byref(c), np.ascontiguousarray(cam_np).tobytes(), 84)
    return c


def u01_sequence(seed, n):
    L = lib()
    st = C.c_uint32(L.pto_lcg_seed(seed))
    return np.array([L.pto_u01(C.byref(st)) for _ in range(n)], dtype=np.float32)


def raw_sequence(seed, n):
    L = lib()
    st = C.c_uint32(L.pto_lcg_seed(seed))
    return np.array([L.pto_lcg_next(C.byref(st)) for _ in range(n)], dtype=np.uint32)


def sincos(x):
    s, c = C.c_float(), C.c_float()
    lib().pto_sincos(C.c_float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def sincos_sums(first_bits, count):
    """CPU side of ptmi355.probe_sincos_sums: (sum of bits(sin) * (2k+1), the same for cos) mod 2^64 over `count`
    consecutive float32 values from bit pattern `first_bits` on."""
    out = (C.c_uint64 * 2)()
    lib().pto_sincos_sums(C.c_uint32(first_bits), C.c_uint32(count), out)
    return int(out[0]), int(out[1])


def geom_test(geom_np, rays_np, kind):
    """rays_np: (n,6) float32 -> (n,8) float32 [t, p3, n3, outside]; sentinel -7 where untouched."""
    L = lib()
    fn = L.pto_box_test if kind == CUBE else L.pto_sphere_test
    g = np.ascontiguousarray(geom_np)
    out = np.full((len(rays_np), 8), -7.0, dtype=np.float32)
    for i, r in enumerate(rays_np):
        p, n, o = Vec3(-7, -7, -7), Vec3(-7, -7, -7), C.c_int(1)  # bool sentinel: true
        t = fn(_p(g), ray(r[:3], r[3:]), C.byref(p), C.byref(n), C.byref(o))
        out[i] = [t, p.x, p.y, p.z, n.x, n.y, n.z, float(o.value)]
    return out


def hemisphere(normals, seeds, trig):
    L = lib()
    out = np.zeros((len(seeds), 3), dtype=np.float32)
    for i, (nrm, s) in enumerate(zip(normals, seeds)):
        st = C.c_uint32(L.pto_lcg_seed(int(s)))
        v = L.pto_hemisphere(vec3(nrm), C.byref(st), trig)
        out[i] = [v.x, v.y, v.z]
    return out


def generate_rays(cam_np, trace_depth):
    cam = camera_struct(cam_np)
    n = int(cam.resolution[0]) * int(cam.resolution[1])
    paths = np.zeros(n, dtype=PATH_DT)
    lib().pto_generate_rays(C.byref(cam), trace_depth, _p(paths))
    return paths


def generate_rays_ex(cam_np, trace_depth, it, aa=False, lens=(0.0, 0.0), trig=TRIG_SHARED):
    """Camera rays of iteration `it` with pixel jitter and / or a thin lens (completion spec)."""
    sc = Scene()
    sc.camera = camera_struct(cam_np)
    sc.traceDepth, sc.flags, sc.trig = trace_depth, F_AA if aa else 0, trig
    sc.lensRadius, sc.focalDistance = lens
    n = int(sc.camera.resolution[0]) * int(sc.camera.resolution[1])
    paths = np.zeros(n, dtype=PATH_DT)
    lib().pto_generate_rays_ex(C.byref(sc), it, _p(paths))
    return paths


def compute_intersections(paths, geoms, tris=None, meshes=None, n=None):
    n = len(paths) if n is None else n
    isects = np.zeros(len(paths), dtype=ISECT_DT)
    outside = np.zeros(len(paths), dtype=np.uint8)
    lib().pto_compute_intersections(n, _p(paths), _p(geoms), len(geoms), _p(tris), _p(meshes),
                                    0 if meshes is None else len(meshes), _p(isects), _p(outside))
    return isects, outside


def mesh_pad(tris, first=0, count=None):
    """Pad of the spec's hit-point test for the mesh tris[first : first + count]."""
    count = len(tris) - first if count is None else count
    return np.float32(lib().pto_mesh_pad(_p(tris), first, count))


def mesh_winners(tris, paths, first=0, count=None):
    """Winning triangle index (-1: none) and its bary.z for every ray: the loop over all triangles."""
    count = len(tris) - first if count is None else count
    idx = np.zeros(len(paths), dtype=np.int32)
    tz = np.zeros(len(paths), dtype=np.float32)
    lib().pto_mesh_winners(_p(tris), first, count, _p(paths), len(paths), _p(idx), _p(tz))
    return idx, tz


def mesh_accepted(tris, paths, first=0, count=None):
    """accepted[ray, triangle] = the spec counts this triangle for this ray (glm::intersectRayTriangle, bary.z > 0, the
    hit-point test) -- every accepted pair, not only the winner."""
    count = len(tris) - first if count is None else count
    out = np.zeros((len(paths), count), dtype=np.uint8)
    lib().pto_mesh_accepted(_p(tris), first, count, _p(paths), len(paths), _p(out))
    return out


def make_scene(geoms, materials, cam_np, trace_depth, flags=F_COMPACT, trig=TRIG_SHARED,
               tris=None, meshes=None, lens=(0.0, 0.0)):
    sc = Scene()
    sc.geoms, sc.ngeoms = _p(geoms), len(geoms)
    sc.materials, sc.nmaterials = _p(materials), len(materials)
    sc.tris, sc.ntris = _p(tris), 0 if tris is None else len(tris)
    sc.meshes, sc.nmeshes = _p(meshes), 0 if meshes is None else len(meshes)
    sc.camera = camera_struct(cam_np)
    sc.traceDepth, sc.flags, sc.trig = trace_depth, flags, trig
    sc.lensRadius, sc.focalDistance = lens
    sc._keep = (geoms, materials, tris, meshes)
    return sc


class Tracer:
    """Stateful wrapper: accumulates `image` (running sum) across iterations."""

    def __init__(self, geoms, materials, cam_np, trace_depth, flags=F_COMPACT, trig=TRIG_SHARED,
                 tris=None, meshes=None, lens=(0.0, 0.0)):
        self.scene = make_scene(geoms, materials, cam_np, trace_depth, flags, trig, tris, meshes, lens)
        self.n = int(self.scene.camera.resolution[0]) * int(self.scene.camera.resolution[1])
        self.image = np.zeros((self.n, 3), dtype=np.float32)
        self.paths = np.zeros(self.n, dtype=PATH_DT)
        self.isects = np.zeros(self.n, dtype=ISECT_DT)

    def iterate_parallel(self, iter0, count, threads):
        """`count` iterations, one whole iteration per thread; adds them to self.image in iteration order."""
        rays = lib().pto_trace_iterations_parallel(C.byref(self.scene), iter0, count, _p(self.image), threads)
        if rays < 0:
            raise MemoryError("pto_trace_iterations_parallel")
        return rays

    def iterate_rows(self, it, y0, y1, threads=1):
        """Iteration `it` for the pixels of rows [y0, y1) only (what a frame tile owns)."""
        st = Stats()
        lib().pto_trace_rows_mt(C.byref(self.scene), it, _p(self.image), y0, y1, C.byref(st), threads)
        return st

    def iterate(self, it, snapshots=None, threads=0):
        """Run iteration `it` (1-based). If `snapshots` is a list, append per-bounce
        dicts {depth, n_before, n_live, paths (copy of [0,n_before)), isects}."""
        st = Stats()
        if threads and threads > 0:
            lib().pto_trace_iteration_mt(C.byref(self.scene), it, _p(self.image), _p(self.paths),
                                         _p(self.isects), C.byref(st), threads)
            return st
        cb = None
        if snapshots is not None:
            def _cb(user, depth, n_before, n_live, paths_p, isects_p):
                snapshots.append(dict(depth=depth, n_before=n_before, n_live=n_live,
                                      paths=self.paths[:n_before].copy(),
                                      isects=self.isects[:n_before].copy()))
            cb = BOUNCE_CB(_cb)
        lib().pto_trace_iteration(C.byref(self.scene), it, _p(self.image), _p(self.paths),
                                  _p(self.isects), C.byref(st),
                                  C.cast(cb, C.c_void_p) if cb else None, None)
        return st


# ---------------------------------------------------------------------------
# reference builds (this container only)
# ---------------------------------------------------------------------------
def ref_available():
    return all(os.path.exists(os.path.join(HERE, "_ref", f))
               for f in ("libptref_a.so", "libptref_b_libm.so", "libptref_b_shared.so"))


_refs = {}


def ref(which):
    """which in {'a', 'b_libm', 'b_shared'}"""
    if which not in _refs:
        if which == "b_shared":
            lib()  # make sure libptoracle.so is resident for pto_sincos
            C.CDLL(os.path.join(HERE, "libptoracle.so"), mode=C.RTLD_GLOBAL)
        L = C.CDLL(os.path.join(HERE, "_ref", "libptref_%s.so" % which))
        L.ref_utilhash.restype = C.c_uint32
        L.ref_utilhash.argtypes = [C.c_uint32]
        _refs[which] = L
    return _refs[which]
