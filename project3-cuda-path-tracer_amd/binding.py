"""ctypes binding of libptmi355.so.  Mirrors the reference's renderer interface:

    pathtraceInit(scene)            src/pathtrace.h:6   (pathtrace.cu:79-98)
    pathtraceFree()                 src/pathtrace.h:7   (pathtrace.cu:100-112)
    pathtrace(pbo, frame, iter)     src/pathtrace.h:8   (pathtrace.cu:284-393)

`Scene` carries what the reference's Scene/RenderState carry for this path
(scene.h:13-26, sceneStructs.h:54-60) as numpy arrays with the reference's
struct layouts.  There is no CPU fallback: if the HIP library is missing or no
GPU is present, calls raise PtError.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# byte-compatible with src/sceneStructs.h (x86-64): 236 / 44 / 84 / 44 / 20 bytes
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
MESH_DT = np.dtype([("geom_index", "<i4"), ("first_triangle", "<i4"), ("triangle_count", "<i4")])

PT_COMPACT, PT_SORT_MATERIAL, PT_FAKE_SHADER, PT_CACHE_FIRST, PT_UNFUSED, PT_MESH_BVH, PT_AA_JITTER, PT_ASYNC_IMAGE, PT_PIN_IMAGE = 1, 2, 4, 8, 16, 32, 64, 128, 256
PT_HOST_SPARSE = 1024
PT_SHARED_IMAGE = 512
PT_LOOKAHEAD = 2048         # pt_trace traces ahead of its caller (include/ptmi355.h)
BVH_NODE_WORDS = 16


class PtError(RuntimeError):
    pass


class _Camera(C.Structure):
    _fields_ = [("raw", C.c_uint8 * 84)]


class _SceneDesc(C.Structure):
    _fields_ = [("geoms", C.c_void_p), ("num_geoms", C.c_int32),
                ("materials", C.c_void_p), ("num_materials", C.c_int32),
                ("triangles", C.c_void_p), ("num_triangles", C.c_int32),
                ("meshes", C.c_void_p), ("num_meshes", C.c_int32),
                ("camera", _Camera), ("trace_depth", C.c_int32), ("flags", C.c_uint32),
                ("device", C.c_int32), ("stream", C.c_void_p),
                ("tile_index", C.c_int32), ("tile_count", C.c_int32), ("strip_rows", C.c_int32),
                ("max_batch", C.c_int32), ("device_image", C.c_void_p),
                ("lens_radius", C.c_float), ("focal_distance", C.c_float),
                ("devices", C.c_void_p), ("num_devices", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("bounces", C.c_int32), ("rays", C.c_int64), ("live", C.c_int32 * 64),
                ("total_rays", C.c_int64), ("total_iterations", C.c_int64)]


class BvhInfo(C.Structure):
    _fields_ = [("nodes", C.c_int32), ("triangles", C.c_int32), ("depth", C.c_int32),
                ("pad", C.c_float), ("prune", C.c_float)]


class Profile(C.Structure):
    _fields_ = [("ms", C.c_double * 6), ("launches", C.c_int64 * 6)]


STAGES = ("raygen", "bounce", "intersect", "sort", "gather", "mesh")


class Scene:
    """What pathtraceInit reads from the reference's Scene* (scene.h:23-25)."""

    def __init__(self, geoms, materials, camera, trace_depth, iterations=1, triangles=None,
                 meshes=None, name="scene"):
        self.geoms = np.ascontiguousarray(geoms, dtype=GEOM_DT)
        self.materials = np.ascontiguousarray(materials, dtype=MATERIAL_DT)
        self.camera = np.ascontiguousarray(camera, dtype=CAMERA_DT).reshape(1)
        self.traceDepth = int(trace_depth)
        self.iterations = int(iterations)
        self.triangles = None if triangles is None else np.ascontiguousarray(triangles, dtype=TRI_DT)
        self.meshes = None if meshes is None else np.ascontiguousarray(meshes, dtype=MESH_DT)
        self.name = name
        w, h = self.camera[0]["resolution"]
        self.image = np.zeros((int(w) * int(h), 3), dtype=np.float32)     # state.image (running sum)

    @property
    def resolution(self):
        w, h = self.camera[0]["resolution"]
        return int(w), int(h)


_lib = None
_scene = None
_host_sparse = False


def library():
    """Load libptmi355.so; raise loudly if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        # PTMI355_LIB: another build of the same library (A/B measurements of kernel variants within one GPU box)
        path = os.environ.get("PTMI355_LIB") or os.path.join(HERE, "libptmi355.so")
        if not os.path.exists(path):
            raise PtError("libptmi355.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)")
        # torch is the plumbing for streams / device buffers / RCCL.  Import it first so the HIP
        # runtime it bundles is the one (and only one) resident in the process: loading the
        # system libamdhip64 first and torch's copy second leaves torch without devices.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(path)
        L.pt_last_error.restype = C.c_char_p
        L.pt_version.restype = C.c_char_p
        L.pt_device_image.restype = C.c_void_p
        L.pt_init.argtypes = [C.POINTER(_SceneDesc)]
        L.pt_set_camera.argtypes = [C.c_void_p, C.c_int]
        L.pt_set_lens.argtypes = [C.c_float, C.c_float]
        L.pt_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.pt_trace_batch.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.pt_trace_batch_async.argtypes = [C.c_int, C.c_int]
        L.pt_trace_begin.argtypes = [C.c_int, C.c_int]
        L.pt_trace_bounce.argtypes = [C.c_int, C.POINTER(C.c_int)]
        L.pt_export_paths.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.pt_export_intersections.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.pt_intersect_once.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.pt_get_image.argtypes = [C.c_void_p]
        L.pt_tonemap.argtypes = [C.c_void_p, C.c_int]
        L.pt_get_stats.argtypes = [C.POINTER(Stats)]
        L.pt_total_rays.restype = C.c_int64
        L.pt_get_counters.argtypes = [C.POINTER(C.c_int64)] * 3
        L.pt_set_profiling.argtypes = [C.c_int]
        L.pt_get_profile.argtypes = [C.POINTER(Profile)]
        L.pt_get_bvh_info.argtypes = [C.POINTER(BvhInfo)]
        L.pt_bvh_build.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.pt_cull_boxes.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_void_p]
        L.pt_tri_bounds.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p]
        L.pt_tri_records.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        try:
            L.pt_set_image.argtypes = [C.c_void_p]
            L.pt_probe_rng.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
            L.pt_probe_sincos.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
            L.pt_probe_hemisphere.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
            L.pt_probe_sqrt.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
            L.pt_probe_clock.argtypes = [C.c_int, C.POINTER(C.c_double)]
        except AttributeError:
            if not os.environ.get("PTMI355_LIB"):        # only an older A/B build (profiles/tools/ab.sh) may lack them
                raise
        L.pt_free.restype = None
        L.pt_exchange_transport.restype = C.c_char_p
        _lib = L
    return _lib


def _chk(rc):
    if rc < 0:
        raise PtError("ptmi355 error %d: %s" % (rc, library().pt_last_error().decode()))
    return rc


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def version():
    return library().pt_version().decode()


def has_experiments():
    """True for a -DPT_EXPERIMENTS build (profiles/tools/build_variant.sh, loaded through PTMI355_LIB): the library then
    reads the experiment / test-hook environment variables of rounds 1-4; the shipped build reads the ten documented in
    include/ptmi355.h and nothing else."""
    return "+experiments" in version()


def pathtraceInit(scene, flags=PT_COMPACT, device=0, stream=None, tile=(0, 1, 8), max_batch=1,
                  device_image=None, lens=(0.0, 0.0), devices=None, pin_image=True, host_sparse=False):
    """pathtraceInit(Scene*) (pathtrace.cu:79-98) + the run-time toggles of include/ptmi355.h.
    devices=[d0, d1, ...]: the frame tiled over several GPUs inside the library (tile[2] = rows per strip).
    pin_image: PT_PIN_IMAGE -- pathtrace() below always hands over scene.image, which lives as long as the scene
    (like the reference's scene->state.image); callers that pass their own short-lived buffers to pt_trace say False.
    host_sparse: PT_HOST_SPARSE -- the caller only READS the image between calls, so a call writes just the pixels whose
    sum changed; pathtrace() then returns a read-only view of scene.image (a host that scribbles on it gets an error
    instead of stale pixels)."""
    global _scene, _host_sparse
    _host_sparse = bool(host_sparse or (flags & PT_HOST_SPARSE))
    d = _SceneDesc()
    d.geoms, d.num_geoms = _p(scene.geoms), len(scene.geoms)
    d.materials, d.num_materials = _p(scene.materials), len(scene.materials)
    d.triangles, d.num_triangles = _p(scene.triangles), 0 if scene.triangles is None else len(scene.triangles)
    d.meshes, d.num_meshes = _p(scene.meshes), 0 if scene.meshes is None else len(scene.meshes)
    C.memmove(C.byref(d.camera), scene.camera.tobytes(), 84)
    d.trace_depth, d.device = scene.traceDepth, device
    d.flags = flags | (PT_PIN_IMAGE if pin_image else 0) | (PT_HOST_SPARSE if host_sparse else 0)
    d.stream = stream
    d.tile_index, d.tile_count, d.strip_rows = tile
    d.max_batch = max_batch
    d.device_image = device_image
    d.lens_radius, d.focal_distance = lens
    devs = None
    if devices is not None:
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        d.devices, d.num_devices = _p(devs), len(devs)
    _chk(library().pt_init(C.byref(d)))
    _scene = scene


def pathtraceFree():
    """pathtraceFree() (pathtrace.cu:100-112): idempotent."""
    global _scene
    library().pt_free()
    _scene = None


def pathtrace(pbo, frame, iteration, copy_image=True):
    """pathtrace(uchar4 *pbo, int frame, int iter) (pathtrace.cu:284-393).  `pbo` is a device
    pointer (int) or None.  Like the reference (pathtrace.cu:285-286, 389-390) it re-reads the
    camera / traceDepth from the scene and refreshes scene.image (the running sum)."""
    if _scene is None:
        raise PtError("pathtrace: pathtraceInit has not been called")
    L = library()
    _chk(L.pt_set_camera(_p(_scene.camera), _scene.traceDepth))
    _chk(L.pt_trace(pbo, frame, iteration, _p(_scene.image) if copy_image else None))
    if _host_sparse:
        view = _scene.image.view()
        view.setflags(write=False)
        return view
    return _scene.image


def set_lens(lens_radius, focal_distance):
    _chk(library().pt_set_lens(lens_radius, focal_distance))


def set_camera(camera, trace_depth):
    cam = np.ascontiguousarray(camera, dtype=CAMERA_DT).reshape(1)
    _chk(library().pt_set_camera(_p(cam), trace_depth))


def trace_batch(iter0, count, host_image=None):
    _chk(library().pt_trace_batch(iter0, count, _p(host_image)))


def trace_batch_async(iter0, count):
    _chk(library().pt_trace_batch_async(iter0, count))


def synchronize():
    _chk(library().pt_synchronize())


def trace_begin(iter0, count=1):
    _chk(library().pt_trace_begin(iter0, count))


def trace_bounce(depth):
    n = C.c_int(0)
    _chk(library().pt_trace_bounce(depth, C.byref(n)))
    return n.value


def trace_end():
    _chk(library().pt_trace_end())


def export_paths(capacity):
    buf = np.zeros(capacity, dtype=PATH_DT)
    live = C.c_int(0)
    n = _chk(library().pt_export_paths(_p(buf), capacity, C.byref(live)))
    return buf[:n], live.value


def export_intersections(capacity):
    buf = np.zeros(capacity, dtype=ISECT_DT)
    out = np.zeros(capacity, dtype=np.uint8)
    n = _chk(library().pt_export_intersections(_p(buf), _p(out), capacity))
    return buf[:n], out[:n]


def intersect_once(paths):
    paths = np.ascontiguousarray(paths, dtype=PATH_DT)
    isects = np.zeros(len(paths), dtype=ISECT_DT)
    outside = np.zeros(len(paths), dtype=np.uint8)
    _chk(library().pt_intersect_once(_p(paths), len(paths), _p(isects), _p(outside)))
    return isects, outside


def get_image(npix):
    img = np.zeros((npix, 3), dtype=np.float32)
    _chk(library().pt_get_image(_p(img)))
    return img


def tonemap(npix, iteration):
    out = np.zeros((npix, 4), dtype=np.uint8)
    _chk(library().pt_tonemap(_p(out), iteration))
    return out


def clear_image():
    _chk(library().pt_clear_image())


def set_image(image_sum):
    """Resume an accumulation: the running sum becomes `image_sum` (W*H*3 floats)."""
    img = np.ascontiguousarray(image_sum, dtype=np.float32)
    _chk(library().pt_set_image(_p(img)))


def device_image_ptr():
    return library().pt_device_image()


def num_devices():
    return library().pt_num_devices()


def exchange_transport():
    return library().pt_exchange_transport().decode()


def get_stats():
    s = Stats()
    _chk(library().pt_get_stats(C.byref(s)))
    return s


def set_profiling(enable):
    _chk(library().pt_set_profiling(1 if enable else 0))


def get_profile():
    """{stage: (summed ms, launches)} measured with HIP events on the launch stream."""
    p = Profile()
    _chk(library().pt_get_profile(C.byref(p)))
    return {name: (p.ms[i], p.launches[i]) for i, name in enumerate(STAGES)}


def bvh_info():
    info = BvhInfo()
    _chk(library().pt_get_bvh_info(C.byref(info)))
    return info


def bvh_build(triangles):
    """Host-only: the hierarchy pt_init builds under PT_MESH_BVH.  Returns (nodes[n, 16] uint32 -- layout in
    csrc/pt_bvh.hpp --, order[count] int32, grid[8] float32 = origin xyz, step xyz, padding, prune margin)."""
    tris = np.ascontiguousarray(triangles, dtype=TRI_DT)
    L = library()
    need = L.pt_bvh_build(_p(tris), len(tris), None, 0, None, None)
    if need < 0:
        raise PtError(L.pt_last_error().decode())
    nodes = np.zeros((need, BVH_NODE_WORDS), dtype=np.uint32)
    order = np.zeros(max(1, len(tris)), dtype=np.int32)
    grid = np.zeros(8, dtype=np.float32)
    _chk(min(0, L.pt_bvh_build(_p(tris), len(tris), _p(nodes), need, _p(order), _p(grid))))
    return nodes, order[:len(tris)], grid


def cull_boxes(geoms, eye=(0.0, 0.0, 0.0)):
    """Host-only: (boxes[n, 2, 3] float32 = lo / hi, origin bound, reject[n, 5] = mode / row of the inverseTransform)
    pt_init derives for the cull stage."""
    g = np.ascontiguousarray(geoms, dtype=GEOM_DT)
    e = np.asarray(eye, dtype=np.float32)
    out = np.zeros((len(g), 2, 3), dtype=np.float32)
    rej = np.zeros((len(g), 5), dtype=np.float32)
    r = C.c_float(0.0)
    _chk(library().pt_cull_boxes(_p(g), len(g), _p(e), _p(out), C.byref(r), _p(rej)))
    return out, r.value, rej


def tri_bounds(triangles, origin_bound):
    """Host-only: {centre xyz, Rs^2} per triangle of one mesh (the every-triangle loop's first stage)."""
    t = np.ascontiguousarray(triangles, dtype=TRI_DT)
    out = np.zeros(((len(t) + 3) & ~3, 4), dtype=np.float32)
    _chk(library().pt_tri_bounds(_p(t), len(t), float(origin_bound), _p(out)))
    return out[:len(t)]


def tri_records(triangles, origin_bound):
    """Host-only: the triangles' side of the bilinear form the every-triangle loop's first stage evaluates on the matrix pipe:
    (records [n64, 32] float16, frame {gx, gy, gz, 1 / Rm})."""
    t = np.ascontiguousarray(triangles, dtype=TRI_DT)
    n64 = (len(t) + 63) & ~63
    rec = np.zeros((max(n64, 1), 32), dtype=np.float16)
    frame = np.zeros(4, dtype=np.float32)
    n = _chk(library().pt_tri_records(_p(t), len(t), float(origin_bound), _p(rec), _p(frame)))
    return rec[:n], frame


def total_rays():
    """Rays traced since pathtraceInit (device-side counter; synchronises)."""
    r = library().pt_total_rays()
    if r < 0:
        _chk(int(r))
    return int(r)


def counters():
    """(rays, first-bounce rays, iterations) since pathtraceInit, from the device; synchronises."""
    a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
    _chk(library().pt_get_counters(C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def probe_rng(seeds, draws):
    """The device's minstd_rand + u01 (csrc/pt_device.hpp): (engine state, last u01) after `draws` draws per seed."""
    sd = np.ascontiguousarray(seeds, dtype=np.uint32)
    st = np.zeros(len(sd), dtype=np.uint32)
    u = np.zeros(len(sd), dtype=np.float32)
    _chk(library().pt_probe_rng(_p(sd), len(sd), int(draws), _p(st), _p(u)))
    return st, u


def probe_sincos(x):
    """The device's shared sin / cos for the float32 arguments x."""
    xs = np.ascontiguousarray(x, dtype=np.float32)
    s = np.zeros(len(xs), dtype=np.float32)
    c = np.zeros(len(xs), dtype=np.float32)
    _chk(library().pt_probe_sincos(_p(xs), 0, len(xs), _p(s), _p(c), None))
    return s, c


def probe_sincos_sums(first_bits, count):
    """(sum of bits(sin) * (2k+1), the same for cos) mod 2^64 over the `count` consecutive float32 values from bit
    pattern `first_bits` on -- what oracle.pyoracle.sincos_sums computes on the CPU."""
    out = np.zeros(2, dtype=np.uint64)
    _chk(library().pt_probe_sincos(None, int(first_bits), int(count), None, None, _p(out)))
    return int(out[0]), int(out[1])


def probe_sqrt(first_bits, count):
    """(arguments whose sqrt differs, whose 1 / sqrt differs) between the kernels' Newton forms and the correctly rounded sqrtf /
    divide, over the `count` consecutive float32 values from bit pattern `first_bits` on."""
    out = np.zeros(2, dtype=np.uint64)
    _chk(library().pt_probe_sqrt(int(first_bits), int(count), _p(out)))
    return int(out[0]), int(out[1])


def probe_clock(microseconds=200):
    """The shader clock in GHz while whatever is enqueued keeps running (one wave, cycle counter against the 100-MHz counter)."""
    ghz = C.c_double(0.0)
    _chk(library().pt_probe_clock(int(microseconds), C.byref(ghz)))
    return float(ghz.value)


def probe_hemisphere(normals, seeds):
    """calculateRandomDirectionInHemisphere on the device for (normal, engine seed) pairs."""
    nr = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
    sd = np.ascontiguousarray(seeds, dtype=np.uint32)
    out = np.zeros((len(sd), 3), dtype=np.float32)
    _chk(library().pt_probe_hemisphere(_p(nr), _p(sd), len(sd), _p(out)))
    return out
