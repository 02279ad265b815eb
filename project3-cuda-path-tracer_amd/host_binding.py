"""ctypes binding of libpthost.so (host/pthost.cpp): the headless counterpart of the reference's
scene loader (src/scene.cpp), runCuda's camera set-up (src/main.cpp:53-67,102-120) and
saveImage / image::savePNG (src/main.cpp:78-99, src/image.cpp:22-39)."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np

from .binding import CAMERA_DT, GEOM_DT, MATERIAL_DT, MESH_DT, TRI_DT, PtError, Scene

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libpthost.so")
SRC = [os.path.join(HERE, "host", "pthost.cpp"), os.path.join(HERE, "host", "pthost.h")]


class _PthScene(C.Structure):
    _fields_ = [("geoms", C.c_void_p), ("num_geoms", C.c_int32),
                ("materials", C.c_void_p), ("num_materials", C.c_int32),
                ("triangles", C.c_void_p), ("num_triangles", C.c_int32),
                ("meshes", C.c_void_p), ("num_meshes", C.c_int32),
                ("camera_loaded", C.c_uint8 * 84), ("camera", C.c_uint8 * 84),
                ("iterations", C.c_int32), ("trace_depth", C.c_int32), ("image_name", C.c_char * 256)]


def build_host(force=False):
    stale = force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in SRC)
    if stale:
        cxx = shutil.which("g++") or "g++"
        subprocess.run([cxx, "-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-shared", "-o", LIB, SRC[0]],
                       check=True, cwd=HERE)
    return LIB


PTBENCH = os.path.join(HERE, "ptbench")


def build_ptbench(force=False):
    """The headless host binary (host/ptbench.cpp): links libptmi355.so by rpath."""
    src = [os.path.join(HERE, "host", "ptbench.cpp"), SRC[0], SRC[1], os.path.join(HERE, "libptmi355.so")]
    stale = force or not os.path.exists(PTBENCH) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(PTBENCH) for s in src)
    if stale:
        cxx = shutil.which("g++") or "g++"
        subprocess.run([cxx, "-std=c++17", "-O2", "-ffp-contract=off", "-o", PTBENCH, src[0], src[1],
                        "-L" + HERE, "-lptmi355", "-Wl,-rpath,$ORIGIN"], check=True, cwd=HERE)
    return PTBENCH


_lib = None


def host_library():
    global _lib
    if _lib is None:
        L = C.CDLL(build_host())
        L.pth_load_scene.restype = C.POINTER(_PthScene)
        L.pth_load_scene.argtypes = [C.c_char_p]
        L.pth_free_scene.argtypes = [C.POINTER(_PthScene)]
        L.pth_last_error.restype = C.c_char_p
        L.pth_build_geom_matrices.argtypes = [C.c_void_p]
        L.pth_image_to_rgb8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.pth_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
        L.pth_write_pfm.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
        L.pth_read_pfm.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
        _lib = L
    return _lib


def _copy(ptr, n, dt):
    if n == 0:
        return np.zeros(0, dtype=dt)
    buf = (C.c_uint8 * (n * dt.itemsize)).from_address(ptr)
    return np.frombuffer(bytes(buf), dtype=dt).copy()


def load_scene(path):
    """Scene::Scene(path) + the camera recompute runCuda does before the first pathtrace()."""
    L = host_library()
    p = L.pth_load_scene(os.fsencode(path))
    if not p:
        raise PtError("load_scene(%s): %s" % (path, L.pth_last_error().decode()))
    s = p.contents
    try:
        geoms = _copy(s.geoms, s.num_geoms, GEOM_DT)
        mats = _copy(s.materials, s.num_materials, MATERIAL_DT)
        tris = _copy(s.triangles, s.num_triangles, TRI_DT) if s.num_triangles else None
        meshes = _copy(s.meshes, s.num_meshes, MESH_DT) if s.num_meshes else None
        cam = np.frombuffer(bytes(s.camera), dtype=CAMERA_DT).copy()
        cam_loaded = np.frombuffer(bytes(s.camera_loaded), dtype=CAMERA_DT).copy()
        scene = Scene(geoms, mats, cam, s.trace_depth, iterations=s.iterations, triangles=tris, meshes=meshes,
                      name=s.image_name.decode())
        scene.camera_loaded = cam_loaded
    finally:
        L.pth_free_scene(p)
    return scene


def image_to_rgb8(image_sum, width, height, samples):
    """saveImage's pixel pipeline: sum / samples, x-flip, clamp, * 255.f, truncate -> (H, W, 3) uint8."""
    img = np.ascontiguousarray(image_sum, dtype=np.float32).reshape(-1)
    out = np.zeros((height, width, 3), dtype=np.uint8)
    host_library().pth_image_to_rgb8(img.ctypes.data, width, height, C.c_float(samples), out.ctypes.data)
    return out


def save_png(path, image_sum, width, height, samples):
    rgb = image_to_rgb8(image_sum, width, height, samples)
    if host_library().pth_write_png(os.fsencode(path), rgb.ctypes.data, width, height) != 0:
        raise PtError(host_library().pth_last_error().decode())
    return rgb


def save_pfm(path, image_sum, width, height, samples):
    img = np.ascontiguousarray(image_sum, dtype=np.float32).reshape(-1)
    if host_library().pth_write_pfm(os.fsencode(path), img.ctypes.data, width, height, C.c_float(samples)) != 0:
        raise PtError(host_library().pth_last_error().decode())


def load_pfm(path, width, height):
    """The floats of a little-endian colour PFM (a running sum saved with samples = 1 comes back exactly)."""
    img = np.zeros((width * height, 3), dtype=np.float32)
    if host_library().pth_read_pfm(os.fsencode(path), img.ctypes.data, width, height) != 0:
        raise PtError(host_library().pth_last_error().decode())
    return img
