"""Procedural triangle meshes for the PT_TRIANGLE_MESH extension (BASELINE config C4:
"Cornell + 100k-triangle mesh, naive triangle loop").  Deterministic, seedless; vertices are
emitted in world space as float32 (the library takes world-space triangles)."""
import numpy as np

from .binding import GEOM_DT, MESH_DT, TRI_DT


def uv_sphere(center=(1.5, 3.0, 1.0), radius=1.5, n_lat=97, n_lon=521):
    """Outward-facing (counter-clockwise from outside) triangles of a UV sphere; the default
    97 x 521 grid has 2*521*96 = 100 032 triangles (BASELINE config C4)."""
    c = np.asarray(center, dtype=np.float64)
    th = np.linspace(0.0, np.pi, n_lat + 1)                  # polar
    ph = np.linspace(0.0, 2.0 * np.pi, n_lon + 1)            # azimuth
    st, ct = np.sin(th), np.cos(th)
    pts = np.stack([np.outer(st, np.cos(ph)), np.tile(ct[:, None], (1, n_lon + 1)), np.outer(st, np.sin(ph))], axis=-1)
    pts = (c + radius * pts).astype(np.float32)              # (n_lat+1, n_lon+1, 3)
    tris = []
    for i in range(n_lat):
        a, b = pts[i, :-1], pts[i, 1:]
        d, e = pts[i + 1, :-1], pts[i + 1, 1:]
        if i > 0:                                            # upper triangle (degenerate at the north pole)
            tris.append(np.stack([a, b, d], axis=1))
        if i < n_lat - 1:                                    # lower triangle (degenerate at the south pole)
            tris.append(np.stack([b, e, d], axis=1))
    t = np.concatenate(tris, axis=0)                         # (T, 3, 3)
    out = np.zeros(len(t), dtype=TRI_DT)
    out["v0"], out["v1"], out["v2"] = t[:, 0], t[:, 1], t[:, 2]
    return out


def triangle_count(n_lat, n_lon):
    return 2 * n_lon * (n_lat - 1)


def add_mesh(geoms, triangles, material_id, existing_triangles=None, existing_meshes=None):
    """Append one PT_TRIANGLE_MESH geom that owns `triangles`; returns (geoms, triangles, meshes)."""
    g = np.zeros(1, dtype=GEOM_DT)
    g["type"] = 2
    g["materialid"] = material_id
    for k in ("transform", "inverseTransform", "invTranspose"):
        g[k][0] = np.eye(4, dtype=np.float32)
    g["scale"] = 1.0
    first = 0 if existing_triangles is None else len(existing_triangles)
    tris = triangles if existing_triangles is None else np.concatenate([existing_triangles, triangles])
    m = np.zeros(1, dtype=MESH_DT)
    m["geom_index"], m["first_triangle"], m["triangle_count"] = len(geoms), first, len(triangles)
    meshes = m if existing_meshes is None else np.concatenate([existing_meshes, m])
    return np.concatenate([np.asarray(geoms, dtype=GEOM_DT), g]), tris, meshes
