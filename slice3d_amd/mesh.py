"""MISE and marching cubes — ctypes binding of the native host library
slice3d_amd/csrc_mesh/libslice3d_mesh.so (C++; built by __graft_entry__.build()), with the Python
surface reconstruct.py uses: `MISE(resolution0, depth, threshold).query()/update()/to_dense()`
(reference libmise/mise.pyx) and `marching_cubes(volume, isovalue)` (reference libmcubes)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc_mesh", "libslice3d_mesh.so")
_lib = None

_SIG = {
    "s3d_mise_create": (C.c_void_p, [C.c_int, C.c_int, C.c_double]),
    "s3d_mise_destroy": (None, [C.c_void_p]),
    "s3d_mise_resolution": (C.c_int, [C.c_void_p]),
    "s3d_mise_query_count": (C.c_long, [C.c_void_p]),
    "s3d_mise_query": (None, [C.c_void_p, C.c_void_p]),
    "s3d_mise_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]),
    "s3d_mise_to_dense": (None, [C.c_void_p, C.c_void_p]),
    "s3d_mise_num_points": (C.c_long, [C.c_void_p]),
    "s3d_mc_run": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double]),
    "s3d_mc_num_vertices": (C.c_long, [C.c_void_p]),
    "s3d_mc_num_triangles": (C.c_long, [C.c_void_p]),
    "s3d_mc_copy": (None, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "s3d_mc_free": (None, [C.c_void_p]),
}


def load():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError("libslice3d_mesh.so not built (run __graft_entry__.build())")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIG.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


class MISE:
    def __init__(self, resolution_0, depth, threshold):
        self._lib = load()
        self._h = self._lib.s3d_mise_create(int(resolution_0), int(depth), float(threshold))
        if not self._h:
            raise ValueError("bad MISE parameters")
        self.resolution_0, self.depth, self.threshold = resolution_0, depth, threshold
        self.resolution = self._lib.s3d_mise_resolution(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.s3d_mise_destroy(self._h)
            self._h = None

    def query(self):
        n = self._lib.s3d_mise_query_count(self._h)
        pts = np.zeros((n, 3), dtype=np.int64)
        if n:
            self._lib.s3d_mise_query(self._h, pts.ctypes.data)
        return pts

    def update(self, points, values):
        points = np.ascontiguousarray(points, dtype=np.int64)
        values = np.ascontiguousarray(values, dtype=np.float64)
        assert points.shape[0] == values.shape[0] and points.shape[1] == 3
        if self._lib.s3d_mise_update(self._h, points.ctypes.data, values.ctypes.data, points.shape[0]) != 0:
            raise ValueError("Point not in grid!")

    def to_dense(self):
        r = self.resolution + 1
        out = np.empty((r, r, r), dtype=np.float64)
        self._lib.s3d_mise_to_dense(self._h, out.ctypes.data)
        return out

    def num_points(self):
        return self._lib.s3d_mise_num_points(self._h)


def marching_cubes(volume, isovalue):
    """-> (vertices (V,3) float64 in index units + 0.5, triangles (F,3) int64), libmcubes conventions."""
    lib = load()
    vol = np.ascontiguousarray(volume, dtype=np.float64)
    assert vol.ndim == 3
    h = lib.s3d_mc_run(vol.ctypes.data, vol.shape[0], vol.shape[1], vol.shape[2], float(isovalue))
    try:
        nv, nt = lib.s3d_mc_num_vertices(h), lib.s3d_mc_num_triangles(h)
        verts = np.empty((nv, 3), dtype=np.float64)
        tris = np.empty((nt, 3), dtype=np.int64)
        lib.s3d_mc_copy(h, verts.ctypes.data, tris.ctypes.data)
    finally:
        lib.s3d_mc_free(h)
    return verts, tris


class Mesh:
    """Minimal stand-in for the trimesh.Trimesh the reference returns (trimesh is not installed here):
    vertices, faces and Wavefront .obj export (reconstruct.py:415 `mesh.export(path)`)."""

    def __init__(self, vertices, faces):
        self.vertices = np.asarray(vertices, dtype=np.float64)
        self.faces = np.asarray(faces, dtype=np.int64)

    def export(self, path):
        with open(path, "w") as f:
            for v in self.vertices:
                f.write("v %.8f %.8f %.8f\n" % (v[0], v[1], v[2]))
            for t in self.faces + 1:
                f.write("f %d %d %d\n" % (t[0], t[1], t[2]))
        return path
