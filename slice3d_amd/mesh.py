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


# --------------------------------------------------------------------------------------------------
# device versions (libslice3d_hip.so, csrc/mesh.hip): points, values, the dense grid and the mesh stay in HBM
# --------------------------------------------------------------------------------------------------
class DeviceMISE:
    """MISE on the GPU (s3d_mise_dev_*): same constructor; `query()` returns the next round's points as a device
    int32 tensor of linear grid indices (ascending — the reference's SET per round, not its insertion order),
    `points(idx, box)` their float32 coordinates exactly as reconstruct.py:160-161 computes them, `update(idx, values)`
    takes the logits as a device tensor, `to_dense()` returns the (r,r,r) float64 device grid."""

    def __init__(self, resolution_0, depth, threshold, device="cuda"):
        import torch
        from . import _lib
        self._torch, self._L = torch, _lib
        self._lib = _lib.load()
        nb = self._lib.s3d_mise_dev_workspace_bytes(int(resolution_0), int(depth))
        if nb == 0:
            raise ValueError("bad MISE parameters")
        self.device = torch.device(device)
        self._ws = torch.empty(nb, dtype=torch.uint8, device=self.device)
        self._h = self._lib.s3d_mise_dev_create(self._ws.data_ptr(), nb, int(resolution_0), int(depth),
                                                float(threshold), self._stream())
        if not self._h:
            raise _lib.S3dError("s3d_mise_dev_create: %s" % self._lib.s3d_last_error().decode())
        self.resolution_0, self.depth, self.threshold = resolution_0, depth, threshold
        self.resolution = self._lib.s3d_mise_dev_resolution(self._h)
        self._idx = torch.empty(max((resolution_0 + 1) ** 3, 1 << 16), dtype=torch.int32, device=self.device)

    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.s3d_mise_dev_destroy(self._h)
            self._h = None

    def query(self):
        n = C.c_long(0)
        self._L.check(self._lib.s3d_mise_dev_query(self._h, self._idx.data_ptr(), self._idx.numel(), C.byref(n),
                                                   self._stream()), "s3d_mise_dev_query")
        if n.value > self._idx.numel():       # grow and repeat (the count came back with a full buffer)
            self._idx = self._torch.empty(int(n.value * 1.5), dtype=self._torch.int32, device=self.device)
            return self.query()
        return self._idx[:n.value]

    def points(self, idx, box=1.0):
        out = self._torch.empty((idx.numel(), 3), dtype=self._torch.float32, device=self.device)
        self._L.check(self._lib.s3d_mise_dev_points(self._h, idx.data_ptr(), idx.numel(), float(box), out.data_ptr(),
                                                    self._stream()), "s3d_mise_dev_points")
        return out

    def coords(self, idx):
        """(n,3) int64 grid coordinates of linear indices (host-side bookkeeping / tests)."""
        r = self.resolution + 1
        i = idx.long()
        return self._torch.stack([i // (r * r), (i // r) % r, i % r], 1)

    def update(self, idx, values):
        idx = idx.to(device=self.device, dtype=self._torch.int32).contiguous()
        values = values.to(self.device).contiguous().reshape(-1)
        assert values.numel() == idx.numel()
        fn = self._lib.s3d_mise_dev_update_f64 if values.dtype == self._torch.float64 else self._lib.s3d_mise_dev_update
        if values.dtype not in (self._torch.float32, self._torch.float64):
            values = values.float()
        rc = fn(self._h, idx.data_ptr(), values.data_ptr(), idx.numel(), self._stream())
        if rc != 0:
            raise ValueError("Point not in grid! (%s)" % self._lib.s3d_last_error().decode())

    def to_dense(self):
        r = self.resolution + 1
        out = self._torch.empty((r, r, r), dtype=self._torch.float64, device=self.device)
        self._L.check(self._lib.s3d_mise_dev_to_dense(self._h, out.data_ptr(), self._stream()), "s3d_mise_dev_to_dense")
        return out


def marching_cubes_device(volume, isovalue, pad_value=None):
    """Device marching cubes (s3d_mc_dev_*): `volume` a CUDA tensor (nx,ny,nz) float32 / float64; pad_value != None
    behaves like np.pad(volume, 1, constant_values=pad_value) without building it.  -> (vertices (V,3) float64,
    triangles (F,3) int64) as device tensors, bit-identical to libmcubes' output and numbering."""
    import torch
    from . import _lib
    lib = _lib.load()
    assert volume.is_cuda and volume.dim() == 3
    vol = volume.contiguous()
    if vol.dtype not in (torch.float32, torch.float64):
        vol = vol.float()
    is64 = 1 if vol.dtype == torch.float64 else 0
    nx, ny, nz = vol.shape
    pad = 0 if pad_value is None else 1
    pv = 0.0 if pad_value is None else float(pad_value)
    nb = lib.s3d_mc_dev_workspace_bytes(nx, ny, nz, pad)
    ws = torch.empty(nb, dtype=torch.uint8, device=vol.device)
    st = C.c_void_p(torch.cuda.current_stream(vol.device).cuda_stream)
    nv, nt = C.c_long(0), C.c_long(0)
    args = (vol.data_ptr(), is64, nx, ny, nz, pad, pv, float(isovalue), ws.data_ptr(), nb)
    _lib.check(lib.s3d_mc_dev_count(*args, C.byref(nv), C.byref(nt), st), "s3d_mc_dev_count")
    verts = torch.empty((nv.value, 3), dtype=torch.float64, device=vol.device)
    tris = torch.empty((nt.value, 3), dtype=torch.int64, device=vol.device)
    if nv.value or nt.value:
        _lib.check(lib.s3d_mc_dev_emit(*args, verts.data_ptr(), tris.data_ptr(), st), "s3d_mc_dev_emit")
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
