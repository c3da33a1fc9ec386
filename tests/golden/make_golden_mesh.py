"""Generates tests/golden/mesh_*.npz from the REFERENCE's native mesh utilities, compiled from the
sources under /root/reference by `make -C oracle` into oracle/_ref/ (authoring container only):
libmise (Cython) and libmcubes' marching_cubes template.  Run from the repo root."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref")
sys.path.insert(0, REF)
OUT = os.path.dirname(os.path.abspath(__file__))


def ref_marching_cubes(vol, iso):
    lib = C.CDLL(os.path.join(REF, "libmcref.so"))
    lib.mcref_run.restype = C.c_long
    vol = np.ascontiguousarray(vol, dtype=np.float64)
    nt = C.c_long()
    nv = lib.mcref_run(C.c_void_p(vol.ctypes.data), vol.shape[0], vol.shape[1], vol.shape[2], C.c_double(iso), C.byref(nt))
    v = np.empty(nv, dtype=np.float64)
    p = np.empty(nt.value, dtype=np.int64)
    lib.mcref_copy(C.c_void_p(v.ctypes.data), C.c_void_p(p.ctypes.data))
    return v.reshape(-1, 3), p.reshape(-1, 3)


def field(n, seed):
    rng = np.random.default_rng(seed)
    g = np.stack(np.meshgrid(*[np.linspace(-1, 1, n)] * 3, indexing="ij"), -1)
    f = 0.6 - np.linalg.norm(g - rng.uniform(-0.2, 0.2, 3), axis=-1)
    for _ in range(4):
        c = rng.uniform(-0.7, 0.7, 3)
        f = np.maximum(f, rng.uniform(0.15, 0.35) - np.linalg.norm(g - c, axis=-1))
    return f + 0.01 * rng.standard_normal(f.shape)


def main():
    from mise import MISE   # the compiled reference extension
    rec = {}
    # marching cubes: padded noisy blobs (incl. ties f == iso on a plateau) at two sizes
    for name, n, seed in (("a", 14, 1), ("b", 23, 2)):
        vol = np.pad(field(n, seed), 1, "constant", constant_values=-1e6)
        vol[3:5, 3:5, 3:5] = 0.0          # exact ties with the iso value
        v, t = ref_marching_cubes(vol, 0.0)
        rec["mc_%s_vol" % name], rec["mc_%s_v" % name], rec["mc_%s_t" % name] = vol, v, t
    # sphere counts of SURVEY.md 8(c): MISE(64,2,0) on a radius-0.3 sphere, then marching cubes
    def sphere(p, res):
        return 0.3 - np.linalg.norm(p / res - 0.5, axis=-1)
    m = MISE(64, 2, 0.0)
    rounds, nq = 0, 0
    pts = m.query()
    while pts.shape[0]:
        m.update(pts, sphere(pts.astype(np.float64), m.resolution))
        nq += pts.shape[0]
        rounds += 1
        pts = m.query()
    dense = m.to_dense()
    v, t = ref_marching_cubes(np.pad(dense, 1, "constant", constant_values=-1e6), 0.0)
    rec["sphere_counts"] = np.array([rounds, nq, dense.shape[0], v.shape[0], t.shape[0]], dtype=np.int64)
    # MISE traces: every round's query points and the final dense grid, small configs
    # t4 / t5: isolated positives and pure noise — hanging points of refined neighbours keep flagging coarse leaves, so
    # the refinement cascades over many rounds (the case an incremental update() must reproduce query for query)
    for name, (r0, d, thr, seed) in {"t1": (1, 2, 0.0, 0), "t2": (4, 2, 0.1, 3), "t3": (6, 1, -0.05, 4),
                                     "t4": (6, 3, 0.0, 5), "t5": (3, 3, 0.0, 6)}.items():
        rng = np.random.default_rng(seed)
        c, rad = rng.uniform(0.35, 0.65, 3), rng.uniform(0.2, 0.35)
        m = MISE(r0, d, thr)
        pts, k = m.query(), 0
        while pts.shape[0]:
            rec["mise_%s_q%d" % (name, k)] = pts
            vals = rad - np.linalg.norm(pts / m.resolution - c, axis=-1)
            if name == "t1":
                vals = pts[:, 0].astype(np.float64) / m.resolution - 0.45     # libmise/test.py style half-space
            elif name == "t4":
                vals = np.where(rng.random(len(pts)) < 0.02, 1.0, -1.0) * (0.1 + rng.random(len(pts)))
            elif name == "t5":
                vals = np.round(rng.standard_normal(len(pts)), 1)             # also exact-threshold hits
            rec["mise_%s_v%d" % (name, k)] = vals
            m.update(pts, vals)
            pts, k = m.query(), k + 1
        rec["mise_%s_dense" % name] = m.to_dense()
        rec["mise_%s_cfg" % name] = np.array([r0, d, thr, k], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "mesh_reference.npz"), **rec)
    print("sphere counts (rounds, queries, dense n, verts, faces):", rec["sphere_counts"])
    print("size %.1f KB" % (os.path.getsize(os.path.join(OUT, "mesh_reference.npz")) / 1024))


if __name__ == "__main__":
    main()
