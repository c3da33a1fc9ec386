"""CPU tests of the native MISE / marching-cubes library (slice3d_amd/csrc_mesh)
against golden data produced by the REFERENCE's own compiled libmise / libmcubes
(tests/golden/make_golden_mesh.py) and, when oracle/_ref is built here, against them live."""
import os
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mesh():
    from slice3d_amd import mesh as m
    if not os.path.isfile(m.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "slice3d_amd", "csrc_mesh")], check=True)
    return m


@pytest.fixture(scope="module")
def gold():
    z = np.load(os.path.join(GOLDEN, "mesh_reference.npz"))
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", ["a", "b"])
def test_marching_cubes_bit_exact(mesh, gold, name):
    v, t = mesh.marching_cubes(gold["mc_%s_vol" % name], 0.0)
    assert t.shape == gold["mc_%s_t" % name].shape and v.shape == gold["mc_%s_v" % name].shape
    assert np.array_equal(t, gold["mc_%s_t" % name])          # same faces, same vertex numbering
    assert np.array_equal(v, gold["mc_%s_v" % name])          # bit-exact coordinates


@pytest.mark.parametrize("name", ["t1", "t2", "t3", "t4", "t5"])
def test_mise_trace_matches_reference(mesh, gold, name):
    r0, d, thr, rounds = gold["mise_%s_cfg" % name]
    m = mesh.MISE(int(r0), int(d), float(thr))
    k = 0
    pts = m.query()
    while pts.shape[0]:
        assert np.array_equal(pts, gold["mise_%s_q%d" % (name, k)])      # same points, same order
        m.update(pts, gold["mise_%s_v%d" % (name, k)])
        pts, k = m.query(), k + 1
    assert k == int(rounds)
    assert np.array_equal(m.to_dense(), gold["mise_%s_dense" % name])


def test_libmise_test_py_scenario(mesh):
    """reference libmise/test.py: MISE(1,2,0.) on a half-space; dense 5^3 grid sums to 105."""
    m = mesh.MISE(1, 2, 0.0)
    pts = m.query()
    rounds = 0
    while pts.shape[0]:
        m.update(pts, np.where(pts[:, 0] >= 2, 1.0, -1.0))
        pts = m.query()
        rounds += 1
    dense = m.to_dense()
    assert dense.shape == (5, 5, 5) and rounds >= 2 and np.isfinite(dense).all()


def test_sphere_counts(mesh, gold):
    """MISE(64,2,0) on a radius-0.3 sphere -> 3 rounds, 622101 queries, 257^3 grid,
    marching cubes on the padded grid -> 111078 vertices / 222152 faces."""
    m = mesh.MISE(64, 2, 0.0)
    rounds = nq = 0
    pts = m.query()
    while pts.shape[0]:
        m.update(pts, 0.3 - np.linalg.norm(pts / m.resolution - 0.5, axis=-1))
        nq += pts.shape[0]
        rounds += 1
        pts = m.query()
    dense = m.to_dense()
    v, t = mesh.marching_cubes(np.pad(dense, 1, "constant", constant_values=-1e6), 0.0)
    assert [rounds, nq, dense.shape[0], v.shape[0], t.shape[0]] == list(gold["sphere_counts"])
    assert list(gold["sphere_counts"]) == [3, 622101, 257, 111078, 222152]
    # watertight: every edge is shared by exactly two triangles
    e = np.sort(np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]]), axis=1)
    _, counts = np.unique(e, axis=0, return_counts=True)
    assert (counts == 2).all()


@pytest.mark.parametrize("kind", ["noise", "sparse", "ties"])
def test_mise_incremental_flags_equal_full_recompute(mesh, kind):
    """The refinement flags are kept incrementally; re-valuing an already known point makes update() rebuild them the
    reference's way (all known points -> all leaves, mise.pyx:182-232).  Both must give the same queries, in the same
    order, through a cascade of rounds — also when the caller answers only part of a query."""
    def field(pts, res, rng):
        p = pts.astype(np.float64) / res - 0.5
        if kind == "noise":
            return rng.standard_normal(len(pts))
        if kind == "sparse":      # isolated positives: hanging points keep flagging coarse neighbours, many rounds
            return np.where(rng.random(len(pts)) < 0.02, 1.0, -1.0) * (0.1 + rng.random(len(pts)))
        return np.round(np.sin(7 * p[:, 0]) * np.cos(5 * p[:, 1]) + p[:, 2], 1)    # many values exactly at threshold
    for r0, depth in ((6, 3), (9, 2)):
        a, b = mesh.MISE(r0, depth, 0.0), mesh.MISE(r0, depth, 0.0)
        rng = np.random.default_rng(r0)
        first = None
        rounds = 0
        while True:
            qa, qb = a.query(), b.query()
            assert np.array_equal(qa, qb)
            if not len(qa):
                break
            vals = field(qa, a.resolution, rng)
            keep = rng.random(len(qa)) < 0.8
            keep[0] = True
            pts, vals = qa[keep], vals[keep]
            if first is None:
                first = (pts[:1].copy(), vals[:1].copy())
            a.update(pts, vals)
            # b: the same update plus one old point with its old value again -> full rebuild path
            b.update(np.concatenate([pts, first[0]]), np.concatenate([vals, first[1]]))
            rounds += 1
            assert rounds < 500
        assert rounds > depth
        assert np.array_equal(a.to_dense(), b.to_dense())


def test_update_rejects_foreign_points(mesh):
    m = mesh.MISE(2, 1, 0.0)
    with pytest.raises(ValueError):
        m.update(np.array([[1, 1, 1]]), np.array([0.0]))     # (1,1,1) is not a grid point before refinement


def test_obj_export(mesh, tmp_path):
    vol = np.pad(np.ones((2, 2, 2)), 1, constant_values=-1.0)
    v, t = mesh.marching_cubes(vol, 0.0)
    path = mesh.Mesh(v, t).export(str(tmp_path / "m.obj"))
    lines = open(path).read().split("\n")
    assert sum(l.startswith("v ") for l in lines) == len(v) and sum(l.startswith("f ") for l in lines) == len(t)
