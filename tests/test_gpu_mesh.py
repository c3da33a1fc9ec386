"""GPU tests of the device-side MISE / marching cubes (csrc/mesh.hip) through the C ABI, against
(1) the goldens produced by the REFERENCE's own compiled libmise / libmcubes (tests/golden/mesh_reference.npz,
    tests/golden/make_golden_mesh.py) and (2) this package's host C++ twin (bit-exact against the same reference
    libraries in tests/test_mesh.py) on larger and nastier fields.
Bar: bit-exact vertices / faces / dense grids; every MISE round queries the reference's SET of points (the device
returns them in ascending grid index, the reference in insertion order)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold():
    z = np.load(os.path.join(GOLDEN, "mesh_reference.npz"))
    return {k: z[k] for k in z.files}


def lin(points, r):
    p = np.asarray(points, dtype=np.int64)
    return (p[:, 0] * r + p[:, 1]) * r + p[:, 2]


@pytest.mark.parametrize("name", ["a", "b"])
def test_device_marching_cubes_bit_exact_vs_reference(gold, name):
    from slice3d_amd.mesh import marching_cubes_device
    vol = torch.from_numpy(gold["mc_%s_vol" % name]).cuda()
    v, t = marching_cubes_device(vol, 0.0)
    assert np.array_equal(t.cpu().numpy(), gold["mc_%s_t" % name])      # same faces, same vertex numbering
    assert np.array_equal(v.cpu().numpy(), gold["mc_%s_v" % name])      # bit-exact coordinates


@pytest.mark.parametrize("name", ["t1", "t2", "t3", "t4", "t5"])
def test_device_mise_trace_matches_reference(gold, name):
    from slice3d_amd.mesh import DeviceMISE
    r0, d, thr, rounds = gold["mise_%s_cfg" % name]
    m = DeviceMISE(int(r0), int(d), float(thr))
    r = m.resolution + 1
    k = 0
    idx = m.query()
    while idx.numel():
        want_pts, want_val = gold["mise_%s_q%d" % (name, k)], gold["mise_%s_v%d" % (name, k)]
        order = np.argsort(lin(want_pts, r))
        assert np.array_equal(idx.cpu().numpy().astype(np.int64), lin(want_pts, r)[order])   # the same set of points
        assert np.array_equal(m.coords(idx).cpu().numpy(), want_pts[order])
        m.update(idx, torch.from_numpy(want_val[order]).cuda())
        idx, k = m.query(), k + 1
    assert k == int(rounds)
    assert np.array_equal(m.to_dense().cpu().numpy(), gold["mise_%s_dense" % name])


def test_device_sphere_counts_and_mesh_equal_host_library(gold):
    """MISE(64,2,0) on a radius-0.3 sphere -> 3 rounds, 622101 queries, 257^3 grid; marching cubes on
    the (implicitly) padded grid -> 111078 vertices / 222152 faces; dense grid and mesh bit-identical to the host
    library's (itself bit-exact against the reference's libmise / libmcubes, tests/test_mesh.py)."""
    from slice3d_amd import mesh
    dm, hm = mesh.DeviceMISE(64, 2, 0.0), mesh.MISE(64, 2, 0.0)
    field = lambda pts, res: 0.3 - np.linalg.norm(pts / res - 0.5, axis=-1)
    rounds = nq = 0
    idx = dm.query()
    while idx.numel():
        pts = dm.coords(idx).cpu().numpy()
        hp = hm.query()
        assert np.array_equal(np.sort(lin(hp, dm.resolution + 1)), idx.cpu().numpy())
        dm.update(idx, torch.from_numpy(field(pts, dm.resolution)).cuda())
        hm.update(hp, field(hp, hm.resolution))
        nq, rounds = nq + idx.numel(), rounds + 1
        idx = dm.query()
    assert hm.query().shape[0] == 0
    dense = dm.to_dense()
    assert np.array_equal(dense.cpu().numpy(), hm.to_dense())
    v, t = mesh.marching_cubes_device(dense, 0.0, pad_value=-1e6)
    assert [rounds, nq, dense.shape[0], v.shape[0], t.shape[0]] == list(gold["sphere_counts"]) == \
           [3, 622101, 257, 111078, 222152]
    hv, ht = mesh.marching_cubes(np.pad(hm.to_dense(), 1, "constant", constant_values=-1e6), 0.0)
    assert np.array_equal(t.cpu().numpy(), ht) and np.array_equal(v.cpu().numpy(), hv)
    # float32 storage (the dense-grid path hands the decoder's fp32 logits straight to marching cubes)
    d32 = dense.float()
    v32, t32 = mesh.marching_cubes_device(d32, 0.0, pad_value=-1e6)
    hv32, ht32 = mesh.marching_cubes(np.pad(d32.cpu().numpy(), 1, "constant", constant_values=-1e6), 0.0)
    assert np.array_equal(t32.cpu().numpy(), ht32) and np.array_equal(v32.cpu().numpy(), hv32)


@pytest.mark.parametrize("kind", ["noise", "sparse", "ties"])
@pytest.mark.parametrize("r0,depth", [(6, 3), (9, 2), (5, 4)])
def test_device_mise_cascades_equal_host_library(kind, r0, depth):
    """Fields that keep flagging coarse neighbours through hanging points (many rounds), values exactly at the
    threshold, partial answers are not used here (every round answers all points): per round the same point set,
    at the end the same dense grid, and marching cubes of it (no padding: boundary-owned vertices) bit-identical."""
    from slice3d_amd import mesh

    def field(li, res):
        rng_vals = np.sin(li.astype(np.float64) * 12.9898) * 43758.5453
        u = rng_vals - np.floor(rng_vals)                   # hash of the grid index in [0,1): order-independent
        if kind == "noise":
            return u - 0.5
        if kind == "sparse":
            return np.where(u < 0.02, 1.0, -1.0) * (0.1 + u)
        r = res + 1
        p = np.stack([li // (r * r), (li // r) % r, li % r], 1).astype(np.float64) / res - 0.5
        return np.round(np.sin(7 * p[:, 0]) * np.cos(5 * p[:, 1]) + p[:, 2], 1)      # many exact ties with thr = 0
    dm, hm = mesh.DeviceMISE(r0, depth, 0.0), mesh.MISE(r0, depth, 0.0)
    r = dm.resolution + 1
    rounds = 0
    while True:
        idx, hp = dm.query(), hm.query()
        li = idx.cpu().numpy().astype(np.int64)
        assert np.array_equal(li, np.sort(lin(hp, r))), rounds
        if not len(hp):
            break
        dm.update(idx, torch.from_numpy(field(li, dm.resolution)).cuda())
        hm.update(hp, field(lin(hp, r), hm.resolution))
        rounds += 1
        assert rounds < 500
    assert rounds > depth
    dense = dm.to_dense()
    assert np.array_equal(dense.cpu().numpy(), hm.to_dense())
    v, t = mesh.marching_cubes_device(dense, 0.0)
    hv, ht = mesh.marching_cubes(hm.to_dense(), 0.0)
    assert np.array_equal(t.cpu().numpy(), ht) and np.array_equal(v.cpu().numpy(), hv)


def test_device_update_rejects_foreign_points():
    from slice3d_amd.mesh import DeviceMISE
    m = DeviceMISE(2, 1, 0.0)
    with pytest.raises(ValueError):
        m.update(torch.tensor([(1 * 5 + 1) * 5 + 1], dtype=torch.int32), torch.tensor([0.0]))   # (1,1,1): no grid point yet


def test_generator3d_device_and_host_backends_give_the_same_mesh():
    """reconstruct.py's default options scaled down (MISE 16 -> 64, and a dense 48^3 grid): the device pipeline
    (points, values, grid, classify/scan/emit on the GPU) and the host pipeline (reference point order, float64
    round trips) produce bit-identical value grids and meshes."""
    from slice3d_amd.generator import Generator3D
    from slice3d_amd.models import Slices3DRegModel
    from slice3d_amd.synth import make_feed_dict
    from slice3d_amd.weights import load_seeded
    model = load_seeded(Slices3DRegModel(n_slices=12, mode="test", prec="f16x3"), 0).cuda().eval()
    fd = {k: v.cuda() for k, v in make_feed_dict(1, 64, 16, 12, seed=77, with_slices=False).items()}
    for res0, ups in ((16, 2), (48, 0)):
        kw = dict(threshold=0.5, resolution0=res0, upsampling_steps=ups, pred_type="sdf")
        gd, gh = Generator3D(model, mesh_backend="device", **kw), Generator3D(model, mesh_backend="host", **kw)
        grid_d, grid_h = gd.generate_value_grid(fd), gh.generate_value_grid(fd)
        assert grid_d.shape == grid_h.shape and np.array_equal(grid_d, grid_h)
        (md, sd), (mh, sh) = gd.generate_mesh(fd), gh.generate_mesh(fd)
        assert len(md.faces) > 0
        assert np.array_equal(md.faces, mh.faces) and np.array_equal(md.vertices, mh.vertices)
        assert "time (eval points)" in sd and "time (marching cubes)" in sd
