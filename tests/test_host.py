"""CPU tests of the host side: state_dict key parity with the reference, the C-ABI library exports,
loud failure without a GPU / library, synthetic feed contract."""
import ctypes
import json
import os
import re

import pytest
import torch

from helpers import GOLDEN
from slice3d_amd import _lib
from slice3d_amd.models import Slices3DRegModel
from slice3d_amd.synth import make_feed_dict
from slice3d_amd.weights import load_seeded, seeded_array

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_keys_match_reference():
    want = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    m = Slices3DRegModel(n_slices=12, backend="none")
    got = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert list(got.keys()) == list(want.keys())          # same keys, same order (244 tensors)
    assert got == want
    assert len(got) == 244


def test_strict_load_of_reference_format_checkpoint():
    want = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    ckpt = {"model": {k: torch.from_numpy(seeded_array(k, shp, 1)) for k, shp in want.items()}}
    m = Slices3DRegModel(n_slices=12, backend="none")
    m.load_state_dict(ckpt["model"], strict=True)         # reconstruct.py:342-343


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "slice3d_hip.h")).read()
    declared = set(re.findall(r"\b(s3d_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    try:
        lib = ctypes.CDLL(_lib.LIB_PATH)
    except OSError as e:
        pytest.skip("HIP runtime not loadable here: %s" % e)
    for name in declared:
        assert hasattr(lib, name), name
    lib.s3d_version.restype = ctypes.c_int
    assert lib.s3d_version() >= 100
    lib.s3d_head_packed_bytes.restype = ctypes.c_size_t
    assert lib.s3d_head_packed_bytes() > 4 * 3 * (2 * 128 * 2048)


def test_mesh_library_exports_every_declared_symbol():
    """include/slice3d_mesh.h <-> slice3d_amd/mesh.py's binding table <-> libslice3d_mesh.so (host C++, loads anywhere)."""
    from slice3d_amd import mesh
    hdr = open(os.path.join(ROOT, "include", "slice3d_mesh.h")).read()
    declared = set(re.findall(r"\b(s3d_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(mesh._SIG), declared ^ set(mesh._SIG)
    lib = mesh.load()
    for name in declared:
        assert hasattr(lib, name), name


def test_no_cpu_fallback():
    m = Slices3DRegModel(n_slices=12, backend="none").eval()
    fd = make_feed_dict(1, 32, 10, 12)
    with pytest.raises(_lib.S3dError):
        m(fd)
    if not torch.cuda.is_available() and os.path.isfile(_lib.LIB_PATH):
        try:
            m2 = Slices3DRegModel(n_slices=12).eval()
        except OSError:
            return
        with pytest.raises(_lib.S3dError):          # parameters on CPU -> refuses, no silent torch path
            m2(fd)


def test_feed_dict_contract():
    fd = make_feed_dict(2, 32, 100, 12, seed=1)
    assert fd["img_input"].shape == (2, 3, 32, 32) and fd["img_slices"].shape == (2, 36, 32, 32)
    assert fd["qry_norot"].shape == (2, 100, 3) and fd["qry_norot"].abs().max() <= 0.5
    assert fd["obj_rot_mat"].shape == (2, 3, 3) and fd["trans_mat_wo_rot_tp"].shape == (2, 4, 3)
    r = fd["obj_rot_mat"][0].double()
    assert torch.allclose(r @ r.t(), torch.eye(3, dtype=torch.float64), atol=1e-6)
    assert fd["img_input"].abs().max() <= 1.0


def test_seeded_weights_are_deterministic():
    a = seeded_array("fc_s.weight", (128, 992), 0)
    b = seeded_array("fc_s.weight", (128, 992), 0)
    assert (a == b).all() and (a != seeded_array("fc_s.weight", (128, 992), 1)).any()
    m = load_seeded(Slices3DRegModel(n_slices=12, backend="none"))
    assert float(m.vggptlossfunc.mean.flatten()[0]) == pytest.approx(0.485)


def test_bench_self_launch_builds_a_torchrun_job(monkeypatch):
    """`python bench.py --gpus N` without a launcher re-executes itself as an N-rank torch.distributed.run job on
    127.0.0.1 with a free port and forwards every argument (the driver's single-command form); with WORLD_SIZE set it
    does not."""
    import importlib.util
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-6:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_generator3d_refuses_the_reference_dead_post_processing_branches():
    """reconstruct.py:205-240: with_normals / refinement_step / simplify_nfaces run through model.decode(p, c).logits under
    autograd w.r.t. the points — dead in the reference (its model has no decode()).  They must not be accepted and ignored."""
    from slice3d_amd.generator import Generator3D
    for kw in (dict(with_normals=True), dict(refinement_step=30), dict(simplify_nfaces=5000)):
        with pytest.raises(NotImplementedError):
            Generator3D(object(), pred_type="sdf", **kw)
    with pytest.raises(ValueError):
        Generator3D(object(), pred_type="occ")
    Generator3D(object(), pred_type="sdf")
