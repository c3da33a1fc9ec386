"""World-size-2 gloo tests (CPU) of the N>1 host logic: slab sharding tiles the query range exactly and
the sharded decode + gather reproduces the single-process result for any decode function."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from slice3d_amd.parallel import all_reduce_mean_, decode_points_sharded, object_indices, shard_range


def test_shard_range_tiles_exactly():
    for n in (0, 1, 5, 16, 100000, 256 ** 3):
        for world in (1, 2, 3, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
    assert sorted(sum((object_indices(10, r, 4) for r in range(4)), [])) == list(range(10))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_qry, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    qry = torch.rand(1, n_qry, 3) - 0.5
    w = torch.tensor([0.3, -1.1, 2.0])

    def fake_decode(q):  # any per-query function: queries are independent given the latent
        return torch.sin(q @ w) + q[..., 0] * q[..., 2]

    full = decode_points_sharded(fake_decode, qry)
    ok = torch.equal(full, fake_decode(qry))
    t = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        ret.put(float(t))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_qry", [1, 7, 1000])
def test_query_parallel_decode_gloo_world2(n_qry):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_qry, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=10) == 1.0


def _grad_worker(rank, world, port, ret):
    """Rank r holds the REAL reference's gradients of shard r (tests/golden/g6_ddp_*: one-sample shards of a two-sample
    batch, train mode, dropout 0) in a flat buffer laid out like HipTrainer's; the exchange step (all_reduce_mean_,
    bucket by bucket as HipTrainer.all_reduce_grads does) must produce the golden mean-of-shards gradient, and the
    oracle's Adam restatement the parameters torch.optim.Adam gives the reference (train.py:136)."""
    import numpy as np
    from helpers import GOLDEN
    from oracle.ref_cpu import adam_step
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = np.load(os.path.join(GOLDEN, "g6_ddp_shards_s32_n12_q160_b2.npz"))
    names = [str(k) for k in z["full_names"]]
    sizes = [z["p0:" + k].size for k in names]
    flat = torch.cat([torch.from_numpy(z["g%d:%s" % (rank, k)]).reshape(-1) for k in names])
    want = torch.cat([torch.from_numpy((z["g0:" + k] + z["g1:" + k]) / 2).reshape(-1) for k in names])
    # three uneven buckets, reduced in the order the backward would finish them (last one first)
    cuts = [0, sum(sizes[:5]), sum(sizes[:14]), sum(sizes)]
    for lo, hi in reversed(list(zip(cuts[:-1], cuts[1:]))):
        all_reduce_mean_(flat[lo:hi])
    ok = torch.allclose(flat, want, rtol=0, atol=1e-9)
    off = 0
    for k, n in zip(names, sizes):
        p0 = torch.from_numpy(z["p0:" + k]).reshape(-1)
        p1, _, _ = adam_step(p0, flat[off:off + n], torch.zeros(n), torch.zeros(n), 1)
        ok = ok and bool((p1 - torch.from_numpy(z["p1:" + k]).reshape(-1)).abs().max() < 1e-7)
        gathered = [torch.empty_like(p1) for _ in range(world)]
        dist.all_gather(gathered, p1)
        ok = ok and all(torch.equal(gathered[0], t) for t in gathered)      # replicas stay bit-identical
        off += n
    t = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        ret.put(float(t))
    dist.destroy_process_group()


def test_gradient_all_reduce_gloo_world2():
    """C3 exchange step on CPU with the reference's own per-shard gradients as payload (SURVEY.md 8(e) parity oracle;
    the HIP step is tied to the same goldens by tests/test_gpu_train.py::test_shard_gradients_match_ddp_golden)."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=10) == 1.0


def test_gradient_buckets_partition_the_flat_buffer():
    """HipTrainer's four all-reduce buckets (finish order of the backward) are contiguous, disjoint and cover every
    trainable parameter of the reference's state_dict order; sizes as DESIGN.md section 6 quotes them."""
    from slice3d_amd.models import Slices3DRegModel
    from slice3d_amd.trainer import HipTrainer, bucket_ranges
    m = Slices3DRegModel(n_slices=12, backend="none")
    named = [(k, p.numel()) for k, p in m.named_parameters() if HipTrainer._trainable(k)]
    r = bucket_ranges([k for k, _ in named], [n for _, n in named])
    assert len(r) == 4
    assert sorted(r) == [(r[3][0], r[3][1]), (r[2][0], r[2][1]), (r[1][0], r[1][1]), (r[0][0], r[0][1])]
    assert r[3][0] == 0 and r[0][1] == sum(n for _, n in named) == 20182116
    assert r[3][1] == r[2][0] and r[2][1] == r[1][0] and r[1][1] == r[0][0]
    off = {}
    o = 0
    for k, n in named:
        off[k] = o
        o += n
    inside = lambda k, b: r[b][0] <= off[k] < r[b][1]
    assert inside("fc_out.0.weight", 0) and inside("att_decoder.layers.1.linear1.weight", 0)
    assert inside("slices_generator.up3.conv.double_conv.0.weight", 1) and inside("slices_generator.emds.weight", 1)
    assert inside("slices_generator.down4.24.weight", 2) and inside("slices_generator.down5.40.bias", 2)
    assert inside("slices_generator.down1.0.weight", 3) and inside("slices_generator.down4.21.weight", 3) and inside("slices_generator.down3.20.weight", 3)


class _FakeModel:
    """Stands in for Slices3DRegModel in Generator3D's sharding logic: analytic 'logits' of the grid index."""
    mode = "test"

    def encode(self, data):
        return "code"

    @staticmethod
    def _f(i):
        return torch.sin(i.double() * 0.37).float() + (i % 7).float()

    def decode_grid(self, code, nx, box=1.0, trans_mat_wo_rot_tp=None, q_range=None):
        lo, hi = (0, nx ** 3) if q_range is None else q_range
        out = self._f(torch.arange(lo, hi))
        return out.view(nx, nx, nx) if q_range is None else out


def _gen_worker(rank, world, port, nx, ret):
    from slice3d_amd.generator import Generator3D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gen = Generator3D(_FakeModel(), resolution0=nx, upsampling_steps=0, pred_type="sdf")
    grid = gen.decode_dense_grid("code", nx, 1.0, None)
    ok = torch.equal(grid, _FakeModel().decode_grid("code", nx))
    t = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        ret.put(float(t))
    dist.destroy_process_group()


@pytest.mark.parametrize("nx", [5, 16])
def test_generator3d_dense_grid_is_sharded_over_ranks_gloo_world2(nx):
    """Generator3D.decode_dense_grid under torch.distributed: each rank decodes its slab of the grid's linear index
    (model.decode_grid(q_range=...)), one all_gather, every rank ends with the full grid."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gen_worker, args=(r, 2, port, nx, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=10) == 1.0
