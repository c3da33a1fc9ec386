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
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # per-rank "gradients" of a flat bucket: the all-reduced bucket must equal the mean over shards
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(10007, generator=g)
    want = sum(torch.randn(10007, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)) / world
    all_reduce_mean_(flat)
    ok = torch.allclose(flat, want, atol=1e-6)
    # identical parameters after an identical Adam update on every rank (oracle restatement of Adam)
    from oracle.ref_cpu import adam_step
    p0 = torch.linspace(-1, 1, 10007)
    p1, m1, v1 = adam_step(p0, flat, torch.zeros_like(p0), torch.zeros_like(p0), 1)
    gathered = [torch.empty_like(p1) for _ in range(world)]
    dist.all_gather(gathered, p1)
    ok = ok and all(torch.equal(gathered[0], t) for t in gathered)
    t = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        ret.put(float(t))
    dist.destroy_process_group()


def test_gradient_all_reduce_gloo_world2():
    """C3 exchange step on CPU: mean of per-shard gradients, replicas stay bit-identical after Adam."""
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
