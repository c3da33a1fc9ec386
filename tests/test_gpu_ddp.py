"""Data-parallel training and query-parallel reconstruction with TWO PROCESSES (torch.distributed, gloo carrying CUDA
tensors; both ranks share the one GPU of the test box — the code path is the one `torchrun --nproc-per-node N` takes
with RCCL, minus the transport).  Oracles: the REAL reference's gradients —
  * per-rank BatchNorm statistics (torch DDP's default): mean of the reference's per-shard gradients (g6 goldens);
  * --sync_bn: two one-sample ranks with cross-rank statistics ARE the reference's two-sample batch: its full-batch
    gradients, losses and updated running statistics (g5 goldens)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import GOLDEN

pytestmark = pytest.mark.gpu
KEYS = ("img_input", "img_slices", "qry_norot", "sdf", "obj_rot_mat", "trans_mat_wo_rot_tp")
PRE_BN_BIASES = {"slices_generator.%s.bias" % k for k in
                 ("down1.0", "down2.7", "down3.14", "down3.17", "down4.24", "down4.27", "down5.34", "down5.37")}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(fn, *args, world=2):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=fn, args=(r, world, port, ret) + args) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    return ret.get(timeout=10)


def _init(rank, world, port, backend):
    """gloo: every rank on GPU 0 (one-GPU test box); nccl: one rank per GPU over RCCL (tests/test_gpu_multi.py)."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)


def _train_worker(rank, world, port, ret, golden, sync_bn, overlap, prec, backend="gloo"):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import check_grads_against_golden
    from slice3d_amd.models import Slices3DRegModel
    from slice3d_amd.trainer import HipTrainer
    from slice3d_amd.weights import load_seeded
    _init(rank, world, port, backend)
    z = np.load(os.path.join(GOLDEN, golden + ".npz"))
    batch = {k: torch.from_numpy(z[k][rank:rank + 1]).cuda() for k in KEYS}
    m = load_seeded(Slices3DRegModel(n_slices=12, mode="train"), 0).cuda()
    tr = HipTrainer(m, sync_bn=sync_bn, overlap_all_reduce=overlap, prec=prec)
    losses = tr.forward_backward(batch).clone()
    tr.all_reduce_grads()
    dist.all_reduce(losses)
    losses /= world
    torch.cuda.synchronize()
    msg = "ok"
    try:
        named = dict(m.named_parameters())
        grads = {k: named[k].grad.reshape(-1).cpu().numpy() for k in tr.names}
        worst = check_grads_against_golden(z, grads, skip=PRE_BN_BIASES)
        if sync_bn:   # the reference's two-sample batch
            want = z["losses"]
            got = losses.cpu().numpy()
            for i in range(3):
                assert abs(got[i] - want[i]) < 2e-5 * abs(want[i]) + 1e-7, (i, got[i], want[i])
            for key in z.files:
                if key.startswith("bn:") and ".down5_." not in key:
                    assert np.abs(m.state_dict()[key[3:]].cpu().numpy() - z[key]).max() < 1e-5, key
        # replicas hold identical gradients after the exchange
        other = tr.grad_flat.clone()
        dist.broadcast(other, 0)
        assert torch.equal(other, tr.grad_flat)
        msg = "ok worst %.2e" % worst
    except AssertionError as e:
        msg = "rank %d: %r" % (rank, e)
    gathered = [None] * world
    dist.all_gather_object(gathered, msg)
    if rank == 0:
        ret.put(gathered)
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_ddp_world2_per_rank_bn_matches_mean_of_reference_shard_gradients(overlap):
    """train.py under two ranks (one sample each), BatchNorm statistics per rank: the exchanged gradient is the mean
    of the reference's per-shard gradients; with `overlap` the buckets are reduced on a side stream behind the
    hipEvents the backward records as it finishes them."""
    msgs = _spawn(_train_worker, "g6_ddp_shards_s32_n12_q160_b2", False, overlap, "f32")
    assert all(m.startswith("ok") for m in msgs), msgs


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_ddp_world2_sync_bn_reproduces_the_reference_full_batch(prec):
    msgs = _spawn(_train_worker, "g5_train_s32_n12_q128_b2", True, True, prec)
    assert all(m.startswith("ok") for m in msgs), msgs


def _recon_worker(rank, world, port, ret, backend="gloo", cases=((64, 12, 2), (64, 40, 0))):
    from slice3d_amd.generator import Generator3D
    from slice3d_amd.models import Slices3DRegModel
    from slice3d_amd.synth import make_feed_dict
    from slice3d_amd.weights import load_seeded
    _init(rank, world, port, backend)
    ok = True
    for size, res0, ups in cases:
        model = load_seeded(Slices3DRegModel(img_size=size, n_slices=12, mode="test", prec="f16x3"), 0).cuda().eval()
        fd = {k: v.cuda() for k, v in make_feed_dict(1, size, 16, 12, seed=77, with_slices=False).items()}
        kw = dict(resolution0=res0, upsampling_steps=ups, pred_type="sdf")
        sharded = Generator3D(model, **kw).generate_value_grid(fd)                       # slabs + all_gather
        alone = Generator3D(model, shard_queries=False, **kw).generate_value_grid(fd)    # this rank does it all
        ok = ok and np.array_equal(sharded, alone)
    t = torch.tensor([1.0 if ok else 0.0], device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        ret.put(float(t))
    dist.destroy_process_group()


def test_query_parallel_reconstruction_world2_equals_single_rank():
    """C4 split: every rank encodes the object, decodes its slab of the dense grid / of each MISE
    round, one all_gather per grid / round — bit-identical to one rank doing everything."""
    assert _spawn(_recon_worker) == 1.0
