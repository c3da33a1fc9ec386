"""train.py — host entry point kept from the reference (reg_slices/train.py): same flags
(options.get_parser), same loop structure (train_step / val_step / checkpoint naming / lr decay), the
compute swapped for the HIP path (slice3d_amd.trainer.HipTrainer = zero_grad + forward + cal_loss_pred +
backward + Adam of train.py:41-53).  Launch one process per GPU with torch.distributed.run for data
parallel training (the reference's nn.DataParallel path is broken for this model).

    python reg_slices/train.py --name_exp demo --name_dataset synthetic --img_size 64 --n_qry 1000 --n_bs 1
"""
import glob
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from options import get_parser  # noqa: E402
from slice3d_amd.models import Slices3DRegModel  # noqa: E402
from slice3d_amd.synth import SyntheticSlice3DDataset, collate  # noqa: E402
from slice3d_amd.trainer import HipTrainer  # noqa: E402


def cal_acc(x, gt, pred_type="sdf"):
    """train.py:21-27 (sdf branch; the model never emits occ_pred)."""
    acc = ((x["sdf_pred"] >= 0) == (gt["sdf"] >= 0)).float().sum(dim=-1) / x["sdf_pred"].shape[1]
    return acc.mean(-1)


def train_step(batch, trainer, args=None):
    """train.py:41-53 -> (loss_pred, loss_img, loss_img_vgg, acc) as python floats."""
    return trainer.train_step(batch)


@torch.no_grad()
def val_step(model, val_loader, pred_type="sdf"):
    """train.py:71-93: eval-mode forward, mean L1(sdf) and sign accuracy over batches, last image loss."""
    tot_l, tot_a, n, loss_img = 0.0, 0.0, 0, 0.0
    for batch in val_loader:
        batch = {k: v.cuda() for k, v in batch.items()}
        x = model(batch)
        tot_l += float((x["sdf_pred"] - batch["sdf"]).abs().mean())
        tot_a += float(cal_acc(x, batch))
        loss_img = float((x["slices_rec"] - batch["img_slices"]).abs().mean())
        n += 1
    n = max(n, 1)
    return tot_l / n, tot_a / n, loss_img


def make_loaders(args, rank, world):
    if args.shards:   # pre-packed uint8 shards, staged on the GPU (slice3d_amd/shards.py)
        from slice3d_amd.shards import ShardLoader
        mk = lambda split: ShardLoader(args.shards, split, args.n_bs, args.n_qry, device="cuda", seed=args.seed,
                                       rank=rank if split == "train" else 0, world=world if split == "train" else 1,
                                       cache_on_device=args.shards_in_hbm)
        return mk("train"), mk("val")
    if args.name_dataset != "synthetic":   # on-disk dataset in the reference's layout (train.py:123-127)
        from slice3d_amd.datasets import Slice3DDataset

        def disk_loader(split):
            ds = Slice3DDataset(split=split, args=args)
            sampler = None
            if world > 1 and split == "train":   # one shard of the samples per rank (the reference uses DataParallel)
                sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True)
            return torch.utils.data.DataLoader(ds, batch_size=args.n_bs, shuffle=(split == "train" and sampler is None),
                                               sampler=sampler, num_workers=args.n_wk, drop_last=True)
        return disk_loader("train"), disk_loader("val")

    def loader(split):
        ds = SyntheticSlice3DDataset(args.synthetic_len, args.img_size, args.n_qry, args.n_slices, split=split,
                                     rank=rank, world=world)
        return torch.utils.data.DataLoader(ds, batch_size=args.n_bs, shuffle=(split == "train"), num_workers=0,
                                           drop_last=len(ds) >= args.n_bs, collate_fn=collate)
    return loader("train"), loader("val")


def train(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if world > 1:
        torch.distributed.init_process_group("nccl")
    dir_ckpt = os.path.join("experiments", args.name_exp, "ckpt")
    if rank == 0:
        os.makedirs(dir_ckpt, exist_ok=True)
        with open(os.path.join("experiments", args.name_exp, "opts.txt"), "w") as f:
            for key, value in vars(args).items():
                f.write(str(key) + ": " + str(value) + "\n")
    train_loader, val_loader = make_loaders(args, rank, world)
    model = Slices3DRegModel(img_size=args.img_size, n_slices=args.n_slices, mode=args.mode).cuda()
    if world > 1:   # identical initial weights on every rank
        for t in list(model.parameters()) + list(model.buffers()):
            torch.distributed.broadcast(t.data, 0)
    trainer = HipTrainer(model, lr=args.lr, dropout=args.dropout, seed=args.seed * 65537 + rank,   # per-rank dropout streams
                         sync_bn=args.sync_bn, prec=args.prec)
    epoch_latest, n_iter = 0, 0
    if args.resume:
        ckpts = glob.glob(os.path.join(dir_ckpt, "*"))
        ckpt = torch.load(max(ckpts, key=os.path.getctime), map_location="cuda")
        model.load_state_dict(ckpt["model"])
        trainer.load_state_dict(ckpt["opt"])
        epoch_latest, n_iter = ckpt["n_epoch"] + 1, ckpt["n_iter"]
    n_epoch = epoch_latest
    for _ in range(epoch_latest, args.n_epochs):
        model.train()
        sampler = getattr(train_loader, "sampler", train_loader)
        if hasattr(sampler, "set_epoch"):
            sampler.set_epoch(n_epoch)   # a fresh shuffle per epoch (DistributedSampler replays epoch 0 otherwise)
        for batch in train_loader:
            batch = {k: v.cuda() for k, v in batch.items()}
            loss_pred, loss_img, loss_img_vgg, acc = train_step(batch, trainer, args)
            if n_iter % args.freq_log == 0 and rank == 0:
                print("[train] epcho:", n_epoch, " ,iter:", n_iter, " loss_pred:", loss_pred, " loss_img:", loss_img,
                      " loss_img_vgg:", loss_img_vgg, " acc:", acc)
            n_iter += 1
        if n_epoch % args.freq_ckpt == 0 and rank == 0:
            model.eval()
            avg_loss_pred, avg_acc, avg_loss_img = val_step(model, val_loader, args.pred_type)
            print("[val] epcho:", n_epoch, " ,iter:", n_iter, " avg_loss_pred:", avg_loss_pred, " acc:", avg_acc)
            torch.save({"model": model.state_dict(), "opt": trainer.state_dict(), "n_epoch": n_epoch, "n_iter": n_iter},
                       f"{dir_ckpt}/{n_epoch}_{n_iter}_{avg_loss_pred:.4}_{avg_acc:.4}_{avg_loss_img:.4}.ckpt")
        if n_epoch > 0 and n_epoch % args.freq_decay == 0:
            trainer.lr *= args.weight_decay
        n_epoch += 1


def main():
    args = get_parser().parse_args()
    if args.mode == "train":
        train(args)
    else:   # the reference calls an undefined test() here (train.py:188-191); run a validation pass instead
        model = Slices3DRegModel(img_size=args.img_size, n_slices=args.n_slices, mode=args.mode).cuda().eval()
        print(val_step(model, make_loaders(args, 0, 1)[1], args.pred_type))


if __name__ == "__main__":
    main()
