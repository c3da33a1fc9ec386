"""train_gt.py — host entry point kept from the reference (reg_slices/train_gt.py): same flags
(options.get_parser), same loop (train_step / val_step / checkpoint naming `{epoch}_{iter}_{loss:.4}_{acc:.4}.ckpt`,
train_gt.py:153 / lr decay), for Slices3DGTModel — the regression model that reads the GIVEN slice images.  The
compute is the HIP path (slice3d_amd.trainer.HipGtTrainer = zero_grad + forward + L1 + backward + Adam of
train_gt.py:38-52).  One process per GPU under torch.distributed.run for data-parallel training (gradients are
averaged over ranks with one RCCL all-reduce; the reference uses nn.DataParallel, train_gt.py:110-111).

    python reg_slices/train_gt.py --name_exp demo_gt --name_dataset synthetic --img_size 64 --n_qry 256 --n_bs 2
"""
import glob
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from options import get_parser  # noqa: E402
from slice3d_amd.models_gt import Slices3DGTModel  # noqa: E402
from slice3d_amd.trainer import HipGtTrainer  # noqa: E402
from train import cal_acc, make_loaders  # noqa: E402


def train_step(batch, trainer, args=None):
    """train_gt.py:38-52 -> (loss_pred, acc) as python floats."""
    return trainer.train_step(batch)


@torch.no_grad()
def val_step(model, val_loader, pred_type="sdf"):
    """train_gt.py:55-73: eval-mode forward, mean L1(sdf) and sign accuracy over the batches."""
    tot_l, tot_a, n = 0.0, 0.0, 0
    for batch in val_loader:
        batch = {k: v.cuda() for k, v in batch.items()}
        x = model(batch)
        tot_l += float((x["sdf_pred"] - batch["sdf"]).abs().mean())
        tot_a += float(cal_acc(x, batch))
        n += 1
    n = max(n, 1)
    return tot_l / n, tot_a / n


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
    model = Slices3DGTModel(img_size=args.img_size, n_slices=args.n_slices, mode=args.mode).cuda()
    if world > 1:   # identical initial weights on every rank
        for t in list(model.parameters()) + list(model.buffers()):
            torch.distributed.broadcast(t.data, 0)
    trainer = HipGtTrainer(model, lr=args.lr, dropout=args.dropout, seed=args.seed * 65537 + rank,   # per-rank dropout streams
                         sync_bn=args.sync_bn)
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
            loss_pred, acc = train_step(batch, trainer, args)
            if n_iter % args.freq_log == 0 and rank == 0:
                print("[train] epcho:", n_epoch, " ,iter:", n_iter, " loss_pred:", loss_pred, " acc:", acc)
            n_iter += 1
        if n_epoch % args.freq_ckpt == 0 and rank == 0:
            model.eval()
            avg_loss_pred, avg_acc = val_step(model, val_loader, args.pred_type)
            print("[val] epcho:", n_epoch, " ,iter:", n_iter, " avg_loss_pred:", avg_loss_pred, " acc:", avg_acc)
            torch.save({"model": model.state_dict(), "opt": trainer.state_dict(), "n_epoch": n_epoch, "n_iter": n_iter},
                       f"{dir_ckpt}/{n_epoch}_{n_iter}_{avg_loss_pred:.4}_{avg_acc:.4}.ckpt")
        if n_epoch > 0 and n_epoch % args.freq_decay == 0:
            trainer.lr *= args.weight_decay
        n_epoch += 1


def main():
    args = get_parser().parse_args()
    if args.mode == "train":
        train(args)
    else:   # the reference calls an undefined test() here (train_gt.py:166-170); run a validation pass instead
        model = Slices3DGTModel(img_size=args.img_size, n_slices=args.n_slices, mode=args.mode).cuda().eval()
        print(val_step(model, make_loaders(args, 0, 1)[1], args.pred_type))


if __name__ == "__main__":
    main()
