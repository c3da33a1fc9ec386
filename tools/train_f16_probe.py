"""Round 6: the training step's single-pass f16 throughput mode (S3D_PREC_F16) beside the split-precision step.
(1) gradient deviation on one step from identical weights / batch / dropout masks (relative L2 per tensor);
(2) step time at the bench's size; (3) a loss trajectory of N Adam steps on a fixed synthetic set, both modes from the same start.
    python tools/train_f16_probe.py [steps=200]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slice3d_amd.models import Slices3DRegModel
from slice3d_amd.synth import make_feed_dict
from slice3d_amd.trainer import HipTrainer
from slice3d_amd.weights import load_seeded

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def trainer(prec, dropout, seed=0):
    m = load_seeded(Slices3DRegModel(img_size=256, n_slices=12, mode="train"), 0).cuda()
    return m, HipTrainer(m, dropout=dropout, seed=seed, prec=prec, process_group=False)


# ---- (1) one step, same masks ----
fd = make_feed_dict(2, 128, 16384, 12, seed=77, device="cuda")
grads = {}
for prec in ("f16x3", "f16"):
    m, tr = trainer(prec, 0.1)
    losses, _, _ = tr.forward_backward(fd, want_outputs=True)
    grads[prec] = ({k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}, losses.cpu().numpy().copy())
    del m, tr
PRE = {"slices_generator.%s.bias" % k for k in ("down1.0", "down2.7", "down3.14", "down3.17", "down4.24", "down4.27", "down5.34", "down5.37")}   # exact gradient 0
rel = sorted(((float((grads["f16"][0][k] - g).norm() / g.norm()), k) for k, g in grads["f16x3"][0].items() if k not in PRE and float(g.norm()) > 0), reverse=True)
print("one step, 2 x 128^2 x 16 384 queries, dropout 0.1: losses f16x3 %s | f16 %s" % (grads["f16x3"][1], grads["f16"][1]))
print("  relative L2 gradient deviation f16 vs f16x3: max %.3e (%s), median %.3e, tensors %d" % (rel[0][0], rel[0][1], rel[len(rel) // 2][0], len(rel)))
dec = [r for r in rel if r[1].startswith(("att_decoder", "fc_"))]
enc = [r for r in rel if r[1].startswith("slices_generator")]
print("  decoder tensors: max %.3e median %.3e | U-Net tensors: max %.3e median %.3e" % (dec[0][0], dec[len(dec) // 2][0], enc[0][0], enc[len(enc) // 2][0]))

# ---- (2) step time at the bench's size ----
fdb = make_feed_dict(4, 256, 100000, 12, seed=4321, device="cuda")
for prec in ("f16x3", "f16", "f16x3", "f16"):
    m, tr = trainer(prec, 0.1)
    tr.train_step(fdb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        tr.train_step(fdb)
    torch.cuda.synchronize()
    print("train step %-6s %.2f ms (4 objects x 100 000 queries, dropout 0.1)" % (prec, (time.perf_counter() - t0) / 8 * 1e3))
    del m, tr
del fdb

# ---- (3) loss trajectory ----
batches = [make_feed_dict(2, 128, 8192, 12, seed=900 + i, device="cuda") for i in range(8)]
traj = {}
for prec in ("f16x3", "f16"):
    m, tr = trainer(prec, 0.1, seed=5)
    out = []
    for s in range(steps):
        l = tr.train_step(batches[s % len(batches)])
        if s % 10 == 9 or s == 0:
            out.append((s + 1, [float(v) for v in l[:4]]))
    traj[prec] = out
    del m, tr
print("| step | f16x3: loss_pred / loss_img / loss_img_vgg / acc | f16: loss_pred / loss_img / loss_img_vgg / acc |")
print("|---|---|---|")
for (s, a), (_, b) in zip(traj["f16x3"], traj["f16"]):
    print("| %d | %s | %s |" % (s, " / ".join("%.5f" % v for v in a), " / ".join("%.5f" % v for v in b)))
