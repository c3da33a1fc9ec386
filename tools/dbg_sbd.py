"""Atomic-free sampling backward (train_sbd.hip) against the atomic kernels it replaces: the same step twice, S3D_SBD_OFF=1 / 0.
   python tools/dbg_sbd.py [b s q ns prec]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slice3d_amd.models import Slices3DRegModel
from slice3d_amd.trainer import HipTrainer
from slice3d_amd.weights import load_seeded
from slice3d_amd.synth import make_feed_dict
b, s, q, ns = (int(v) for v in (sys.argv[1:5] or (2, 32, 4200, 3)))
prec = sys.argv[5] if len(sys.argv) > 5 else "f32"
fd = make_feed_dict(b, s, q, ns, seed=5, device="cuda")
m = load_seeded(Slices3DRegModel(img_size=s, n_slices=ns, mode="train"), 0).cuda()
tr = HipTrainer(m, prec=prec, dropout=0.0, seed=3)
stats0 = {k: v.clone() for k, v in m.state_dict().items() if "running" in k}
res = {}
for mode in ("1", "0", "0"):
    os.environ["S3D_SBD_OFF"] = mode
    tr._calls = 0
    m.load_state_dict(stats0, strict=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    losses = tr.forward_backward(fd).cpu()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("S3D_SBD_OFF=%s: %.1f ms, losses %s" % (mode, dt * 1e3, losses.tolist()))
    res.setdefault(mode, []).append(tr.grad_flat.clone())
ga, gb, gc = res["1"][0], res["0"][0], res["0"][1]
print("dense run to run bit-identical:", bool(torch.equal(gb, gc)))
worst = []
for k, p in zip(tr.names, tr.params):
    off, n = tr.offsets[k], p.numel()
    x, y = ga[off:off + n], gb[off:off + n]
    if float(x.norm()) > 1e-6 * float(ga.norm()):
        worst.append((float((x - y).norm()) / float(x.norm()), k))
worst.sort(reverse=True)
for d, k in worst[:4]:
    print("  %.3e  %s" % (d, k))
for d, k in worst:
    if k.endswith(("up4.up.weight", "up3.up.weight", "up2.up.weight", "up1.up.weight", "down5.34.weight", "fc_s.weight")):
        print("  %.3e  %s" % (d, k))
