"""Per-tensor relative error of the smooth-output-gradient parity case (tests/test_gpu_train.py) — diagnosis aid."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_train import _shapes, _smooth_output_grads, PRE_BN_BIASES   # noqa: E402
from helpers import seeded_sd_from_shapes                                  # noqa: E402
from oracle import ref_cpu                                                  # noqa: E402
from slice3d_amd.models import Slices3DRegModel                             # noqa: E402
from slice3d_amd.synth import make_feed_dict                                # noqa: E402
from slice3d_amd.weights import load_seeded                                 # noqa: E402

b, s, q, ns = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (2, 32, 128, 12)))
fd = make_feed_dict(b, s, q, ns, seed=4000 + q)
w_sdf, w_rec, w_vgg = _smooth_output_grads(b, s, q, ns, seed=q)
sd = seeded_sd_from_shapes(_shapes(ns))
for k, v in sd.items():
    if v.is_floating_point() and "running" not in k and not k.startswith("vggptlossfunc"):
        v.requires_grad_(True)
_, _, out, _ = ref_cpu.forward_train(sd, fd, ns, 0.0)
((out["sdf_pred"] * w_sdf).sum() + (out["slices_rec"] * w_rec).sum() + w_vgg * out["vgg_loss"]).backward()
grads = {k: v.grad for k, v in sd.items() if v.grad is not None}
for prec in ("f32", "f16x3"):
    m = load_seeded(Slices3DRegModel(n_slices=ns, mode="train", prec=prec), 0).cuda().train()
    m.train_dropout = 0.0
    o = m({k: v.cuda() for k, v in fd.items()})
    ((o["sdf_pred"] * w_sdf.cuda()).sum() + (o["slices_rec"] * w_rec.cuda()).sum() + w_vgg * o["vgg_loss"]).backward()
    rels = []
    for k, p in m.named_parameters():
        if p.grad is None or k not in grads or k in PRE_BN_BIASES:
            continue
        rels.append((float((p.grad.cpu() - grads[k]).norm() / grads[k].norm()), k, float(grads[k].norm())))
    rels.sort(reverse=True)
    print(prec, "sdf err %.2e" % float((o["sdf_pred"].detach().cpu() - out["sdf_pred"].detach()).abs().max()))
    order = [k for k, _ in m.named_parameters()]
    for r, k, n in sorted(rels, key=lambda t: order.index(t[1])):
        if (prec == "f32" and not k.startswith("vgg")) or k.startswith(("att_decoder.layers.2", "fc_out", "att_decoder.layers.1.linear2", "att_decoder.layers.1.norm2")):
            print("   %-50s rel %.2e  |g| %.3e" % (k, r, n))
    print("   median %.2e" % rels[len(rels) // 2][0])
    if prec == "f32":
        g = dict(m.named_parameters())["fc_p.weight"].grad.cpu()
        print("fc_p.weight hip:\n", g[:6], "\nref:\n", grads["fc_p.weight"][:6])
        print("fc_p.bias hip:", dict(m.named_parameters())["fc_p.bias"].grad.cpu()[:6], "ref:", grads["fc_p.bias"][:6])
