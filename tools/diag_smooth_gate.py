"""Diagnostic: the smooth-gradient case (1, 128, 16384, 12) in f32 against the committed oracle golden and against a live oracle
on this box: per tensor rel(hip, ref64), rel(ref32, ref64) for the ten worst ratios of each."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch

import test_gpu_train as T
from helpers import fp64_anchored_rows
from slice3d_amd.models import Slices3DRegModel
from slice3d_amd.synth import make_feed_dict
from slice3d_amd.weights import load_seeded

b, s, q, ns = 1, 128, 16384, 12
fd = make_feed_dict(b, s, q, ns, seed=4000 + q)
w_sdf, w_rec, w_vgg = T._smooth_output_grads(b, s, q, ns, seed=q)
grads = {}
for prec in ("f32", "f16x3"):
    m = load_seeded(Slices3DRegModel(n_slices=ns, mode="train", prec=prec), 0).cuda().train()
    m.train_dropout = 0.0
    out = m({k: v.cuda() for k, v in fd.items()})
    ((out["sdf_pred"] * w_sdf.cuda()).sum() + (out["slices_rec"] * w_rec.cuda()).sum() + w_vgg * out["vgg_loss"]).backward()
    grads[prec] = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
for tag in ("golden", "live"):
    if tag == "live":
        os.environ["S3D_LIVE_ORACLE"] = "1"
        T._smooth_oracle.clear()
    z = T._smooth_case_oracle(b, s, q, ns)
    for prec in ("f32", "f16x3"):
        rows = fp64_anchored_rows(z, grads[prec], skip=T.PRE_BN_BIASES)
        med_ref = sorted(r[2] for r in rows)[len(rows) // 2]
        med_hip = sorted(r[1] for r in rows)[len(rows) // 2]
        rows.sort(key=lambda r: -r[1] / (3 * max(r[2], med_ref) + 3e-4))
        print("== %s oracle, %s: median hip %.2e ref %.2e" % (tag, prec, med_hip, med_ref))
        for k, eh, er, e32 in rows[:8]:
            print("   %-55s hip %.2e  ref32 %.2e  hip-vs-ref32 %.2e  ratio-to-gate %.2f" % (k, eh, er, e32, eh / (3 * max(er, med_ref) + 3e-4)))
