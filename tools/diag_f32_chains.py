"""Round 6 diagnostic: distance of the HIP parameter gradients from the fp64 oracle in the smooth-gradient case (1 object x 128^2 x
12 slices x 16 384 queries), per arithmetic mode, for the library S3D_HIP_LIB selects (tools/patches/f32_short_chains.patch builds the
variants with shorter fp32 accumulation chains).  Prints the encoder / decoder conv-weight rows and the medians.
    S3D_HIP_LIB=build/abl/lib_bothshort.so python tools/diag_f32_chains.py"""
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
z = T._smooth_case_oracle(b, s, q, ns)
print("library:", os.environ.get("S3D_HIP_LIB", "product"))
for prec in (sys.argv[1:] or ["f32", "f16x3"]):
    m = load_seeded(Slices3DRegModel(n_slices=ns, mode="train", prec=prec), 0).cuda().train()
    m.train_dropout = 0.0
    out = m({k: v.cuda() for k, v in fd.items()})
    ((out["sdf_pred"] * w_sdf.cuda()).sum() + (out["slices_rec"] * w_rec.cuda()).sum() + w_vgg * out["vgg_loss"]).backward()
    grads = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    rows = fp64_anchored_rows(z, grads, skip=T.PRE_BN_BIASES)
    med_hip = sorted(r[1] for r in rows)[len(rows) // 2]
    med_ref = sorted(r[2] for r in rows)[len(rows) // 2]
    enc = [r for r in rows if r[0].startswith(("unet.down", "unet.inc")) and r[0].endswith(".weight") and "bn" not in r[0]]
    worst = max(rows, key=lambda r: r[1])
    print("== %s: median rel(hip, fp64) %.2e (fp32 oracle of the golden's host: %.2e); worst %s %.2e" % (prec, med_hip, med_ref, worst[0], worst[1]))
    for k, eh, er, e32 in sorted(rows, key=lambda r: -r[1])[:10]:
        print("   %-50s hip %.2e   fp32 oracle %.2e" % (k, eh, er))
