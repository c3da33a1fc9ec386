"""Repeat the GT decode that failed intermittently and print the error against the oracle each time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import seeded_sd_from_shapes
from test_gt_oracle import gt_shapes
from oracle import ref_cpu
from slice3d_amd.models_gt import Slices3DGTModel
from slice3d_amd.synth import make_feed_dict
from slice3d_amd.weights import load_seeded
b, s, q, ns, mode = 2, 48, 4500, 5, "test"
fd = make_feed_dict(b, s, q, ns, seed=900 + q)
sd = seeded_sd_from_shapes(gt_shapes())
with torch.no_grad():
    want, _ = ref_cpu.gt_forward(sd, fd, mode, ns)
m = load_seeded(Slices3DGTModel(n_slices=ns, mode=mode, prec="f16x3"), 0).cuda().eval()
fdc = {k: v.cuda() for k, v in fd.items()}
prev = None
for it in range(40):
    got = m(fdc)["sdf_pred"].cpu()
    e = (got - want).abs()
    msg = "iter %2d max err %.3e" % (it, float(e.max()))
    if prev is not None and not torch.equal(prev, got):
        d = (got - prev).abs()
        idx = torch.nonzero(d > 0)
        msg += "  DIFFERS from previous run at %d outputs (max %.3e), e.g. %s" % (idx.shape[0], float(d.max()), idx[:6].tolist())
    print(msg)
    prev = got
