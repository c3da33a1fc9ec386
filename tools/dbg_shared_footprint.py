"""Token tensors of the shared-window and per-lane forms of the token builder, compared bit for bit (debug, round 6):
    python tools/dbg_shared_footprint.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slice3d_amd import _lib
from slice3d_amd.models import Slices3DRegModel
from slice3d_amd.weights import load_seeded
from slice3d_amd.synth import make_feed_dict
lib = _lib.load()
for prec in ("f32", "f16x3"):
    m = load_seeded(Slices3DRegModel(n_slices=12, mode="test", prec=prec), 0).cuda().eval()
    for size in (64, 256):
        fd = {k: v.cuda() for k, v in make_feed_dict(1, size, 2048, 12, seed=55, with_slices=False).items()}
        # clustered queries: every group of 16 inside a small ball, so that the windows fit
        g = torch.Generator(device="cuda").manual_seed(1)
        centers = fd["qry_norot"][:, ::16].repeat_interleave(16, dim=1)
        q = (centers + 0.004 * torch.randn(centers.shape, device="cuda", generator=g)).contiguous()
        code = m.encode(fd)
        lib.s3d_decode_set_shared_footprint(0)
        a = m.decode_stages(q, code)
        lib.s3d_decode_set_shared_footprint(1)
        b = m.decode_stages(q, code)
        ta, tb = a["fc_s"], b["fc_s"]   # (B, Q, 12, 128) slice tokens
        d = (ta != tb)
        print(prec, size, "tokens shape", tuple(ta.shape), "differing elements", int(d.sum()), "of", d.numel(),
              "max abs diff %.3e" % float((ta - tb).abs().max()), "| sdf differs:", int((a["sdf"] != b["sdf"]).sum()))
        if d.any():
            idx = d.nonzero()
            print("  per slice:", [int((idx[:, 2] == t).sum()) for t in range(12)])
            print("  per channel block of 16:", [int(((idx[:, 3] // 16) == j).sum()) for j in range(8)])
            print("  per query-in-group:", [int(((idx[:, 1] % 16) == j).sum()) for j in range(16)])
            i = idx[0].tolist()
            print("  first:", i, float(ta[tuple(i)]), float(tb[tuple(i)]))
