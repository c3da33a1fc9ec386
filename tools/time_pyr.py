"""Times s3d_sample_pyramid_fwd and a plain device memset of the same output size (write-bandwidth yardstick)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slice3d_amd.models import Slices3DRegModel, LEVEL_CHANNELS
from slice3d_amd.synth import make_feed_dict
from slice3d_amd.weights import load_seeded
S, Q, ns = 256, 100000, 12
m = load_seeded(Slices3DRegModel(img_size=S, n_slices=ns, mode="test"), 0).cuda().eval()
fd = make_feed_dict(1, S, Q, ns, seed=1, with_slices=False, device="cuda")
code = m.encode(fd, build_latent=False)
g = m.project_coord(fd["qry_norot"] * torch.tensor([1.0, -1.0, -1.0], device="cuda"), fd["trans_mat_wo_rot_tp"])
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: m.sample_pyramid(code.pyramid, g))
out = torch.empty(ns, Q, 992, device="cuda")
ms_set = t(lambda: out.zero_())
src = torch.randn(ns, Q, 992, device="cuda")
ms_cp = t(lambda: out.copy_(src))
gb = out.numel() * 4 / 1e9
print("sample_pyramid %.3f ms (%.2f TB/s on %.2f GB written) | memset %.3f ms (%.2f TB/s) | copy %.3f ms (%.2f TB/s r+w)"
      % (ms, gb / ms, gb, ms_set, gb / ms_set, ms_cp, 2 * gb / ms_cp))
