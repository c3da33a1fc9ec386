import os, sys, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slice3d_amd import _lib
from slice3d_amd.models import Slices3DRegModel
from slice3d_amd.weights import load_seeded
from slice3d_amd.synth import make_feed_dict
m = load_seeded(Slices3DRegModel(n_slices=12, mode="test", prec="f16x3"), 0).cuda().eval()
fd = {k: v.cuda() for k, v in make_feed_dict(1, 256, 100000, 12, seed=1, with_slices=False).items()}
lib = _lib.load()
code = m.encode(fd)
def t(q, name):
    for _ in range(2): m.decode_sdf(q, code)
    torch.cuda.synchronize(); lib.s3d_prof_enable(1)
    for _ in range(5): m.decode_sdf(q, code)
    torch.cuda.synchronize()
    ms, cnt = C.c_double(), C.c_long(); lib.s3d_prof_read(_lib.PROF_SAMPLE, C.byref(ms), C.byref(cnt))
    print("%-28s sample_tokens %.3f ms" % (name, ms.value / 5)); lib.s3d_prof_enable(0)
q = fd["qry_norot"]
t(q, "random queries")
t(torch.zeros_like(q), "all queries identical")
g = torch.linspace(-0.5, 0.5, 47, device="cuda")
grid = torch.stack(torch.meshgrid(g, g, g, indexing="ij"), -1).reshape(1, -1, 3)[:, :100000].contiguous()
t(grid, "grid order (z fastest)")
