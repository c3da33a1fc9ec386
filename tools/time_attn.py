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
for _ in range(2): m.decode_sdf(fd["qry_norot"], code)
torch.cuda.synchronize(); lib.s3d_prof_enable(1)
for _ in range(5): m.decode_sdf(fd["qry_norot"], code)
torch.cuda.synchronize()
for i, n in enumerate(_lib.PROF_NAMES):
    ms, cnt = C.c_double(), C.c_long(); lib.s3d_prof_read(i, C.byref(ms), C.byref(cnt))
    if cnt.value: print("%-14s %.3f ms/step" % (n, ms.value / 5))
