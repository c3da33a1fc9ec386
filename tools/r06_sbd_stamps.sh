# cycle stamps of sample_bwd_dense_kernel's phases (a -DSBD_STAMPS build of train_sbd.o, not the shipped library)
cd $GRAFT_REPO_ROOT/slice3d_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSBD_STAMPS -c train_sbd.hip -o /tmp/train_sbd_st.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v "_safe\|train_sbd.o") /tmp/train_sbd_st.o -o /tmp/libslice3d_stamps.so || exit 1
cd $GRAFT_REPO_ROOT
S3D_HIP_LIB=/tmp/libslice3d_stamps.so python - <<'P'
import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from slice3d_amd import _lib
from slice3d_amd.models import Slices3DRegModel
from slice3d_amd.trainer import HipTrainer
from slice3d_amd.weights import load_seeded
from slice3d_amd.synth import make_feed_dict
fd = make_feed_dict(4, 256, 100000, 12, seed=1, device="cuda")
m = load_seeded(Slices3DRegModel(n_slices=12, mode="train"), 0).cuda()
tr = HipTrainer(m, prec="f16x3", dropout=0.1, seed=3)
lib = ctypes.CDLL("/tmp/libslice3d_stamps.so")
out = (ctypes.c_ulonglong * 8)()
tr.forward_backward(fd); torch.cuda.synchronize(); lib.s3d_debug_sbd_stamps(out)
tr.forward_backward(fd); torch.cuda.synchronize(); lib.s3d_debug_sbd_stamps(out)
names = ["prologue + k-steps", "wait barrier 1", "staging", "wait barrier 2", "epilogue", "vmcnt(0) after barrier 1"]
tot = sum(out[i] for i in range(6))
for i, n in enumerate(names):
    print("%-20s %6.1f %%   %.1f us per workgroup (100 MHz clock)" % (n, 100.0 * out[i] / tot, out[i] / 12288 / 100.0))
P
