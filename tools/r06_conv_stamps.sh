# 100 MHz stamps of conv3x3_lds_f16x3_kernel's phases over one LDM denoise step (a -DC3_STAMPS build of conv.o, not the shipped library)
cd $GRAFT_REPO_ROOT/slice3d_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-pass-failed -DC3_STAMPS -c conv.hip -o /tmp/conv_st.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v "_safe\|^conv.o") /tmp/conv_st.o -o /tmp/libslice3d_stamps.so || exit 1
cd $GRAFT_REPO_ROOT
S3D_HIP_LIB=/tmp/libslice3d_stamps.so python - <<'P'
import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from helpers import ldm_inputs
from test_ldm import LDM_FULL
from slice3d_amd.ldm_unet import UNetModel
from slice3d_amd.weights import load_seeded
m = load_seeded(UNetModel(**LDM_FULL), 0).cuda().eval()
x, t, cf = ldm_inputs(LDM_FULL, 1, 1)
x, t, cf = x.cuda(), t.cuda(), {k: v.cuda() for k, v in cf.items()}
lib = ctypes.CDLL("/tmp/libslice3d_stamps.so")
out = (ctypes.c_ulonglong * 32)()
for _ in range(3):
    m(x, t, c_fmaps=cf)
torch.cuda.synchronize(); lib.s3d_debug_c3_stamps(out)
for _ in range(5):
    m(x, t, c_fmaps=cf)
torch.cuda.synchronize(); lib.s3d_debug_c3_stamps(out)
names = ["start -> first fetch issued, table in LDS", "first halo arrived + parked", "barrier", "nine taps (all chunks)", "next halo parked", "barrier", "partial stores issued"]
for c, cn in enumerate(["gridDim.x = 96 (64 x 64 maps)", "48 (32 x 32)", "12 (8 x 8 / 4 x 4)", "other"]):
    n = out[c * 8 + 7]
    if not n: continue
    print("%s: %d workgroups" % (cn, n))
    for i, nm in enumerate(names):
        print("    %-45s %6.2f us per workgroup" % (nm, out[c * 8 + i] / n / 100.0))
P
