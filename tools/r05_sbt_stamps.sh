cd $GRAFT_REPO_ROOT
S3D_HIP_LIB=$PWD/build/abl/lib_sbt_stamps.so python - > gpurun_out/r05_sbt_stamps.log 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from slice3d_amd.models import Slices3DRegModel
from slice3d_amd.trainer import HipTrainer
from slice3d_amd.weights import load_seeded
from slice3d_amd.synth import make_feed_dict
m = load_seeded(Slices3DRegModel(n_slices=12, mode="train"), 0).cuda()
tr = HipTrainer(m, prec="f16x3")
tr.dropout = 0.1
fd = make_feed_dict(4, 256, 100000, 12, seed=1, device="cuda")
tr.train_step(fd)
torch.cuda.synchronize()
PY
grep -c SBT2 gpurun_out/r05_sbt_stamps.log; grep SBT2 gpurun_out/r05_sbt_stamps.log | sort | head -40; tail -3 gpurun_out/r05_sbt_stamps.log
