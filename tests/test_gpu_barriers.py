"""The attention kernels' ring barriers wait with hand-counted `s_waitcnt vmcnt(N)` (AQ_BARRIER_N, slice3d_amd/csrc/attnq.h):
correct only while hipcc emits exactly the counted vector-memory operations.  `make` also builds the same library with every
counted barrier in its always-safe vmcnt(8) form (libslice3d_hip_safe.so, -DS3D_AQ_SAFE_BARRIERS); a too-large count would let a
wave read an LDS slot before its DMA landed — wrong rows — so the two libraries must agree bit for bit on the inference
decode (attn_layer_q_kernel), on the training forward (its TRAIN form) and on the fused backward (attn_bwd_q_kernel; checked
through the gradients no float atomic sits upstream of)."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "slice3d_amd", "csrc")

_SCRIPT = r"""
import sys, json, hashlib
sys.path.insert(0, %(root)r)
import torch
from slice3d_amd import _lib
from slice3d_amd.models import Slices3DRegModel
from slice3d_amd.synth import make_feed_dict
from slice3d_amd.trainer import HipTrainer
from slice3d_amd.weights import load_seeded
assert _lib.LIB_PATH.endswith(%(libname)r), _lib.LIB_PATH
sha = lambda t: hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()
out = {}
# inference: 12 slices (T = 13 >= 9: the counted form) and 4 slices (T = 5: the vmcnt(8) fallback), many items per workgroup
for ns, q in ((12, 70000), (4, 9000)):
    m = load_seeded(Slices3DRegModel(n_slices=ns, mode="test", prec="f16x3"), 0).cuda().eval()
    fd = {k: v.cuda() for k, v in make_feed_dict(2, 64, q, ns, seed=5, with_slices=False).items()}
    out["infer_%%d" %% ns] = sha(m.decode_sdf(fd["qry_norot"], m.encode(fd)))
# training step with dropout: forward (TRAIN form of the kernel) and fused backward
m = load_seeded(Slices3DRegModel(n_slices=12, mode="train"), 0).cuda()
tr = HipTrainer(m, prec="f16x3", dropout=0.1, seed=3, process_group=False)
batch = {k: v.cuda() for k, v in make_feed_dict(2, 32, 6000, 12, seed=8).items()}
losses = tr.forward_backward(batch)
torch.cuda.synchronize()
det = torch.cat([tr.grad_flat[tr.offsets[k]:tr.offsets[k] + p.numel()] for k, p in zip(tr.names, tr.params)
                 if k.startswith(("att_decoder.", "fc_out.", "fc_p."))])
out["train_sdf_loss"] = sha(losses[:1])
out["train_det_grads"] = sha(det)
out["grad_norm"] = float(det.norm())
print("RESULT " + json.dumps(out))
"""


def _run(libname):
    lib = os.path.join(CSRC, libname)
    assert os.path.isfile(lib), "%s is not built (make -C slice3d_amd/csrc)" % lib
    env = dict(os.environ, S3D_HIP_LIB=lib)
    r = subprocess.run([sys.executable, "-c", _SCRIPT % {"root": ROOT, "libname": libname}], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


def test_counted_ring_barriers_match_the_always_safe_build_bit_for_bit():
    counted, safe = _run("libslice3d_hip.so"), _run("libslice3d_hip_safe.so")
    assert counted["grad_norm"] > 0
    assert counted == safe, (counted, safe)
