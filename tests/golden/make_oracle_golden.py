"""Writes tests/golden/oracle_*.npz: the CPU ORACLE's (oracle/ref_cpu.py) outputs for the slow fixed-seed cases of the GPU
suite, so that the GPU box stops recomputing them in every `-m gpu` run (370 of 615 s in round 4).

The numbers come from the very functions the tests call when a file is missing (or S3D_LIVE_ORACLE=1): this script imports
the test modules and runs their `_oc_*` / `_*_oracle*` functions with S3D_LIVE_ORACLE=1, in the authoring container (CPU only;
the full-size training case takes ~20-30 minutes of its 8 cores because of the fp64 pass).  Nothing here touches the
reference: these are goldens of the ORACLE (itself pinned against the reference by tests/test_oracle.py and make_golden.py),
and every family keeps a live oracle case in the suite.

    python tests/golden/make_oracle_golden.py [case ...]        # default: all cases
    S3D_ORACLE_GOLDEN_DIR=gpurun_out/golden python tests/golden/make_oracle_golden.py train_full_b4_s256 ...

WHERE the three training cases were generated matters: their gate measures the HIP gradients' distance from the fp64 oracle in
units of the fp32 oracle's own distance, and ATen's fp32 CPU kernels round differently on different hosts — the fp64 pass of
the authoring container (Intel Xeon) and of the GPU box's host (2x EPYC 9575F) agree to 1e-13 per tensor, the two fp32 passes
differ from each other by up to 9e-3 in the encoder's weight gradients (profiles/r05_oracle_hosts.md).  The committed
train_* / smooth_* files were therefore written on the GPU box's host CPU (this script under gpurun, S3D_ORACLE_GOLDEN_DIR
pointing into gpurun_out/), the yardstick the live tests of rounds 3-4 used; the forward-only cases are host-independent at
the 1e-4 gate and were written in the authoring container.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
sys.path[:0] = [ROOT, TESTS]
os.environ["S3D_LIVE_ORACLE"] = "1"

import torch  # noqa: E402

import test_gpu_gt  # noqa: E402
import test_gpu_parity  # noqa: E402
import test_gpu_train  # noqa: E402
from helpers import golden_digest, oracle_golden_path  # noqa: E402

CASES = {
    "full256": test_gpu_parity._oc_full256,
    "white_s256_q6000": test_gpu_parity._oc_white,
    "sweep24": test_gpu_parity._oc_sweep,
    "gt_white_s128_q5000": test_gpu_gt._oc_gt_white,
    "train_128_16k": test_gpu_train._big_case_oracle,
    "smooth_b1_s128_q16384_n12": lambda: test_gpu_train._smooth_case_oracle(1, 128, 16384, 12),
    "train_full_b4_s256": test_gpu_train._full_size_oracle_compact,
}

def write_digests():
    """tests/golden/oracle_digests.json: sha256 of the ARRAYS of every committed oracle_*.npz (helpers.golden_digest).
    tests/test_oracle.py checks the committed files against it and recomputes the host-independent forward case `full256` live."""
    import glob
    import json
    out = {}
    for p in sorted(glob.glob(os.path.join(HERE, "oracle_*.npz"))):
        z = np.load(p)
        out[os.path.basename(p)] = golden_digest({k: z[k] for k in z.files})
        print("%-44s %s" % (os.path.basename(p), out[os.path.basename(p)]))
    # the sources the goldens are a function of: an edit of the oracle or of the synthetic inputs makes the test ask for
    # a regeneration instead of letting the files go stale silently
    import hashlib
    out["_sources"] = {f: hashlib.sha256(open(os.path.join(ROOT, f), "rb").read()).hexdigest()
                       for f in ("oracle/ref_cpu.py", "slice3d_amd/synth.py")}
    json.dump(out, open(os.path.join(HERE, "oracle_digests.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    torch.manual_seed(0)
    if sys.argv[1:] == ["--digests"]:
        write_digests()
        sys.exit(0)
    for name in (sys.argv[1:] or list(CASES)):
        t0 = time.time()
        z = CASES[name]()
        path = oracle_golden_path(name, os.environ.get("S3D_ORACLE_GOLDEN_DIR"))
        os.makedirs(os.path.dirname(path), exist_ok=True)
        np.savez_compressed(path, **{k: np.asarray(v) for k, v in z.items()})
        print("%-28s %6.1f s  %7.2f MB  %d arrays  sha256(arrays) %s" % (name, time.time() - t0, os.path.getsize(path) / 1e6, len(z),
                                                                         golden_digest({k: np.asarray(v) for k, v in z.items()})), flush=True)
    if not os.environ.get("S3D_ORACLE_GOLDEN_DIR"):
        write_digests()
