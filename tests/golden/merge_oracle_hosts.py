"""Adds the authoring container's fp32 oracle pass to the committed training goldens (round 6).

tests/golden/oracle_{smooth_*,train_128_16k,train_full_b4_s256}.npz were written on the GPU box's host CPU (2x EPYC 9575F) in
round 5: they hold per tensor the sampled entries of that host's fp32 pass ('g32:') and of the fp64 pass ('g64:').  An fp32
evaluation of the network rounds differently on different hosts (profiles/r05_oracle_hosts.md), so a gate stated in units of ONE
host's fp32 distance depends on where the file was written.  This script takes the same cases generated in the authoring container
    S3D_ORACLE_GOLDEN_DIR=/tmp/oracle_container python tests/golden/make_oracle_golden.py smooth_b1_s128_q16384_n12 train_128_16k train_full_b4_s256
checks that the two hosts' fp64 passes (and every non-gradient array) agree, and stores the container's fp32 samples beside the
box's as 'g32c:<name>'.  tests/helpers.py gates on the LARGER of the two fp32 distances.
    python tests/golden/merge_oracle_hosts.py /tmp/oracle_container"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ("smooth_b1_s128_q16384_n12", "train_128_16k", "train_full_b4_s256")

if __name__ == "__main__":
    src = sys.argv[1]
    for name in CASES:
        pc, pb = os.path.join(src, "oracle_%s.npz" % name), os.path.join(HERE, "oracle_%s.npz" % name)
        if not os.path.isfile(pc):
            print("%-28s no container file, skipped" % name)
            continue
        zb, zc = dict(np.load(pb)), dict(np.load(pc))
        worst64, worst32, n = 0.0, 0.0, 0
        for k in list(zb):
            if k.startswith("g64:"):
                if max(np.linalg.norm(zb[k]), np.linalg.norm(zc[k])) < 1e-8 * np.sqrt(zb[k].size):
                    continue                       # the pre-BatchNorm biases: exact gradient 0, rounding noise on every host
                d = np.linalg.norm(zb[k] - zc[k]) / max(np.linalg.norm(zb[k]), 1e-300)
                worst64 = max(worst64, d)
                assert d < 1e-9, (name, k, d)      # the fp64 anchor is host-independent
            elif k.startswith("g32:"):
                kk = k[4:]
                if np.linalg.norm(zb[k]) >= 1e-8 * np.sqrt(zb[k].size):   # (not the exactly-zero gradients of the pre-BatchNorm biases)
                    worst32 = max(worst32, np.linalg.norm(zb[k].astype(np.float64) - zc[k]) / np.linalg.norm(zb[k]))
                zb["g32c:" + kk] = zc[k]
                n += 1
            elif not k.startswith(("g32c:",)):
                assert zb[k].shape == zc[k].shape and (zb[k].dtype.kind not in "fc" or np.allclose(zb[k], zc[k], rtol=1e-4, atol=5e-5)), (name, k)
        np.savez_compressed(pb, **zb)
        print("%-28s %d tensors: fp64 passes agree to %.1e, fp32 passes differ by up to %.1e; %.2f MB" % (name, n, worst64, worst32, os.path.getsize(pb) / 1e6))
