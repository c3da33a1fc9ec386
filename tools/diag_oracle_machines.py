"""Diagnostic: committed oracle golden (authoring container) against a live oracle on this box, per tensor, fp64 and fp32."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import test_gpu_train as T

print(torch.__version__, torch.backends.cpu.get_cpu_capability(), torch.get_num_threads(), flush=True)
b, s, q, ns = 1, 128, 16384, 12
zg = T._smooth_case_oracle(b, s, q, ns)
os.environ["S3D_LIVE_ORACLE"] = "1"
T._smooth_oracle.clear()
zl = T._smooth_case_oracle(b, s, q, ns)
print("sdf32 golden vs live: %.3e" % np.abs(zg["sdf_pred"] - zl["sdf_pred"]).max())
for kind in ("g64", "g32"):
    rows = []
    for k in zg["grad_names"]:
        k = str(k)
        a, c = zg[kind + ":" + k].astype(np.float64), zl[kind + ":" + k].astype(np.float64)
        rows.append((float(np.linalg.norm(a - c) / max(np.linalg.norm(c), 1e-300)), k))
    rows.sort(reverse=True)
    print("== %s golden vs live: worst" % kind)
    for e, k in rows[:10]:
        print("   %.3e %s" % (e, k))
    print("   median %.3e" % rows[len(rows) // 2][0])
