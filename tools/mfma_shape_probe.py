"""Runs tools/mfma_shape_probe.hip's four variants (16x16x32 / 32x32x16 f16 MFMA, operands in registers or B re-read from LDS) for a
few seconds each, alternating twice, with the shader clock and socket power sampled beside them (bench.py's sampler).
Usage (GPU box): python tools/mfma_shape_probe.py [seconds]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import _ClockSampler  # noqa: E402

exe = "/tmp/mfma_shape_probe"
subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", os.path.join(ROOT, "tools/mfma_shape_probe.hip"), "-o", exe], check=True)
secs = sys.argv[1] if len(sys.argv) > 1 else "4"
print("| variant | shape | B from LDS | TFLOP/s first launch | TFLOP/s settled | sclk MHz | W |\n|---|---|---|---|---|---|---|")
for rep in range(2):
    for v in (0, 1, 2, 3):
        with _ClockSampler() as cs:
            r = subprocess.run([exe, str(v), secs], capture_output=True, text=True)
        if r.returncode:
            print("variant", v, "failed:", r.stderr[-400:])
            continue
        d = json.loads(r.stdout.strip().splitlines()[-1])
        sclk, watts = cs.result()
        print("| %d | %s | %s | %.0f | %.0f | %s | %s |" % (v, d["shape"], d["b_from_lds"], d["tflops_first"], d["tflops_settled"],
                                                          "%.0f" % sclk if sclk else "-", "%.0f" % watts if watts else "-"))
        sys.stdout.flush()
