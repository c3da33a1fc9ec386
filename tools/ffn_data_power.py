"""Is the decoder FFN kernel limited by the socket's power?  Same kernel, same launches, two operand sets:
the bench's seeded random weights, and all-zero decoder weights (every MFMA product is 0 x 0: no toggling in the
multipliers).  Prints the FFN stage time per launch (HIP events, s3d_prof) with the clock / power rocm-smi reports
while each runs.  Run on the GPU box: python tools/ffn_data_power.py > profiles/..."""
import ctypes as C
import os
import re
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slice3d_amd import _lib                                  # noqa: E402
from slice3d_amd.models import Slices3DRegModel               # noqa: E402
from slice3d_amd.synth import make_feed_dict                  # noqa: E402
from slice3d_amd.weights import load_seeded                   # noqa: E402


def sample_smi(stop, out):
    while not stop.is_set():
        try:
            t = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            clk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", t)
            pw = re.search(r"Package Power \(W\): ([\d.]+)", t)
            if clk and pw:
                out.append((int(clk.group(1)), float(pw.group(1))))
        except Exception:
            pass
        time.sleep(0.2)


def run(model, fd, lib, steps):
    def step():
        return model.decode_sdf(fd["qry_norot"], model.encode(fd))
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sample_smi, args=(stop, samples))
    th.start()
    lib.s3d_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    res = {}
    for i, name in enumerate(_lib.PROF_NAMES):
        ms, n = C.c_double(), C.c_long()
        lib.s3d_prof_read(i, C.byref(ms), C.byref(n))
        res[name] = (ms.value / steps, n.value / steps)
    lib.s3d_prof_enable(0)
    stop.set()
    th.join()
    samples = samples[2:] if len(samples) > 4 else samples
    clk = sum(s[0] for s in samples) / max(len(samples), 1)
    pw = sum(s[1] for s in samples) / max(len(samples), 1)
    return dt, res, clk, pw, len(samples)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    lib = _lib.load()
    fd = make_feed_dict(4, 256, 100000, 12, seed=1234, with_slices=False, device="cuda")
    print("| decoder weights | ms/step | FFN ms per launch | attention ms/step | sclk MHz (mean) | socket W (mean) | smi samples |")
    print("|---|---|---|---|---|---|---|")
    for label in ("seeded random (the bench)", "all zero (att_decoder.* = 0)", "seeded random again"):
        model = Slices3DRegModel(img_size=256, n_slices=12, mode="test", prec="f16x3")
        load_seeded(model, 0)
        if label.startswith("all zero"):
            with torch.no_grad():
                for k, p in model.named_parameters():
                    if k.startswith("att_decoder."):
                        p.zero_()
        model.cuda().eval()
        dt, res, clk, pw, n = run(model, fd, lib, steps)
        ffn_ms, ffn_n = res["ffn_layer"]
        print("| %s | %.2f | %.3f | %.2f | %.0f | %.0f | %d |" % (label, dt, ffn_ms / max(ffn_n, 1), res["attn_layer"][0], clk, pw, n))
        del model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
