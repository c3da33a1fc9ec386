"""Per-call table of one LDM denoising step (BASELINE configs[4]; reference: openaimodel.py:413-757): every C-ABI launch of one eager step
is recorded (function, arguments), then replayed 20 times back to back between two events — the duration of the call's kernels with
the launch gaps of eager mode amortised.  Prints the calls in step order with the convolutions' shapes, GFLOP and TFLOP/s, and the
sums per (function, map size).  Usage (GPU box): python tools/ldm_layers.py [batch] [latent size] > profiles/rNN_ldm_layers.md"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from helpers import ldm_inputs  # noqa: E402
from test_ldm import LDM_FULL  # noqa: E402
from slice3d_amd.ldm_unet import UNetModel  # noqa: E402
from slice3d_amd.weights import load_seeded  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
SIZE = int(sys.argv[2]) if len(sys.argv) > 2 else 64
REPS = 20


class Recorder:
    def __init__(self, lib):
        self._lib, self.calls, self.on = lib, [], False

    def __getattr__(self, name):
        f = getattr(self._lib, name)
        if not callable(f):
            return f

        def call(*a):
            r = f(*a)
            if self.on and (name.endswith("_fwd") or name in ("s3d_nchw_to_nhwc_pad", "s3d_nhwc_to_nchw")):
                self.calls.append((name, a))
            return r
        return call


cfg = dict(LDM_FULL, image_size=SIZE)
m = load_seeded(UNetModel(**cfg), 0).cuda().eval()
x, t, cf = ldm_inputs(cfg, B, 1)
x, t, cf = x.cuda(), t.cuda(), {k: v.cuda() for k, v in cf.items()}
for _ in range(2):
    y = m(x, t, c_fmaps=cf)
rec = Recorder(m._lib)
m._lib = rec
rec.on = True
y = m(x, t, c_fmaps=cf)
rec.on = False
torch.cuda.synchronize()
lib = rec._lib


def describe(name, a):
    if name in ("s3d_conv_fwd", "s3d_conv_gn_fwd"):
        n, h, w, cout, cin0, cin1, ks = a[5:12]
        gflop = 2.0 * n * h * w * cout * (cin0 + cin1) * ks * ks * 1e-9
        return "%dx%dx%d  %d%s -> %d  k%d" % (n, h, w, cin0, "+%d" % cin1 if cin1 else "", cout, ks), h, gflop
    ints = [v for v in a if isinstance(v, int) and 0 < v < 100000]
    hh = 0
    if name.startswith("s3d_group_norm"):      # (..., n, h * w, channels, groups, ...): the pixel count is the square among them
        hw = next((v for v in ints if v in (16, 64, 256, 1024, 4096, 16384) and v != a[1]), 0)
        hh = int(round(hw ** 0.5))
    if name == "s3d_resample2x_fwd":
        hh = next((v for v in ints if v in (4, 8, 16, 32, 64, 128)), 0)
    if name.startswith("s3d_qkv_attention"):
        tkn = a[3]
        hh = int(round(tkn ** 0.5))
        return "N %d  tokens %d  heads %d  ch %d" % (a[2], tkn, a[4], a[5]), hh, 4.0 * a[2] * tkn * tkn * a[4] * a[5] * 1e-9
    return " ".join(str(v) for v in ints[:6]), hh, 0.0


rows = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, a in rec.calls:
    f = getattr(lib, name)
    f(*a)
    e0.record()
    for _ in range(REPS):
        f(*a)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) / REPS * 1e3
    d, hh, gf = describe(name, a)
    rows.append((name, d, hh, gf, us))

# ---- cold against warm: one call at a time between two events, (a) right after an identical call, (b) after a 1 GiB read-modify-write
#      has swept L2 and the Infinity Cache, (c) convolutions only: after the sweep AND a pass over the call's packed weights — what a
#      weight prefetcher running ahead of the chain would leave behind
flush = torch.zeros(256 << 20, dtype=torch.float32, device="cuda")
wbuf = {}
for v in m._packed.values():
    wbuf[v[0].data_ptr()] = v[0]


def one(f, a, prep):
    ts = []
    for _ in range(5):
        prep()
        e0.record()
        f(*a)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[2]


cold = []
for name, a in rec.calls:
    f = getattr(lib, name)
    t_w = one(f, a, lambda: (torch.cuda._sleep(50000), f(*a)))   # the sleep lets the host queue the timed call behind it
    t_c = one(f, a, lambda: flush.add_(1.0))
    t_p = None
    if name in ("s3d_conv_fwd", "s3d_conv_gn_fwd") and a[0] in wbuf:
        wt = wbuf[a[0]]

        def prep():
            flush.add_(1.0)
            wt.view(torch.int32).sum()
        t_p = one(f, a, prep)
    cold.append((t_w, t_c, t_p))
sw = sum(c[0] for c in cold)
sc = sum(c[1] for c in cold)
sp = sum(c[2] if c[2] is not None else c[1] for c in cold)
convs = [c for c in cold if c[2] is not None]

total = sum(r[4] for r in rows)
print("# LDM denoising step, batch %d, %dx%d latent: every C-ABI launch of one step, replayed %d x back to back (`tools/ldm_layers.py`)\n"
      % (B, SIZE, SIZE, REPS))
print("%d calls, %.0f us in all (the HIP-graph replay of the step is timed by `tools/time_ldm.py` / bench.py).\n" % (len(rows), total))
print("## Cold against warm (one call between two events, median of 5)\n")
print("| state before the call | sum over the step's %d calls, us | the %d convolutions alone, us |\n|---|---|---|" % (len(cold), len(convs)))
print("| an identical call just ran (everything the call touches is cached) | %.0f | %.0f |" % (sw, sum(c[0] for c in convs)))
print("| a 1 GiB read-modify-write just ran (L2 and Infinity Cache swept) | %.0f | %.0f |" % (sc, sum(c[1] for c in convs)))
print("| swept, then the call's packed weights read once (what a weight prefetcher would leave) | %.0f | %.0f |\n" % (sp, sum(c[2] for c in convs)))
print("## Sums per function and map size\n\n| function | map | calls | us | us per call | GFLOP | TFLOP/s |\n|---|---|---|---|---|---|---|")
agg = collections.OrderedDict()
for name, d, hh, gf, us in rows:
    k = (name, hh)
    v = agg.setdefault(k, [0, 0.0, 0.0])
    v[0] += 1
    v[1] += us
    v[2] += gf
for (name, hh), (cnt, us, gf) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("| `%s` | %s | %d | %.0f | %.1f | %s | %s |" % (name, hh or "-", cnt, us, us / cnt, "%.2f" % gf if gf else "-",
                                                     "%.1f" % (gf / us * 1e3) if gf else "-"))
print("\n## Calls in step order\n\n| # | function | arguments | us | GFLOP | TFLOP/s |\n|---|---|---|---|---|---|")
for i, (name, d, hh, gf, us) in enumerate(rows):
    print("| %d | `%s` | %s | %.1f | %s | %s |" % (i, name, d, us, "%.3f" % gf if gf else "-", "%.1f" % (gf / us * 1e3) if gf else "-"))
