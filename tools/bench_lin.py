"""Row-linear layers of the training step as stand-alone launches (s3d_conv_fwd, ks = 1): time and bytes per second.

    python tools/bench_lin.py [rows]          (default 5.2 M rows: 4 objects x 100 k queries x 13 tokens)

Shapes: the attention block's in_proj (128 -> 384), its data gradient (384 -> 128, + residual), out_proj (128 -> 128),
the last layer's K|V gradient (256 -> 128).  Each line also checks the result against a float64 product on 4 096 rows.
"""
import sys
import time

import torch

sys.path.insert(0, ".")
from slice3d_amd import _lib  # noqa: E402

L = _lib.load()
S3D_PREC_F16X3 = _lib.PREC_F16X3


def ptr(t):
    return t.data_ptr() if t is not None else None


def run(rows, cin, cout, residual, reps=10):
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(cin * 1000 + cout)
    w = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).to(dev)
    b = torch.randn(cout, generator=g).to(dev) if cin <= 128 else None   # the K > 128 calls of the step are bias-free
    x = torch.randn(rows, cin, generator=g).to(dev)
    res = torch.randn(rows, cout, generator=g).to(dev) if residual else None
    out = torch.empty(rows, cout, device=dev)
    nb = L.s3d_conv_packed_bytes(cout, cin, 0, 1)
    packed = torch.empty(nb, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    rc = L.s3d_conv_pack(ptr(w), ptr(b), cout, cin, 0, 1, ptr(packed), nb, st)
    _lib.check(rc, "conv_pack")

    def call():
        rc = L.s3d_conv_fwd(ptr(packed), ptr(x), None, ptr(res), ptr(out), 1, 1, rows, cout, cin, 0, 1, S3D_PREC_F16X3,
                            None, 0, st)
        _lib.check(rc, "conv_fwd")

    call()
    torch.cuda.synchronize()
    n = 4096
    ref = x[:n].double() @ w.view(cout, cin).double().t()
    if b is not None:
        ref = ref + b.double()
    if residual:
        ref = ref + res[:n].double()
    err = (out[:n].double() - ref).abs().max().item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gb = rows * 4.0 * (cin + cout * (2 if residual else 1)) / 1e9
    print(f"{cin:4d} -> {cout:4d}{' +res' if residual else '     '}  {ms:7.3f} ms  {gb:6.2f} GB  {gb / ms:6.2f} TB/s"
          f"  max|err| {err:.2e}", flush=True)


if __name__ == "__main__":
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 400000 * 13
    t0 = time.time()
    for cin, cout, res in ((128, 384, False), (384, 128, True), (128, 128, False), (128, 128, True), (256, 128, False),
                           (128, 256, False)):
        run(rows, cin, cout, res)
    print(f"total {time.time() - t0:.1f} s")
