"""Achieved bandwidth of the 1x1-conv (row-linear) path of the conv engine at the decoder's projection shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slice3d_amd import _lib
lib = _lib.load()
rows = int(os.environ.get("ROWS", 1300000))
ws = torch.empty(8 << 20, dtype=torch.float32, device="cuda")
for k, n in ((128, 384), (128, 128), (384, 128), (512, 128), (128, 512), (64, 128)):
    w = torch.randn(n, k, 1, 1, device="cuda") * 0.05
    b = torch.randn(n, device="cuda")
    nb = lib.s3d_conv_packed_bytes(n, k, 0, 1)
    buf = torch.empty(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.s3d_conv_pack(w.data_ptr(), b.data_ptr(), n, k, 0, 1, buf.data_ptr(), nb, None), "pack")
    x = torch.randn(1, 1, rows, k, device="cuda")
    out = torch.empty(1, 1, rows, n, device="cuda")
    for prec in (1, 0):
        def run():
            _lib.check(lib.s3d_conv_fwd(buf.data_ptr(), x.data_ptr(), None, None, out.data_ptr(), 1, 1, rows, n, k, 0, 1,
                                        prec, ws.data_ptr(), ws.numel() * 4, None), "conv")
        for _ in range(3): run()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(10): run()
        torch.cuda.synchronize(); t = (time.time() - t0) / 10
        if prec == 1:
            ref = torch.nn.functional.linear(x[0, 0, :4096].double(), w[:, :, 0, 0].double(), b.double())
            err = float((out[0, 0, :4096].double() - ref).abs().max())
        gb = rows * (k + n) * 4 / 1e9
        print("K=%d N=%d prec=%d: %.3f ms  %.2f TB/s  %.0f TFLOP/s alg  (err %.1e)" %
              (k, n, prec, t * 1e3, gb / t / 1e3, 2.0 * rows * k * n / t / 1e12, err))
