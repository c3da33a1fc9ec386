"""Row-linear layers on long row sets (the streaming kernel of conv.hip) through s3d_conv_fwd against a float64 product."""
import pytest
import torch

from slice3d_amd import _lib

pytestmark = pytest.mark.gpu


def _run(rows, cin, cout, residual, prec):
    L = _lib.load()
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(rows + cin + cout)
    w = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    x = torch.randn(rows, cin, generator=g).to(dev)
    res = torch.randn(rows, cout, generator=g).to(dev) if residual else None
    out = torch.full((rows + 1, cout), 7.0, device=dev)   # one guard row behind the output
    nb = L.s3d_conv_packed_bytes(cout, cin, 0, 1)
    packed = torch.empty(nb, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.s3d_conv_pack(w.data_ptr(), b.data_ptr(), cout, cin, 0, 1, packed.data_ptr(), nb, st), "conv_pack")
    _lib.check(L.s3d_conv_fwd(packed.data_ptr(), x.data_ptr(), None, res.data_ptr() if residual else None, out.data_ptr(),
                              1, 1, rows, cout, cin, 0, 1, prec, None, 0, st), "conv_fwd")
    torch.cuda.synchronize()
    ref = x.double() @ w.view(cout, cin).double().t() + b.double()
    if residual:
        ref = ref + res.double()
    return (out[:rows].double() - ref).abs().max().item(), out[rows]


@pytest.mark.parametrize("rows,cin,cout,residual", [
    (200_003, 128, 384, False),   # in_proj; the last workgroup task is ragged (rows % 192 = 131)
    (131_072, 128, 128, True),    # out_proj + residual at the kernel's lower row bound
    (150_000, 128, 256, False),
    (140_001, 128, 32, True),     # one 32-channel output slot
])
def test_long_row_linear_matches_float64(rows, cin, cout, residual):
    err, guard = _run(rows, cin, cout, residual, _lib.PREC_F16X3)
    assert err < 2e-5, err            # split precision: fp32-class products, fp32 accumulation over 128 terms
    assert bool((guard == 7.0).all())  # nothing written past the last row
