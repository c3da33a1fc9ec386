import sys, math, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slice3d_amd import _lib
lib = _lib.load()
g = torch.Generator().manual_seed(0)
for heads, ch, T in ((4, 24, 150), (2, 48, 100), (8, 24, 4096), (2, 96, 40)):
    n = 1
    qkv = torch.randn(n, heads * 3 * ch, T, generator=g)
    q, k, v = qkv.double().reshape(n * heads, ch * 3, T).split(ch, dim=1)
    sc = 1 / math.sqrt(math.sqrt(ch))
    wgt = torch.softmax(torch.einsum("bct,bcs->bts", q * sc, k * sc), dim=-1)
    want = torch.einsum("bts,bcs->bct", wgt, v).reshape(n, -1, T).permute(0, 2, 1).contiguous()
    qc = qkv.permute(0, 2, 1).contiguous().cuda()
    for prec in (0, 1):
        out = torch.empty(n, T, heads * ch, device="cuda")
        _lib.check(lib.s3d_qkv_attention_fwd(qc.data_ptr(), out.data_ptr(), n, T, heads, ch, prec, None), "attn")
        print(heads, ch, T, "prec", prec, "max err vs fp64 %.3e" % float((out.cpu().double() - want).abs().max()))

# error pattern of the split-precision kernel on the two bad shapes
for heads, ch, T in ((8, 24, 4096), (2, 96, 40)):
    qkv = torch.randn(1, heads * 3 * ch, T, generator=g)
    q, k, v = qkv.double().reshape(heads, ch * 3, T).split(ch, dim=1)
    sc = 1 / math.sqrt(math.sqrt(ch))
    wgt = torch.softmax(torch.einsum("bct,bcs->bts", q * sc, k * sc), dim=-1)
    want = torch.einsum("bts,bcs->bct", wgt, v).reshape(1, -1, T).permute(0, 2, 1).contiguous()   # [1][T][heads*ch]
    qc = qkv.permute(0, 2, 1).contiguous().cuda()
    out = torch.empty(1, T, heads * ch, device="cuda")
    _lib.check(lib.s3d_qkv_attention_fwd(qc.data_ptr(), out.data_ptr(), 1, T, heads, ch, 1, None), "attn")
    err = (out.cpu().double() - want).abs()[0]            # [T][heads*ch]
    e = err.reshape(T, heads, ch)
    print("shape", heads, ch, T, "max", float(err.max()))
    print("  per head max:", [float("%.1e" % x) for x in e.amax(dim=(0, 2))])
    print("  per channel max:", [float("%.1e" % x) for x in e.amax(dim=(0, 1))])
    tq = e.amax(dim=(1, 2))
    print("  queries with err > 1e-6:", int((tq > 1e-6).sum()), "of", T, " first few:", torch.nonzero(tq > 1e-6)[:12, 0].tolist())
    # which keys matter: relate error to the largest |v| and the attention weight mass
    qi = int(tq.argmax())
    print("  worst query", qi, "its max softmax weight", float(wgt[:, qi].max()), "max |score|", float((torch.einsum('bct,bcs->bts', q*sc, k*sc))[:, qi].abs().max()))

# is the split-precision kernel deterministic run to run?
os.environ["S3D_LDM_ATTN_F16X3"] = "1"
heads, ch, T = 8, 24, 4096
qkv = torch.randn(1, heads * 3 * ch, T, generator=g)
qc = qkv.permute(0, 2, 1).contiguous().cuda()
outs = []
for _ in range(6):
    out = torch.empty(1, T, heads * ch, device="cuda")
    _lib.check(lib.s3d_qkv_attention_fwd(qc.data_ptr(), out.data_ptr(), 1, T, heads, ch, 1, None), "attn")
    outs.append(out.cpu())
print("run-to-run max differences:", [float((o - outs[0]).abs().max()) for o in outs[1:]])
