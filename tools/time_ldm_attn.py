"""Time and error of s3d_qkv_attention_fwd at the LDM's attention shapes (set S3D_LDM_ATTN_F16X3=1 for the split kernel)."""
import os, sys, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slice3d_amd import _lib
lib = _lib.load()
g = torch.Generator().manual_seed(0)
for n, heads, ch, T in ((1, 8, 24, 4096), (4, 8, 24, 4096), (1, 8, 48, 1024), (1, 8, 96, 256)):
    qkv = torch.randn(n, heads * 3 * ch, T, generator=g) * float(os.environ.get("SCALE", 1.0))
    qc = qkv.permute(0, 2, 1).contiguous().cuda()
    q, k, v = qkv[:1].double().reshape(heads, ch * 3, T).split(ch, dim=1)
    sc = 1 / math.sqrt(math.sqrt(ch))
    wgt = torch.softmax(torch.einsum("bct,bcs->bts", q * sc, k * sc), dim=-1)
    want = torch.einsum("bts,bcs->bct", wgt, v).reshape(1, -1, T).permute(0, 2, 1)
    for prec in (0, 1):
        out = torch.empty(n, T, heads * ch, device="cuda")
        run = lambda: _lib.check(lib.s3d_qkv_attention_fwd(qc.data_ptr(), out.data_ptr(), n, T, heads, ch, prec, None), "attn")
        for _ in range(3): run()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(20): run()
        torch.cuda.synchronize(); t = (time.time() - t0) / 20
        err = float((out[:1].cpu().double() - want).abs().max())
        print("N=%d heads=%d ch=%d T=%d prec=%d: %.1f us  err %.2e" % (n, heads, ch, T, prec, t * 1e6, err))
