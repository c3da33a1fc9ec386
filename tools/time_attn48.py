"""A/B of the 48-wide attention heads of the gen_slices U-Net: the fp32-MFMA kernel of ldm_ops.hip (s3d_qkv_attention_fwd)
against the key-split f16-MFMA kernel of ldm_attn.hip (s3d_qkv_attention_ws_fwd), per call, HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slice3d_amd import _lib
lib = _lib.load()
print("| N | T | heads | ch | fp32-MFMA kernel us | key-split f16-MFMA (pack + main [+ merge]) us | max abs difference |")
print("|---|---|---|---|---|---|---|")
for n, T, heads, ch in ((1, 1024, 8, 48), (4, 1024, 8, 48), (1, 4096, 8, 48), (1, 4096, 8, 24)):
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(n, T, heads * 3 * ch, generator=g).cuda()
    a = torch.empty(n, T, heads * ch, device="cuda"); b = torch.empty_like(a)
    nb = lib.s3d_qkv_attention_ws_bytes(n, T, heads, ch)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    def f0(): _lib.check(lib.s3d_qkv_attention_fwd(qkv.data_ptr(), a.data_ptr(), n, T, heads, ch, _lib.PREC_F16X3, None), "a")
    def f1(): _lib.check(lib.s3d_qkv_attention_ws_fwd(qkv.data_ptr(), b.data_ptr(), n, T, heads, ch, ws.data_ptr(), nb, None), "b")
    res = []
    for f in (f0, f1):
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 50 * 1e3)
    print("| %d | %d | %d | %d | %.1f | %.1f | %.2e |" % (n, T, heads, ch, res[0], res[1], float((a - b).abs().max())))
