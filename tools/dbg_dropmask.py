import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from slice3d_amd import _lib
lib = _lib.load()
def hash32(seed, site, idx):
    M = 0xFFFFFFFF
    x = (idx & M) ^ (((idx >> 32) * 0x9E3779B1) & M) ^ (seed & M) ^ (((seed >> 32) * 0x85EBCA77) & M) ^ (((site + 1) * 0xC2B2AE3D) & M)
    x ^= x >> 16; x = (x * 0x85EBCA6B) & M; x ^= x >> 13; x = (x * 0xC2B2AE35) & M; x ^= x >> 16; x = (x * 0x9E3779B1) & M; x ^= x >> 15
    return x
def drop(seed, site, idx, p):
    h = hash32(seed, site, idx >> 2)
    if idx & 2:
        h ^= h >> 15; h = (h * 0x2C1B3C6D) & 0xFFFFFFFF; h ^= h >> 12
    f = (h >> 16) if (idx & 1) else (h & 0xFFFF)
    return 1.0 if f >= int(p * 65536 + 0.5) else 0.0
seed, site, p = 987654321987, 6, 0.1
for idx0, n in ((0, 64), (5, 37), (1 << 33, 16)):
    out = torch.empty(n, device="cuda")
    _lib.check(lib.s3d_dropout_mask(seed, site, idx0, n, p, out.data_ptr(), None), "mask")
    got = (out.cpu().numpy() > 0).astype(float)
    want = np.array([drop(seed, site, idx0 + i, p) for i in range(n)])
    print(idx0, n, "mismatch", int((got != want).sum()), "keep", got.mean())
