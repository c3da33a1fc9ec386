"""Per-layer time and TFLOP/s of the U-Net encode (s3d_unet_encode_fwd) from a rocprofv3 kernel trace of
tools/unet_layers_run.py, against the per-layer FLOP table of BASELINE.md section 3 (256^2, 12 slices).

    rocprofv3 --kernel-trace -d /tmp/ul -o u -- python tools/unet_layers_run.py ; python tools/unet_layers.py <db> [B]
The encode launches its layers in a fixed order (api.hip: 13 encoder convs with BN+ReLU+pool after the taps, trans_c,
then trans_up / ConvT / 3x3 / 3x3 per up stage); a split-K layer adds a conv_splitk_finish_kernel right after its conv."""
import sqlite3
import sys

ENC = [("conv1_1 3->64 @256", 0.23), ("conv1_2 64->64 @256", 4.83), ("conv2_1 64->128 @128", 2.42),
       ("conv2_2 128->128 @128", 4.83), ("conv3_1 128->256 @64", 2.42), ("conv3_2 256->256 @64", 4.83),
       ("conv3_3 256->256 @64", 4.83), ("conv4_1 256->512 @32", 2.42), ("conv4_2 512->512 @32", 4.83),
       ("conv4_3 512->512 @32", 4.83), ("conv5_1 512->512 @16", 1.21), ("conv5_2 512->512 @16", 1.21),
       ("conv5_3 512->512 @16", 1.21)]
TAPS = {1, 3, 6, 9}
UP = []
for i, (c, r) in enumerate(((512, 32), (256, 64), (128, 128), (64, 256))):
    # skip 1x1 runs once per image here (the reference runs it on the 12x expanded batch: 3.22 GFLOP)
    UP += [("up%d skip 1x1 %d->%d @%d (per image)" % (i + 1, c, c // 2, r), 3.22 / 12), ("up%d ConvT %d->%d" % (i + 1, c, c // 2), 3.22),
           ("up%d 3x3 %d->%d @%d" % (i + 1, c, c // 2, r), 28.99), ("up%d 3x3 %d->%d @%d" % (i + 1, c // 2, c // 2, r), 14.50)]
LAYERS = ENC + [("trans_c 1x1 (512 per image + slice bias) @16", 2.01 * 512 / 640)] + UP


def main(path, batch):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    conv_like = lambda n: ("conv3x3_lds" in n or "conv_igemm" in n or "lin_rows" in n or "conv3x3_first" in n)
    encodes, cur = [], None
    for name, st, en in rows:
        if "conv3x3_first_kernel" in name:     # the encode's first launch
            cur = [(name, en - st)]
            encodes.append(cur)
            continue
        if cur is None:
            continue
        if conv_like(name) or "conv_splitk_finish" in name or "bn_relu_pool" in name:
            cur.append((name, en - st))
        else:
            cur = None
    acc = [[0.0, 0.0, 0.0, ""] for _ in LAYERS]      # conv us, finish us, pool us
    n_ok = 0
    for e in encodes[2:]:
        i, li, ok = 0, 0, True
        tmp = [[0.0, 0.0, 0.0, ""] for _ in LAYERS]
        while li < len(LAYERS) and i < len(e):
            name, d = e[i]
            if not conv_like(name):
                ok = False
                break
            tmp[li][0] += d / 1e3
            tmp[li][3] = name.split("(")[0].replace("void ", "")[:44]
            i += 1
            if i < len(e) and "conv_splitk_finish" in e[i][0]:
                tmp[li][1] += e[i][1] / 1e3
                i += 1
            if li in TAPS and i < len(e) and "bn_relu_pool" in e[i][0]:
                tmp[li][2] += e[i][1] / 1e3
                i += 1
            li += 1
        if ok and li == len(LAYERS):
            n_ok += 1
            for a, t in zip(acc, tmp):
                a[0] += t[0]; a[1] += t[1]; a[2] += t[2]; a[3] = t[3]
    print("# U-Net encode per layer: %d objects per call, f16x3, averaged over %d encodes (tools/unet_layers.py)\n" % (batch, n_ok))
    print("| layer | GFLOP (x%d objects) | conv us | split-K finish us | BN+ReLU+pool us | TFLOP/s algorithmic | kernel |" % batch)
    print("|---|---|---|---|---|---|---|")
    tot_f = tot_t = 0.0
    for (label, gf), a in zip(LAYERS, acc):
        us = (a[0] + a[1] + a[2]) / max(n_ok, 1)
        f = gf * batch
        tot_f += f; tot_t += us
        print("| %s | %.2f | %.1f | %.1f | %.1f | %.0f | `%s` |" % (label, f, a[0] / max(n_ok, 1), a[1] / max(n_ok, 1),
                                                                     a[2] / max(n_ok, 1), f / us * 1e3 if us else 0, a[3]))
    print("\ntotal %.1f GFLOP in %.1f us of kernel time = %.0f TFLOP/s algorithmic (x3 on the f16 pipe)" % (tot_f, tot_t, tot_f / tot_t * 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4)
