"""TEST INFRASTRUCTURE ONLY — CPU fp32 restatement of the Slice3D regression hot path.

This file is the ORACLE for the HIP path: a from-scratch restatement (torch-CPU fp32 tensor math) of
what the reference computes in
    reg_slices/src/models.py:28-94        (Slices3DRegModel.forward, project_coord, sample_from_planes)
    reg_slices/src/unet_custom.py:40-69   (UNet.forward)
    reg_slices/src/unet_parts.py:8-84     (DoubleConv / Up / OutConv)
    reg_slices/src/vgg_perceptual_loss.py:6-70
    reg_slices/train.py:21-39             (cal_acc / cal_loss_pred)
    reg_slices/reconstruct.py:74-102      (Generator3D.eval_points)
    reg_slices/src_convonet/common.py:145-164 (make_3d_grid)
written from SURVEY.md section 8(a), operating directly on a reference-format `state_dict`.

Pinning: the reference ships no known-answer vectors for this path, so the
oracle is pinned against outputs of the reference itself, imported in the authoring container
(oracle/ref_import.py): tests/test_oracle_vs_reference.py compares live when /root/reference exists,
and tests/golden/*.npz hold the committed vectors (made by tests/golden/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The
product path (slice3d_amd/) never does.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
LN_EPS = 1e-5
N_HEADS = 4
D_MODEL = 128
LEVEL_CHANNELS = (512, 256, 128, 64, 32)  # coarse -> fine, sum = 992 (unet_custom.py:58-67)

# torchvision vgg16_bn.features indices of the 13 convs; BN is idx+1, ReLU idx+2 (unet_custom.py:12-20)
_VGG16_CONV_IDX = (0, 3, 7, 10, 14, 17, 20, 24, 27, 30, 34, 37, 40)
_VGG16_BLOCK_OF = {0: "down1", 3: "down1", 7: "down2", 10: "down2", 14: "down3", 17: "down3",
                   20: "down3", 24: "down4", 27: "down4", 30: "down4", 34: "down5", 37: "down5",
                   40: "down5"}
# which Sequential slice owns the BN that FOLLOWS conv idx (the BN after a tap conv opens the next slice)
_VGG16_BN_OWNER = {0: "down1", 3: "down2", 7: "down2", 10: "down3", 14: "down3", 17: "down3",
                   20: "down4", 24: "down4", 27: "down4", 30: "down5", 34: "down5", 37: "down5",
                   40: "down5_"}
_VGG16_TAPS = (3, 10, 20, 30, 40)       # conv1_2, conv2_2, conv3_3, conv4_3, conv5_3 (pre-BN outputs)
_VGG16_POOL_AFTER_BN_OF = (3, 10, 20, 30)  # maxpool follows BN+ReLU of these convs (40's pool is unused)

# torchvision vgg19.features conv indices and tap positions (vgg_perceptual_loss.py:18-27)
_VGG19_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M",
              512, 512, 512, 512, "M")
_VGG19_TAP_CONVS = (2, 7, 12, 21, 30)
_VGG19_W = (1.0 / 2.6, 1.0 / 4.8, 1.0 / 3.7, 1.0 / 5.6, 10.0 / 1.5)


class TrainState:
    """Train-mode switches for the restatement: batch-statistic BatchNorm (with the running-stat update
    torch performs, momentum 0.1, unbiased variance) and dropout probability (the reference trains with
    nn.TransformerEncoderLayer's default 0.1; parity runs pin it to 0).  `new_stats` collects the
    updated running statistics keyed like the state_dict."""

    def __init__(self, dropout=0.0):
        self.dropout = dropout
        self.new_stats = {}


def _bn(x, sd, prefix, train=None):
    if train is None:
        return _bn_eval(x, sd, prefix)
    n = x.numel() // x.shape[1]
    mu = x.mean(dim=(0, 2, 3))
    var = x.var(dim=(0, 2, 3), unbiased=False)
    with torch.no_grad():
        train.new_stats[prefix + ".running_mean"] = 0.9 * sd[prefix + ".running_mean"] + 0.1 * mu
        train.new_stats[prefix + ".running_var"] = 0.9 * sd[prefix + ".running_var"] + 0.1 * var * n / max(n - 1, 1)
    xh = (x - mu.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + BN_EPS)
    return xh * sd[prefix + ".weight"].view(1, -1, 1, 1) + sd[prefix + ".bias"].view(1, -1, 1, 1)


def _bn_eval(x, sd, prefix):
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    m, v = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    scale = w / torch.sqrt(v + BN_EPS)
    shift = b - m * scale
    return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


# --------------------------------------------------------------------------------------------------
# U-Net  (unet_custom.py:40-69)
# --------------------------------------------------------------------------------------------------
def unet_encoder(sd, x, pfx="slices_generator.", train=None):
    """VGG16-BN encoder; returns the five PRE-BN tap tensors x1..x5."""
    taps = []
    h = x
    for ci in _VGG16_CONV_IDX:
        blk = _VGG16_BLOCK_OF[ci]
        h = F.conv2d(h, sd[f"{pfx}{blk}.{ci}.weight"], sd[f"{pfx}{blk}.{ci}.bias"], padding=1)
        if ci in _VGG16_TAPS:
            taps.append(h)
        if ci == 40:
            break  # down5_ (BN-ReLU-pool) output is never used (unet_custom.py:48)
        h = torch.relu(_bn(h, sd, f"{pfx}{_VGG16_BN_OWNER[ci]}.{ci + 1}", train))
        if ci in _VGG16_POOL_AFTER_BN_OF:
            h = F.max_pool2d(h, 2, 2)
    return taps


def _expand_bs(x, n_slices):
    b, c, h, w = x.shape
    return x.view(b, 1, c, h, w).expand(-1, n_slices, -1, -1, -1).reshape(b * n_slices, c, h, w)


def _double_conv(sd, x, prefix, train=None):
    h = F.conv2d(x, sd[prefix + ".0.weight"], None, padding=1)
    h = torch.relu(_bn(h, sd, prefix + ".1", train))
    h = F.conv2d(h, sd[prefix + ".3.weight"], None, padding=1)
    return torch.relu(_bn(h, sd, prefix + ".4", train))


def _up(sd, x1, x2, prefix, train=None):
    """Up.forward (unet_parts.py:55-75): ConvT 2x2 s2, pad to skip size, cat [skip, up], DoubleConv."""
    x1 = F.conv_transpose2d(x1, sd[prefix + ".up.weight"], sd[prefix + ".up.bias"], stride=2)
    dy, dx = x2.shape[2] - x1.shape[2], x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return _double_conv(sd, torch.cat([x2, x1], dim=1), prefix + ".conv.double_conv", train)


def unet_forward(sd, x, n_slices=12, pfx="slices_generator.", train=None):
    """-> (feats[5] each (B*n_slices, C_l, H_l, W_l) NCHW, slices_rec (B*n_slices, 3, S, S))."""
    x1, x2, x3, x4, x5 = unet_encoder(sd, x, pfx, train)
    b, _, h5, w5 = x5.shape
    emb = sd[pfx + "emds.weight"]  # (n_slices, 128)
    emb_tile = emb.view(1, n_slices, -1, 1, 1).expand(b, n_slices, emb.shape[1], h5, w5)
    emb_tile = emb_tile.reshape(b * n_slices, emb.shape[1], h5, w5)
    latent = torch.cat([_expand_bs(x5, n_slices), emb_tile], 1)
    latent = F.conv2d(latent, sd[pfx + "trans_c.weight"], sd[pfx + "trans_c.bias"])
    feats = [latent]
    h = latent
    for i, skip in zip((1, 2, 3, 4), (x4, x3, x2, x1)):
        proj = F.conv2d(_expand_bs(skip, n_slices), sd[f"{pfx}trans_up{i}.weight"],
                        sd[f"{pfx}trans_up{i}.bias"])
        h = _up(sd, h, proj, f"{pfx}up{i}", train)
        feats.append(h)
    out = torch.tanh(F.conv2d(h, sd[pfx + "outc.conv.weight"], sd[pfx + "outc.conv.bias"]))
    return feats, out


# --------------------------------------------------------------------------------------------------
# per-query path  (models.py:28-84)
# --------------------------------------------------------------------------------------------------
def project_coord(coords, trans_mat_wo_rot_tp):
    """models.py:28-36: [x y z 1] @ T(4x3); uv = XY / Z; 2(uv-.5); clamp [-1,1]."""
    ones = torch.ones(coords.shape[0], coords.shape[1], 1, dtype=coords.dtype)
    pc = torch.bmm(torch.cat([coords, ones], -1), trans_mat_wo_rot_tp)
    uv = pc[:, :, :2] / pc[:, :, 2:]
    return torch.clamp(2 * (uv - 0.5), min=-1, max=1)


def sample_from_planes(planes, coords):
    """models.py:38-46: bilinear grid_sample (zeros padding, align_corners=True).
    planes (N,C,H,W), coords (N,M,2) -> (N,1,M,C)."""
    n, c, _, _ = planes.shape
    out = F.grid_sample(planes, coords.unsqueeze(1).to(planes.dtype), mode="bilinear", padding_mode="zeros",
                        align_corners=True)
    return out.permute(0, 3, 2, 1).reshape(n, 1, coords.shape[1], c)


def bilinear_sample_manual(planes, coords):
    """Independent gather-based restatement of sample_from_planes, (N,C,H,W),(N,M,2) -> (N,M,C).
    ix=(gx+1)/2*(W-1), iy=(gy+1)/2*(H-1); 4 taps from floor; out-of-range taps contribute 0."""
    n, c, h, w = planes.shape
    gx, gy = coords[..., 0], coords[..., 1]
    ix = (gx + 1) * 0.5 * (w - 1)
    iy = (gy + 1) * 0.5 * (h - 1)
    x0, y0 = torch.floor(ix), torch.floor(iy)
    fx, fy = ix - x0, iy - y0
    flat = planes.permute(0, 2, 3, 1).reshape(n, h * w, c)
    out = torch.zeros(n, coords.shape[1], c, dtype=planes.dtype)
    for dy, dx, wt in ((0, 0, (1 - fx) * (1 - fy)), (0, 1, fx * (1 - fy)),
                       (1, 0, (1 - fx) * fy), (1, 1, fx * fy)):
        xi, yi = (x0 + dx).long(), (y0 + dy).long()
        ok = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
        idx = (yi.clamp(0, h - 1) * w + xi.clamp(0, w - 1))
        g = torch.gather(flat, 1, idx.unsqueeze(-1).expand(-1, -1, c))
        out = out + g * (wt * ok).unsqueeze(-1)
    return out


def layer_norm(x, w, b):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + LN_EPS) * w + b


def transformer_layer(sd, x, prefix, dropout=0.0, masks=None):
    """One post-LN nn.TransformerEncoderLayer(d=128, nhead=4, ffn=2048, relu), eval mode
    (models.py:18-19).  x: (R, L, 128)."""
    r, l, d = x.shape
    hd = d // N_HEADS
    qkv = x @ sd[prefix + ".self_attn.in_proj_weight"].t() + sd[prefix + ".self_attn.in_proj_bias"]
    q, k, v = qkv.split(d, dim=-1)
    q = q.view(r, l, N_HEADS, hd).transpose(1, 2)
    k = k.view(r, l, N_HEADS, hd).transpose(1, 2)
    v = v.view(r, l, N_HEADS, hd).transpose(1, 2)
    att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
    # the four dropout sites of nn.TransformerEncoderLayer; `masks` (multipliers 0 or 1/(1-p)) replaces
    # torch's RNG so a run can be compared with an implementation that draws its own masks
    drop = (lambda t, key: t * masks[key]) if masks is not None else \
           (lambda t, key: F.dropout(t, dropout, training=dropout > 0))
    att = drop(att, "att")                                         # MHA dropout on attention weights
    o = (att @ v).transpose(1, 2).reshape(r, l, d)
    o = o @ sd[prefix + ".self_attn.out_proj.weight"].t() + sd[prefix + ".self_attn.out_proj.bias"]
    x = layer_norm(x + drop(o, "o"), sd[prefix + ".norm1.weight"], sd[prefix + ".norm1.bias"])
    hdn = torch.relu(x @ sd[prefix + ".linear1.weight"].t() + sd[prefix + ".linear1.bias"])
    hdn = drop(hdn, "h")
    f = hdn @ sd[prefix + ".linear2.weight"].t() + sd[prefix + ".linear2.bias"]
    return layer_norm(x + drop(f, "f"), sd[prefix + ".norm2.weight"], sd[prefix + ".norm2.bias"])


def sample_pyramid(feats, img_pts, n_slices):
    """models.py:69-78: 5x bilinear sample + concat -> (B*Q, n_slices, 992)."""
    b, q, _ = img_pts.shape
    pts = img_pts.view(b, 1, q, 2).expand(-1, n_slices, -1, -1).reshape(b * n_slices, q, 2)
    interp = [sample_from_planes(f, pts).squeeze(1) for f in feats]
    agg = torch.cat(interp, dim=2)  # (B*n_slices, Q, 992)
    c = agg.shape[-1]
    return agg.view(b, n_slices, q, c).permute(0, 2, 1, 3).reshape(b * q, n_slices, c)


def decode_tokens(sd, tokens_slices, qry_rot, return_layers=False, dropout=0.0, masks=None):
    """fc_p / fc_s / 3-layer transformer / fc_out (models.py:79-84).
    tokens_slices: (B*Q, n_slices, 992) sampled features; qry_rot (B,Q,3) -> sdf (B,Q)."""
    b, q, _ = qry_rot.shape
    feat_qry = qry_rot @ sd["fc_p.weight"].t() + sd["fc_p.bias"]
    feat_slice = tokens_slices @ sd["fc_s.weight"].t() + sd["fc_s.bias"]
    x = torch.cat([feat_qry.view(b * q, 1, D_MODEL), feat_slice], 1)
    layers = [x]
    for i in range(3):
        x = transformer_layer(sd, x, f"att_decoder.layers.{i}", dropout, masks[i] if masks is not None else None)
        layers.append(x)
    tok0 = x[:, 0, :].view(b, q, D_MODEL)
    sdf = (tok0 @ sd["fc_out.0.weight"].t() + sd["fc_out.0.bias"]).squeeze(-1)
    if return_layers:
        return sdf, layers
    return sdf


def rotate_queries(feed_dict, mode):
    """models.py:53-60.  'test': negate y,z, no rotation (on a copy); else qry @ obj_rot_mat."""
    qry = feed_dict["qry_norot"]
    if mode == "test":
        qry = qry.clone()
        qry[:, :, 1:] *= -1
        return qry
    return torch.bmm(qry, feed_dict["obj_rot_mat"])


def decode_points(sd, feats, qry_rot, trans_mat, n_slices, chunk=4096):
    """Per-query path given the pyramid; chunked over queries to bound CPU memory."""
    outs = []
    for s in range(0, qry_rot.shape[1], chunk):
        qr = qry_rot[:, s:s + chunk]
        img_pts = project_coord(qr, trans_mat)
        tok = sample_pyramid(feats, img_pts, n_slices)
        outs.append(decode_tokens(sd, tok, qr))
    return torch.cat(outs, 1)


# --------------------------------------------------------------------------------------------------
# VGG19 perceptual loss  (vgg_perceptual_loss.py:42-70)
# --------------------------------------------------------------------------------------------------
def vgg19_taps(sd, img, pfx="vggptlossfunc.vgg."):
    """Taps as the reference actually sees them: torchvision's in-place ReLU overwrites taps 1-4, so
    they are POST-ReLU; tap 5 (conv5_2) stays pre-ReLU."""
    taps, h, idx = [], img, 0
    slice_of = lambda i: 1 if i < 3 else 2 if i < 8 else 3 if i < 13 else 4 if i < 22 else 5
    for v in _VGG19_CFG:
        if idx > _VGG19_TAP_CONVS[-1]:
            break
        if v == "M":
            h = F.max_pool2d(h, 2, 2)
            idx += 1
            continue
        s = slice_of(idx)
        h = F.conv2d(h, sd[f"{pfx}slice{s}.{idx}.weight"], sd[f"{pfx}slice{s}.{idx}.bias"], padding=1)
        if idx == _VGG19_TAP_CONVS[-1]:
            taps.append(h)
            break
        h = torch.relu(h)
        if idx in _VGG19_TAP_CONVS:
            taps.append(h)
        idx += 2
    return taps


def vgg_perceptual_loss(sd, input_img, target_img):
    mean = sd["vggptlossfunc.mean"]
    std = sd["vggptlossfunc.std"]
    a = ((input_img + 1) / 2.0 - mean) / std
    b = ((target_img + 1) / 2.0 - mean) / std
    xa, xb = vgg19_taps(sd, a), vgg19_taps(sd, b)
    return sum(w * F.l1_loss(p, t) for w, p, t in zip(_VGG19_W, xa, xb))


# --------------------------------------------------------------------------------------------------
# whole forward, losses, eval_points, grid
# --------------------------------------------------------------------------------------------------
@torch.no_grad()
def forward(sd, feed_dict, mode="train", n_slices=12, with_vgg=True):
    """Slices3DRegModel.forward in eval mode (models.py:48-94)."""
    img = feed_dict["img_input"]
    b, _, s1, s2 = img.shape
    qry_rot = rotate_queries(feed_dict, mode)
    feats, slices_rec = unet_forward(sd, img, n_slices)
    sdf = decode_points(sd, feats, qry_rot, feed_dict["trans_mat_wo_rot_tp"], n_slices)
    ret = {"sdf_pred": sdf, "slices_rec": slices_rec.view(b, n_slices * 3, s1, s2)}
    if with_vgg:
        tgt = feed_dict["img_slices"].view(b * n_slices, 3, s1, s2)
        ret["vgg_loss"] = vgg_perceptual_loss(sd, slices_rec, tgt) * 0.001
    return ret


def forward_train(sd, feed_dict, n_slices=12, dropout=0.0, masks=None):
    """Slices3DRegModel.forward in TRAIN mode (batch-stat BN, dropout) + the losses of train.py:41-47,
    differentiable w.r.t. the tensors of `sd` that require grad.  Returns (loss, parts, out, TrainState)."""
    ts = TrainState(dropout)
    img = feed_dict["img_input"]
    b, _, s1, s2 = img.shape
    qry_rot = torch.bmm(feed_dict["qry_norot"], feed_dict["obj_rot_mat"])
    feats, slices_rec = unet_forward(sd, img, n_slices, train=ts)
    img_pts = project_coord(qry_rot, feed_dict["trans_mat_wo_rot_tp"])
    tok = sample_pyramid(feats, img_pts, n_slices)
    sdf = decode_tokens(sd, tok, qry_rot, dropout=dropout, masks=masks)
    tgt = feed_dict["img_slices"].view(b * n_slices, 3, s1, s2)
    out = {"sdf_pred": sdf, "slices_rec": slices_rec.view(b, n_slices * 3, s1, s2),
           "vgg_loss": vgg_perceptual_loss(sd, slices_rec, tgt) * 0.001}
    lp, li, lv = cal_loss_pred(out, feed_dict)
    return lp + li + lv, (lp, li, lv), out, ts


def adam_step(p, g, m, v, step, lr=3e-4, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam (no weight decay, no amsgrad) single-tensor update; step counts from 1."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    denom = torch.sqrt(v) / math.sqrt(1 - b2 ** step) + eps
    return p - (lr / (1 - b1 ** step)) * m / denom, m, v


def cal_loss_pred(x, gt):
    """train.py:29-39 (sdf branch)."""
    return (F.l1_loss(x["sdf_pred"], gt["sdf"]), F.l1_loss(x["slices_rec"], gt["img_slices"]),
            x["vgg_loss"])


def cal_acc(x, gt):
    """train.py:21-27 (sdf branch)."""
    acc = ((x["sdf_pred"] >= 0) == (gt["sdf"] >= 0)).float().sum(dim=-1) / x["sdf_pred"].shape[1]
    return acc.mean(-1)


def make_3d_grid(bb_min, bb_max, shape):
    """src_convonet/common.py:145-164: x slowest, z fastest, (n^3, 3)."""
    xs = torch.linspace(bb_min[0], bb_max[0], shape[0])
    ys = torch.linspace(bb_min[1], bb_max[1], shape[1])
    zs = torch.linspace(bb_min[2], bb_max[2], shape[2])
    g = torch.stack(torch.meshgrid(xs, ys, zs, indexing="ij"), dim=-1)
    return g.reshape(-1, 3)


@torch.no_grad()
def eval_points(sd, data, n_slices=12, chunk_size=3000):
    """Generator3D.eval_points (reconstruct.py:74-102) on a mode='test' model: -sdf_pred, (Q,).
    The U-Net is evaluated once (its output does not depend on the chunk)."""
    feats, _ = unet_forward(sd, data["img_input"], n_slices)
    qry = rotate_queries(data, "test")
    sdf = decode_points(sd, feats, qry, data["trans_mat_wo_rot_tp"], n_slices, chunk=chunk_size)
    return (-sdf).squeeze(0)


# --------------------------------------------------------------------------------------------------
# Slices3DGTModel  (model_gt.py:59-111, vgg16bn_feats.py:42-57)
# --------------------------------------------------------------------------------------------------
_GT_BLOCK = {"down1": "conv1_2", "down2": "conv2_2", "down3": "conv3_3", "down4": "conv4_3", "down5": "conv5_3",
             "down5_": "conv_last"}
GT_LEVEL_CHANNELS = (64, 128, 256, 512, 512)   # conv1_2 ... conv5_3, sum = 1472


def gt_encoder(sd, x, pfx="img_encoder.", train=None):
    """VGG16BNFeats.forward (vgg16bn_feats.py:42-57): the five RAW conv outputs conv1_2..conv5_3 — the slices
    [:4] [4:11] [11:21] [21:31] [31:41] end on a conv, the BN/ReLU/pool that follow open the next slice.
    train: a TrainState -> batch-statistics BatchNorm (running-stat updates collected in train.new_stats)."""
    taps = []
    h = x
    for ci in _VGG16_CONV_IDX:
        blk = _GT_BLOCK[_VGG16_BLOCK_OF[ci]]
        h = F.conv2d(h, sd[f"{pfx}{blk}.{ci}.weight"], sd[f"{pfx}{blk}.{ci}.bias"], padding=1)
        if ci in _VGG16_TAPS:
            taps.append(h)
        if ci == 40:
            # conv_last (BN-ReLU-pool) only feeds feat_global, which forward() never uses; in train mode its
            # BatchNorm still sees the batch, so its running statistics move
            if train is not None:
                with torch.no_grad():
                    _bn(h, sd, f"{pfx}conv_last.41", train)
            break
        h = torch.relu(_bn(h, sd, f"{pfx}{_GT_BLOCK[_VGG16_BN_OWNER[ci]]}.{ci + 1}", train))
        if ci in _VGG16_POOL_AFTER_BN_OF:
            h = F.max_pool2d(h, 2, 2)
    return taps


def gt_decode_points(sd, feats, qry_rot, trans_mat, n_slices, chunk=2048, dropout=0.0, masks=None):
    """model_gt.py:77-106 for a set of (already rotated / flipped) queries.  dropout / masks: train-mode
    transformer (see transformer_layer); masks index all queries, so they need chunk >= Q."""
    outs = []
    b = qry_rot.shape[0]
    for s in range(0, qry_rot.shape[1], chunk):
        qr = qry_rot[:, s:s + chunk]
        q = qr.shape[1]
        img_pts = project_coord(qr, trans_mat)
        pts = img_pts.view(b, 1, q, 2).expand(-1, n_slices, -1, -1).reshape(b * n_slices, q, 2)
        agg = torch.cat([sample_from_planes(f, pts).squeeze(1) for f in feats], dim=2)     # (B*ns, Q, 1472)
        agg = agg.view(b, n_slices, q, 1472).permute(0, 2, 1, 3).reshape(b, q, n_slices, 1472)
        h = qr
        for i in (0, 2, 4):
            h = torch.relu(h @ sd[f"pts_feat_extractor.{i}.weight"].t() + sd[f"pts_feat_extractor.{i}.bias"])
        loc = agg
        for i in (0, 2):
            loc = torch.relu(loc @ sd[f"fc_local.{i}.weight"].t() + sd[f"fc_local.{i}.bias"])
        x = torch.cat([h.reshape(b * q, 1, D_MODEL), loc.reshape(b * q, n_slices, D_MODEL)], 1)
        for i in range(3):
            x = transformer_layer(sd, x, f"att_decoder.layers.{i}", dropout, masks[i] if masks is not None else None)
        tok0 = x[:, 0, :].view(b, q, D_MODEL)
        outs.append((tok0 @ sd["fc_out.0.weight"].t() + sd["fc_out.0.bias"]).squeeze(-1))
    return torch.cat(outs, 1)


def gt_forward(sd, feed_dict, mode="train", n_slices=12):
    """-> (sdf_pred (B,Q), feats[5] NCHW)."""
    sl = feed_dict["img_slices"]
    b, _, hh, ww = sl.shape
    feats = gt_encoder(sd, sl.reshape(b * n_slices, 3, hh, ww))
    qry_rot = rotate_queries(feed_dict, mode)
    return gt_decode_points(sd, feats, qry_rot, feed_dict["trans_mat_wo_rot_tp"], n_slices), feats


def gt_forward_train(sd, feed_dict, n_slices=12, dropout=0.0, masks=None):
    """Slices3DGTModel.forward in TRAIN mode (model_gt.py:59-111: batch-stat BN over the B*n_slices slice images,
    transformer dropout) + the loss / accuracy of train_gt.py:21-36, differentiable w.r.t. the tensors of `sd`
    that require grad.  Returns (loss, acc, sdf_pred, TrainState)."""
    ts = TrainState(dropout)
    sl = feed_dict["img_slices"]
    b, _, hh, ww = sl.shape
    feats = gt_encoder(sd, sl.reshape(b * n_slices, 3, hh, ww), train=ts)
    qry_rot = torch.bmm(feed_dict["qry_norot"], feed_dict["obj_rot_mat"])
    q = qry_rot.shape[1]
    sdf = gt_decode_points(sd, feats, qry_rot, feed_dict["trans_mat_wo_rot_tp"], n_slices, chunk=q,
                           dropout=dropout, masks=masks)
    loss = F.l1_loss(sdf, feed_dict["sdf"])
    acc = (((sdf >= 0) == (feed_dict["sdf"] >= 0)).float().sum(dim=-1) / q).mean()
    return loss, acc, sdf, ts
