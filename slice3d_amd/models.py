"""Slices3DRegModel — MI355X-native mirror of the reference module API
(reference: reg_slices/src/models.py:12-94, unet_custom.py:4-69, unet_parts.py:8-84,
vgg_perceptual_loss.py:6-70).

The nn.Module tree exists ONLY to own parameters under the reference's exact state_dict keys (244
tensors, incl. the dead `att_layer.*` twin and `slices_generator.down5_.*`), so released checkpoints
load with `load_state_dict(strict=True)`.  No torch op computes anything on the hot path: forward /
encode / decode hand raw device pointers to libslice3d_hip.so (include/slice3d_hip.h), and raise if the
library is missing.  PyTorch supplies device memory, the stream and (later) autograd plumbing.

API kept from the reference:
    Slices3DRegModel(img_size=128, n_slices=12, mode='train')
    forward(feed_dict) -> {'sdf_pred' (B,Q), 'slices_rec' (B,3*n_slices,S,S), 'vgg_loss' ()}
    project_coord(coords, trans) ; sample_from_planes(planes, coords) ; slices_generator(x)
added (ConvONet-style split the reference's Generator3D half-expects, reconstruct.py:260,312):
    encode(feed_dict) -> LatentCode ;  decode(p, c) -> obj with .logits (= -sdf) and .sdf
"""
import ctypes as C
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import _lib

LEVEL_CHANNELS = (512, 256, 128, 64, 32)

# torchvision vgg16_bn.features: conv indices, grouped as the reference slices them
# (unet_custom.py:12-20: [:4] [4:11] [11:21] [21:31] [31:41] [41:44]); BN follows each conv at idx+1.
_VGG16_CFG = ((0, 3, 64), (3, 64, 64), (7, 64, 128), (10, 128, 128), (14, 128, 256), (17, 256, 256),
              (20, 256, 256), (24, 256, 512), (27, 512, 512), (30, 512, 512), (34, 512, 512),
              (37, 512, 512), (40, 512, 512))
_VGG16_SLICES = (("down1", 0, 4), ("down2", 4, 11), ("down3", 11, 21), ("down4", 21, 31),
                 ("down5", 31, 41), ("down5_", 41, 44))
# torchvision vgg19.features conv indices up to conv5_2, sliced as vgg_perceptual_loss.py:18-27
_VGG19_CONVS = ((0, 3, 64), (2, 64, 64), (5, 64, 128), (7, 128, 128), (10, 128, 256), (12, 256, 256),
                (14, 256, 256), (16, 256, 256), (19, 256, 512), (21, 512, 512), (23, 512, 512),
                (25, 512, 512), (28, 512, 512), (30, 512, 512))
_VGG19_SLICES = (("slice1", 0, 3), ("slice2", 3, 8), ("slice3", 8, 13), ("slice4", 13, 22),
                 ("slice5", 22, 31))


def _slice_of(idx, table):
    for name, lo, hi in table:
        if lo <= idx < hi:
            return name
    raise KeyError(idx)


class DoubleConv(nn.Module):
    """Parameter container for unet_parts.py:8-26 (conv3x3-BN-ReLU x2, convs without bias)."""

    def __init__(self, cin, cout):
        super().__init__()
        seq = nn.Sequential()
        seq.add_module("0", nn.Conv2d(cin, cout, 3, padding=1, bias=False))
        seq.add_module("1", nn.BatchNorm2d(cout))
        seq.add_module("3", nn.Conv2d(cout, cout, 3, padding=1, bias=False))
        seq.add_module("4", nn.BatchNorm2d(cout))
        self.double_conv = seq


class Up(nn.Module):
    """Parameter container for unet_parts.py:42-75 (ConvTranspose2d 2x2 s2 + DoubleConv)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.up = nn.ConvTranspose2d(cin, cin // 2, kernel_size=2, stride=2)
        self.conv = DoubleConv(cin, cout)


class OutConv(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size=1)


class UNet(nn.Module):
    """Slice generator (unet_custom.py:4-69).  `n_slices` is a ctor argument here (the reference
    hard-codes 12, unet_custom.py:9); calling the module runs the HIP encoder and returns the
    reference's (feats[5] NCHW, slices_rec (B*n_slices,3,S,S))."""

    def __init__(self, n_channels=3, n_slices=12):
        super().__init__()
        self.n_channels = n_channels
        self.n_slices = n_slices
        self.dim_embed = 128
        seqs = {name: nn.Sequential() for name, _, _ in _VGG16_SLICES}
        for idx, cin, cout in _VGG16_CFG:
            seqs[_slice_of(idx, _VGG16_SLICES)].add_module(str(idx), nn.Conv2d(cin, cout, 3, padding=1))
            seqs[_slice_of(idx + 1, _VGG16_SLICES)].add_module(str(idx + 1), nn.BatchNorm2d(cout))
        for name, _, _ in _VGG16_SLICES:
            setattr(self, name, seqs[name])
        self.trans_c = nn.Conv2d(512 + self.dim_embed, 512, 1)
        self.up1 = Up(512, 256)
        self.trans_up1 = nn.Conv2d(512, 256, 1)
        self.up2 = Up(256, 128)
        self.trans_up2 = nn.Conv2d(256, 128, 1)
        self.up3 = Up(128, 64)
        self.trans_up3 = nn.Conv2d(128, 64, 1)
        self.up4 = Up(64, 32)
        self.trans_up4 = nn.Conv2d(64, 32, 1)
        self.outc = OutConv(32, 3)
        self.emds = nn.Embedding(self.n_slices, self.dim_embed)
        self._owner = None  # set by Slices3DRegModel (not a submodule: avoid a reference cycle in state_dict)

    def forward(self, x):
        owner = self._owner() if self._owner is not None else None
        if owner is None:
            raise _lib.S3dError("UNet must be owned by a Slices3DRegModel to run")
        code = owner.encode({"img_input": x}, want_slices=True, build_latent=False)
        feats = [owner._nhwc_to_nchw(f) for f in code.pyramid]
        return feats, code.slices_rec_flat


class VGG19Feats(nn.Module):
    """Parameter container for vgg_perceptual_loss.py:6-40 (frozen VGG19 features[:31])."""

    def __init__(self):
        super().__init__()
        seqs = {name: nn.Sequential() for name, _, _ in _VGG19_SLICES}
        for idx, cin, cout in _VGG19_CONVS:
            seqs[_slice_of(idx, _VGG19_SLICES)].add_module(str(idx), nn.Conv2d(cin, cout, 3, padding=1))
        for name, _, _ in _VGG19_SLICES:
            setattr(self, name, seqs[name])
        for p in self.parameters():
            p.requires_grad = False


class VGGPerceptualLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.vgg = VGG19Feats()
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))
        self.weights = [1.0 / 2.6, 1.0 / 4.8, 1.0 / 3.7, 1.0 / 5.6, 1.0 * 10 / 1.5]


class LatentCode:
    """Result of encode(): the per-object state every query decode needs (the 'c' of ConvONet's
    encode/decode split).  All tensors are fp32 on the model's device, channels-last."""

    def __init__(self):
        self.pyramid = None       # 5 x (B*ns, H_l, W_l, C_l)
        self.proj = None          # 3 x (B*ns, H_l, W_l, 128): levels 0-2 with fc_s folded in
        self.slices_rec_flat = None  # (B*ns, 3, S, S) or None
        self.batch = 0
        self.size = 0
        self.n_slices = 0
        self.obj_rot_mat = None
        self.trans_mat_wo_rot_tp = None
        self._latent_struct = None

    @property
    def slices_rec(self):
        if self.slices_rec_flat is None:
            return None
        s = self.size
        return self.slices_rec_flat.view(self.batch, self.n_slices * 3, s, s)


class _TrainForward(torch.autograd.Function):
    """Train-mode `model(batch)` for autograd (the reference's contract, train.py:41-53: `x = model(batch)`,
    `loss(x).backward()`, `opt.step()`): forward = s3d_train_fwd, backward = s3d_train_bwd on the engine's
    workspace.  The parameters are inputs only so that autograd routes their gradients; the gradients come back as
    copies of the engine's flat buffer (param.grad then accumulates like any torch gradient)."""

    @staticmethod
    def forward(ctx, engine, batch, *params):
        sdf, rec, vgg, tctx = engine.forward_only(batch)
        ctx.engine, ctx.tctx = engine, tctx
        ctx.n_params = len(params)
        return sdf, rec, vgg

    @staticmethod
    def backward(ctx, d_sdf, d_rec, d_vgg):
        eng = ctx.engine
        if eng._last_ctx is not ctx.tctx:
            raise RuntimeError("backward through an older train-mode forward: the activations of this model's "
                               "workspace belong to the most recent model(batch) call")
        flat = eng.backward_from(ctx.tctx, d_sdf, d_rec, float(d_vgg) if d_vgg is not None else 0.0,
                                 grad_scale=eng.auto_grad_scale(d_sdf))
        grads = []
        for k, p in zip(eng.names, eng.params):
            off = eng.offsets[k]
            grads.append(flat[off:off + p.numel()].view_as(p).clone())
        return (None, None) + tuple(grads)


class Slices3DRegModel(nn.Module):
    def __init__(self, img_size=128, n_slices=12, mode="train", backend="hip", prec="f32"):
        super().__init__()
        self.mode = mode
        self.slices_generator = UNet(n_channels=3, n_slices=n_slices)
        self.img_size = img_size
        self.att_layer = nn.TransformerEncoderLayer(d_model=128, nhead=4, batch_first=True)
        self.att_decoder = nn.TransformerEncoder(self.att_layer, num_layers=3)  # deep-copies the layer
        self.fc_p = nn.Linear(3, 128)
        self.fc_s = nn.Linear(992, 128)
        self.fc_out = nn.Sequential(nn.Linear(128, 1))
        self.vggptlossfunc = VGGPerceptualLoss()
        self.n_slices = n_slices
        self.backend = backend
        self.prec = {"f32": _lib.PREC_F32, "f16x3": _lib.PREC_F16X3, "f16": _lib.PREC_F16, "bf16": _lib.PREC_BF16}[prec]
        self.prec_name = prec
        self.train_dropout = 0.1     # nn.TransformerEncoderLayer's default, what the reference trains with (models.py:18)
        self.train_seed = 0
        self._engine = None
        import weakref
        self.slices_generator._owner = weakref.ref(self)
        # engine state (never part of state_dict)
        self._packed_key = None
        self._unet_packed = None
        self._head_packed = None
        self._ws = {}
        self._lib = _lib.load() if backend == "hip" else None

    # ------------------------------------------------------------------------------------------
    # engine plumbing
    # ------------------------------------------------------------------------------------------
    def _require_lib(self):
        if self._lib is None:
            raise _lib.S3dError("Slices3DRegModel(backend=%r) cannot compute: the HIP library is required "
                                "(backend='hip'); there is no CPU fallback in the product path" % self.backend)
        return self._lib

    def _require_eval(self):
        if self.training:
            raise RuntimeError(
                "encode / decode compute the eval-mode forward (running-stat BatchNorm, no dropout): call "
                "model.eval() for inference.  In train mode use model(batch) (autograd) or "
                "slice3d_amd.trainer.HipTrainer.train_step (the fused step)")

    def _train_engine(self):
        """The HIP training engine behind train-mode model(batch): its own flat gradient buffer (param.grad is left
        to autograd), dropout = self.train_dropout, dropout streams from self.train_seed."""
        from .trainer import HipTrainer
        self._require_lib()
        if self.prec_name not in ("f32", "f16x3"):
            raise ValueError("prec=%r is an inference-only throughput mode; train with prec='f16x3' or 'f32'" % self.prec_name)
        if self._engine is None or self._engine.grad_flat.device != self._device():
            self._engine = HipTrainer(self, dropout=self.train_dropout, seed=self.train_seed, prec=self.prec_name,
                                      bind_grads=False)
        self._engine.dropout = self.train_dropout
        return self._engine

    def _forward_train(self, feed_dict):
        """models.py:48-94 in training mode, differentiable w.r.t. the parameters (not the inputs)."""
        if "img_slices" not in feed_dict:
            raise KeyError("img_slices")      # the reference computes vgg_loss from it unconditionally (models.py:90-92)
        eng = self._train_engine()
        if not torch.is_grad_enabled():
            sdf, rec, vgg, _ = eng.forward_only(feed_dict)
        else:
            sdf, rec, vgg = _TrainForward.apply(eng, feed_dict, *eng.params)
        return {"sdf_pred": sdf, "slices_rec": rec, "vgg_loss": vgg}

    def _device(self):
        return self.fc_p.weight.device

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self._device()).cuda_stream)

    def _workspace(self, key, nbytes):
        buf = self._ws.get(key)
        if buf is None or buf.numel() < nbytes or buf.device != self._device():
            buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self._device())
            self._ws[key] = buf
        return buf

    def _params_key(self):
        return tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def _conv_params(self, conv, bn=None):
        cp = _lib.S3dConvParams()
        cp.w = conv.weight.data_ptr()
        cp.b = conv.bias.data_ptr() if conv.bias is not None else None
        if bn is not None:
            for i, t in enumerate((bn.weight, bn.bias, bn.running_mean, bn.running_var)):
                cp.bn[i] = t.data_ptr()
        return cp

    def repack(self):
        """(Re)build the MFMA-fragment-ordered, BN-folded weight images from the current parameters."""
        lib = self._require_lib()
        for t in list(self.parameters()) + list(self.buffers()):
            if t.is_floating_point() and (t.dtype != torch.float32 or not t.is_contiguous()):
                raise _lib.S3dError("parameters must be contiguous fp32")
        dev = self._device()
        if dev.type != "cuda":
            raise _lib.S3dError("model parameters are on %s; move the model to the GPU (model.cuda())" % dev)
        g = self.slices_generator
        up = _lib.S3dUNetParams()
        seqs = {name: getattr(g, name) for name, _, _ in _VGG16_SLICES}
        for i, (idx, _, _) in enumerate(_VGG16_CFG):
            conv = getattr(seqs[_slice_of(idx, _VGG16_SLICES)], str(idx))
            bn = getattr(seqs[_slice_of(idx + 1, _VGG16_SLICES)], str(idx + 1))
            up.enc[i] = self._conv_params(conv, bn)
        up.trans_c = self._conv_params(g.trans_c)
        for i in range(4):
            u = getattr(g, "up%d" % (i + 1))
            dc = u.conv.double_conv
            up.trans_up[i] = self._conv_params(getattr(g, "trans_up%d" % (i + 1)))
            up.up_t[i] = self._conv_params(u.up)
            up.up_c1[i] = self._conv_params(getattr(dc, "0"), getattr(dc, "1"))
            up.up_c2[i] = self._conv_params(getattr(dc, "3"), getattr(dc, "4"))
        up.outc = self._conv_params(g.outc.conv)
        up.emds = g.emds.weight.data_ptr()
        up.n_slices = self.n_slices
        nb = lib.s3d_unet_packed_bytes(self.n_slices)
        self._unet_packed = torch.empty(nb, dtype=torch.uint8, device=dev)
        _lib.check(lib.s3d_unet_pack(C.byref(up), self._unet_packed.data_ptr(), nb, self._stream()), "s3d_unet_pack")

        hp = _lib.S3dHeadParams()
        hp.fc_p_w, hp.fc_p_b = self.fc_p.weight.data_ptr(), self.fc_p.bias.data_ptr()
        hp.fc_s_w, hp.fc_s_b = self.fc_s.weight.data_ptr(), self.fc_s.bias.data_ptr()
        for i, layer in enumerate(self.att_decoder.layers):
            lp = hp.layer[i]
            lp.in_proj_w = layer.self_attn.in_proj_weight.data_ptr()
            lp.in_proj_b = layer.self_attn.in_proj_bias.data_ptr()
            lp.out_proj_w = layer.self_attn.out_proj.weight.data_ptr()
            lp.out_proj_b = layer.self_attn.out_proj.bias.data_ptr()
            lp.lin1_w, lp.lin1_b = layer.linear1.weight.data_ptr(), layer.linear1.bias.data_ptr()
            lp.lin2_w, lp.lin2_b = layer.linear2.weight.data_ptr(), layer.linear2.bias.data_ptr()
            lp.norm1_w, lp.norm1_b = layer.norm1.weight.data_ptr(), layer.norm1.bias.data_ptr()
            lp.norm2_w, lp.norm2_b = layer.norm2.weight.data_ptr(), layer.norm2.bias.data_ptr()
        hp.fc_out_w, hp.fc_out_b = self.fc_out[0].weight.data_ptr(), self.fc_out[0].bias.data_ptr()
        nb = lib.s3d_head_packed_bytes()
        self._head_packed = torch.empty(nb, dtype=torch.uint8, device=dev)
        _lib.check(lib.s3d_head_pack(C.byref(hp), self._head_packed.data_ptr(), nb, self._stream()), "s3d_head_pack")
        vp = _lib.S3dVggParams()
        vgg = self.vggptlossfunc.vgg
        for i, (idx, _, _) in enumerate(_VGG19_CONVS):
            vp.conv[i] = self._conv_params(getattr(getattr(vgg, _slice_of(idx, _VGG19_SLICES)), str(idx)))
        self._vgg_mean = self.vggptlossfunc.mean.reshape(3).contiguous()
        self._vgg_std = self.vggptlossfunc.std.reshape(3).contiguous()
        vp.mean, vp.std = self._vgg_mean.data_ptr(), self._vgg_std.data_ptr()
        nb = lib.s3d_vgg_packed_bytes()
        self._vgg_packed = torch.empty(nb, dtype=torch.uint8, device=dev)
        _lib.check(lib.s3d_vgg_pack(C.byref(vp), self._vgg_packed.data_ptr(), nb, self._stream()), "s3d_vgg_pack")
        self._packed_key = self._params_key()

    def _ensure_packed(self):
        if self._packed_key is None or self._packed_key != self._params_key():
            self.repack()

    def _f32(self, t):
        return t.to(device=self._device(), dtype=torch.float32).contiguous()

    def _nhwc_to_nchw(self, t):
        lib = self._require_lib()
        n, h, w, c = t.shape
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=t.device)
        _lib.check(lib.s3d_nhwc_to_nchw(t.data_ptr(), out.data_ptr(), n, c, h, w, self._stream()), "s3d_nhwc_to_nchw")
        return out

    # ------------------------------------------------------------------------------------------
    # reference helper API (models.py:28-46)
    # ------------------------------------------------------------------------------------------
    def project_coord(self, coordinates, trans_mat_wo_rot_tp):
        lib = self._require_lib()
        coords, tm = self._f32(coordinates), self._f32(trans_mat_wo_rot_tp)
        b, q, _ = coords.shape
        out = torch.empty((b, q, 2), dtype=torch.float32, device=coords.device)
        _lib.check(lib.s3d_project_coord_fwd(coords.data_ptr(), tm.data_ptr(), out.data_ptr(), b, q, self._stream()),
                   "s3d_project_coord_fwd")
        return out

    def sample_from_planes(self, plane_features, projected_coordinates, mode="bilinear", padding_mode="zeros",
                           box_warp=None):
        """plane_features (N,C,H,W) NCHW as in the reference; returns (N,1,M,C)."""
        if mode != "bilinear" or padding_mode != "zeros":
            raise ValueError("only bilinear / zeros is supported (the reference never uses anything else)")
        lib = self._require_lib()
        planes, grid = self._f32(plane_features), self._f32(projected_coordinates)
        n, c, h, w = planes.shape
        m = grid.shape[1]
        nhwc = torch.empty((n, h, w, c), dtype=torch.float32, device=planes.device)
        _lib.check(lib.s3d_nchw_to_nhwc(planes.data_ptr(), nhwc.data_ptr(), n, c, h, w, self._stream()), "s3d_nchw_to_nhwc")
        out = torch.empty((n, 1, m, c), dtype=torch.float32, device=planes.device)
        _lib.check(lib.s3d_sample_planes_fwd(nhwc.data_ptr(), grid.data_ptr(), out.data_ptr(), n, h, w, c, m,
                                             self._stream()), "s3d_sample_planes_fwd")
        return out

    def sample_pyramid(self, pyramid, projected_coordinates, out=None):
        """The reference's sampling block as one HBM-bound op (models.py:63-73): `pyramid` = the five
        channels-last levels of B*n_slices images (LatentCode.pyramid), projected_coordinates (B,Q,2)
        -> (B*n_slices, Q, 992), the tensor torch.cat(feat_interp, dim=2) holds in the reference.
        `out`: an optional caller-owned result tensor of that shape (4.76 GB at 12 x 100 000 rows: a loop that lets every
        call allocate it times the allocator, not the op)."""
        lib = self._require_lib()
        grid = self._f32(projected_coordinates)
        b, q, _ = grid.shape
        n_img, s = pyramid[0].shape[0], pyramid[4].shape[1]
        if n_img % b != 0 or [p.shape[-1] for p in pyramid] != list(LEVEL_CHANNELS):
            raise ValueError("pyramid does not match the batch / the (512,256,128,64,32) channel layout")
        pyr = _lib.S3dPyramid()
        for l in range(5):
            pyr.level[l] = pyramid[l].data_ptr()
        pyr.n_img, pyr.size = n_img, s
        shape = (n_img, q, sum(LEVEL_CHANNELS))
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=grid.device)
        elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != grid.device:
            raise ValueError("sample_pyramid: out must be a contiguous float32 %s tensor on %s" % (shape, grid.device))
        nb = lib.s3d_sample_pyramid_workspace_bytes(b, q)
        ws = self._workspace("sample_pyramid", nb)
        _lib.check(lib.s3d_sample_pyramid_fwd(C.byref(pyr), grid.data_ptr(), out.data_ptr(), b, n_img // b, q,
                                              ws.data_ptr(), nb, self._stream()), "s3d_sample_pyramid_fwd")
        return out

    # ------------------------------------------------------------------------------------------
    # encode / decode
    # ------------------------------------------------------------------------------------------
    def encode(self, feed_dict, want_slices=False, build_latent=True):
        """U-Net slice generator + (optionally) the fc_s-folded latent maps.  Runs once per object."""
        lib = self._require_lib()
        self._require_eval()
        self._ensure_packed()
        img = self._f32(feed_dict["img_input"])
        b, ch, s, s2 = img.shape
        if ch != 3 or s != s2 or s % 16 != 0:
            raise ValueError("img_input must be (B,3,S,S) with S a multiple of 16, got %s" % (tuple(img.shape),))
        ns, dev = self.n_slices, img.device
        code = LatentCode()
        code.batch, code.size, code.n_slices = b, s, ns
        code.pyramid = [torch.empty((b * ns, (s // 16) << l, (s // 16) << l, LEVEL_CHANNELS[l]),
                                    dtype=torch.float32, device=dev) for l in range(5)]
        pyr = _lib.S3dPyramid()
        for l in range(5):
            pyr.level[l] = code.pyramid[l].data_ptr()
        pyr.n_img, pyr.size = b * ns, s
        if want_slices:
            code.slices_rec_flat = torch.empty((b * ns, 3, s, s), dtype=torch.float32, device=dev)
        nb = lib.s3d_unet_workspace_bytes(b, s, ns)
        ws = self._workspace("unet", nb)
        _lib.check(lib.s3d_unet_encode_fwd(self._unet_packed.data_ptr(), img.data_ptr(), C.byref(pyr),
                                           code.slices_rec_flat.data_ptr() if want_slices else None,
                                           b, s, ns, self.prec, ws.data_ptr(), ws.numel(), self._stream()),
                   "s3d_unet_encode_fwd")
        if build_latent:
            code.proj = [torch.empty((b * ns, (s // 16) << l, (s // 16) << l, 128), dtype=torch.float32, device=dev)
                         for l in range(3)]
            lat = _lib.S3dLatent()
            for l in range(3):
                lat.proj[l] = code.proj[l].data_ptr()
            lat.fine[0], lat.fine[1] = code.pyramid[3].data_ptr(), code.pyramid[4].data_ptr()
            lat.n_img, lat.size = b * ns, s
            _lib.check(lib.s3d_latent_build(self._head_packed.data_ptr(), C.byref(pyr), C.byref(lat), self.prec,
                                            self._stream()),
                       "s3d_latent_build")
            code._latent_struct = lat
        for k in ("obj_rot_mat", "trans_mat_wo_rot_tp"):
            if k in feed_dict:
                setattr(code, k, self._f32(feed_dict[k]))
        return code

    def decode_sdf(self, p, c, obj_rot_mat=None, trans_mat_wo_rot_tp=None, mode=None):
        """sdf_pred (B,Q) for query points p (B,Q,3) given a LatentCode (models.py:53-84)."""
        lib = self._require_lib()
        self._require_eval()
        mode = self.mode if mode is None else mode
        qry = self._f32(p)
        b, q, _ = qry.shape
        if b != c.batch:
            raise ValueError("query batch %d != encoded batch %d" % (b, c.batch))
        tm = self._f32(trans_mat_wo_rot_tp) if trans_mat_wo_rot_tp is not None else c.trans_mat_wo_rot_tp
        if tm is None:
            raise KeyError("trans_mat_wo_rot_tp")
        rot = None
        if mode != "test":
            rot = self._f32(obj_rot_mat) if obj_rot_mat is not None else c.obj_rot_mat
            if rot is None:
                raise KeyError("obj_rot_mat")
        out = torch.empty((b, q), dtype=torch.float32, device=qry.device)
        nb = lib.s3d_decode_workspace_bytes(b, q, self.n_slices)
        ws = self._workspace("decode", nb)
        _lib.check(lib.s3d_decode_points_fwd(self._head_packed.data_ptr(), C.byref(c._latent_struct), qry.data_ptr(),
                                             rot.data_ptr() if rot is not None else None, tm.data_ptr(),
                                             1 if mode == "test" else 0, out.data_ptr(), b, q, self.n_slices,
                                             self.prec, ws.data_ptr(), ws.numel(), self._stream()),
                   "s3d_decode_points_fwd")
        return out

    def decode_stages(self, p, c, obj_rot_mat=None, trans_mat_wo_rot_tp=None, mode=None):
        """Debug / test form of decode_sdf that also returns the rows the reference's stage probes look at: a dict with
        'sdf' (B,Q), 'fc_p' (B,Q,128) = fc_p(qry_rot), 'fc_s' (B,Q,n_slices,128) = fc_s of the sampled pyramid
        (models.py:79-82) and 'layer0' / 'layer1' / 'layer2' (B,Q,128) = token 0 after each layer of att_decoder
        (models.py:83).  One decode pass in caller order: Q < 4096 per object."""
        lib = self._require_lib()
        self._require_eval()
        mode = self.mode if mode is None else mode
        qry = self._f32(p)
        b, q, _ = qry.shape
        tm = self._f32(trans_mat_wo_rot_tp) if trans_mat_wo_rot_tp is not None else c.trans_mat_wo_rot_tp
        rot = None
        if mode != "test":
            rot = self._f32(obj_rot_mat) if obj_rot_mat is not None else c.obj_rot_mat
        ns, t = self.n_slices, self.n_slices + 1
        gpb = (q + 15) // 16
        g = gpb * b
        out = torch.empty((b, q), dtype=torch.float32, device=qry.device)
        n_st = lib.s3d_decode_stages_floats(b, q, ns)
        assert n_st == g * 16 * 128 * (t + 3)
        st = torch.empty(n_st, dtype=torch.float32, device=qry.device)
        ws = self._workspace("decode", lib.s3d_decode_workspace_bytes(b, q, ns))
        _lib.check(lib.s3d_decode_points_stages_fwd(self._head_packed.data_ptr(), C.byref(c._latent_struct), qry.data_ptr(),
                                                    rot.data_ptr() if rot is not None else None, tm.data_ptr(),
                                                    1 if mode == "test" else 0, out.data_ptr(), st.data_ptr(), b, q, ns,
                                                    self.prec, ws.data_ptr(), ws.numel(), self._stream()),
                   "s3d_decode_points_stages_fwd")
        tok = st[:g * t * 16 * 128].view(b, gpb, t, 16, 128).permute(0, 1, 3, 2, 4).reshape(b, gpb * 16, t, 128)[:, :q]
        res = {"sdf": out, "fc_p": tok[:, :, 0], "fc_s": tok[:, :, 1:]}
        lay = st[g * t * 16 * 128:].view(3, b, gpb * 16, 128)[:, :, :q]
        for i in range(3):
            res["layer%d" % i] = lay[i]
        return res

    def decode(self, p, c, **kwargs):
        """ConvONet-style decode: `.logits` follows Generator3D's convention (= -sdf, reconstruct.py:97)."""
        sdf = self.decode_sdf(p, c, **kwargs)
        return SimpleNamespace(logits=-sdf, sdf=sdf)

    def decode_grid(self, c, nx, box=1.0, trans_mat_wo_rot_tp=None, q_range=None):
        """Dense nx^3 logits (-sdf) with in-kernel grid coordinates (reconstruct.py:135-146); batch 1.
        q_range=(lo, hi): only that contiguous slab of the grid's linear index (x slowest, z fastest), returned
        flat — the per-rank piece of the query-parallel split (slice3d_amd.parallel).
        The kernel runs the mode='test' prologue (the mode reconstruct.py:336 builds the model with); a model in
        another mode must go through decode_sdf on explicit points, which applies obj_rot_mat."""
        lib = self._require_lib()
        self._require_eval()
        if self.mode != "test":
            raise ValueError("decode_grid evaluates the mode='test' prologue (y, z negated, no obj_rot_mat); this model "
                             "has mode=%r — decode explicit grid points with decode_sdf instead" % self.mode)
        if c.batch != 1:
            raise ValueError("decode_grid expects a single encoded object")
        tm = self._f32(trans_mat_wo_rot_tp) if trans_mat_wo_rot_tp is not None else c.trans_mat_wo_rot_tp
        if tm is None:
            raise KeyError("trans_mat_wo_rot_tp")
        n = nx ** 3
        lo, hi = (0, n) if q_range is None else (int(q_range[0]), int(q_range[1]))
        if not 0 <= lo <= hi <= n:
            raise ValueError("q_range %r outside the %d-point grid" % (q_range, n))
        out = torch.empty((hi - lo,), dtype=torch.float32, device=self._device())
        if q_range is None:
            nb = lib.s3d_decode_workspace_bytes(1, n, self.n_slices)
            ws = self._workspace("decode", nb)
            _lib.check(lib.s3d_decode_grid_fwd(self._head_packed.data_ptr(), C.byref(c._latent_struct), tm.data_ptr(), nx,
                                               float(box), out.data_ptr(), self.n_slices, self.prec, ws.data_ptr(),
                                               ws.numel(), self._stream()), "s3d_decode_grid_fwd")
        elif hi > lo:
            nb = lib.s3d_decode_workspace_bytes(1, hi - lo, self.n_slices)
            ws = self._workspace("decode", nb)
            _lib.check(lib.s3d_decode_grid_slab_fwd(self._head_packed.data_ptr(), C.byref(c._latent_struct),
                                                    tm.data_ptr(), nx, float(box), lo, hi - lo, out.data_ptr(),
                                                    self.n_slices, self.prec, ws.data_ptr(), ws.numel(),
                                                    self._stream()), "s3d_decode_grid_slab_fwd")
        return out.view(nx, nx, nx) if q_range is None else out

    # ------------------------------------------------------------------------------------------
    # reference forward (models.py:48-94)
    # ------------------------------------------------------------------------------------------
    def forward(self, feed_dict):
        """Unlike the reference, `qry_norot` is NOT modified in place in mode='test' (models.py:55).  In training mode
        (model.train()) this is the train-mode forward — batch-statistics BatchNorm, dropout — and the outputs carry
        autograd history back to the parameters."""
        if self.training:
            return self._forward_train(feed_dict)
        code = self.encode(feed_dict, want_slices=True)
        sdf = self.decode_sdf(feed_dict["qry_norot"], code,
                              obj_rot_mat=feed_dict.get("obj_rot_mat"),
                              trans_mat_wo_rot_tp=feed_dict["trans_mat_wo_rot_tp"])
        ret = {"sdf_pred": sdf, "slices_rec": code.slices_rec}
        if "img_slices" in feed_dict:
            ret["vgg_loss"] = self.vgg_loss(code.slices_rec_flat, feed_dict["img_slices"])
        else:  # inference callers (Generator3D.eval_points) never read it
            ret["vgg_loss"] = torch.zeros((), dtype=torch.float32, device=sdf.device)
        return ret

    def vgg_loss(self, slices_rec_flat, img_slices):
        """VGGPerceptualLoss(slices_rec, img_slices)['pt_c_loss'] * 0.001 (models.py:90-92)."""
        lib = self._require_lib()
        self._ensure_packed()
        pred = self._f32(slices_rec_flat)
        n, _, s, _ = pred.shape
        tgt = self._f32(img_slices).reshape(n, 3, s, s)
        out = torch.empty((), dtype=torch.float32, device=pred.device)
        nb = lib.s3d_vgg_workspace_bytes(n, s)
        ws = self._workspace("vgg", nb)
        _lib.check(lib.s3d_vgg_loss_fwd(self._vgg_packed.data_ptr(), pred.data_ptr(), tgt.data_ptr(), n, s,
                                        out.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()), "s3d_vgg_loss_fwd")
        return out
