"""Slices3DGTModel — MI355X-native counterpart of reg_slices/src/model_gt.py:12-111.

The regression model of the generation-based pipeline: it reads the 12 GIVEN slice images (ground truth at
training time, LDM samples at test time) instead of generating them.  Same module surface and the same
state_dict keys as the reference (including the unused `att_layer`, `fc_global` and `img_encoder.classifier`
parameters, so released checkpoints load with strict=True); the compute runs through libslice3d_hip.so
(C ABI: include/slice3d_hip.h, "Slices3DGTModel" section).  There is no CPU fallback.

Split as for Slices3DRegModel: `encode()` once per object (VGG16-BN pyramid of the slice images + the
fc_local[0]-folded latent maps), `decode_sdf()` per query batch.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .models import _VGG16_CFG, _slice_of

# vgg16bn_feats.py:33-38: torchvision vgg16_bn.features sliced [:4] [4:11] [11:21] [21:31] [31:41] [41:44]
_GT_SLICES = (("conv1_2", 0, 4), ("conv2_2", 4, 11), ("conv3_3", 11, 21), ("conv4_3", 21, 31),
              ("conv5_3", 31, 41), ("conv_last", 41, 44))
GT_LEVEL_CHANNELS = (64, 128, 256, 512, 512)


class VGG16BNFeats(nn.Module):
    """Parameter container with the reference's keys (vgg16bn_feats.py:26-40)."""

    def __init__(self):
        super().__init__()
        seqs = {name: nn.Sequential() for name, _, _ in _GT_SLICES}
        for idx, cin, cout in _VGG16_CFG:
            seqs[_slice_of(idx, _GT_SLICES)].add_module(str(idx), nn.Conv2d(cin, cout, 3, padding=1))
            seqs[_slice_of(idx + 1, _GT_SLICES)].add_module(str(idx + 1), nn.BatchNorm2d(cout))
        for name, _, _ in _GT_SLICES:
            setattr(self, name, seqs[name])
        self.classifier = nn.Linear(512 * 4 * 4, 128)   # feat_global: computed by the reference, never used


class GtLatentCode:
    """Output of encode(): raw pyramid + folded latent maps of B*n_slices images."""
    pyramid = None
    proj = None
    batch = size = n_slices = 0

    def latent_struct(self):
        lat = _lib.S3dGtLatent()
        for l in range(4):
            lat.proj[l] = self.proj[l].data_ptr()
        lat.fine = self.pyramid[0].data_ptr()
        lat.n_img, lat.size = self.batch * self.n_slices, self.size
        return lat


class Slices3DGTModel(nn.Module):
    def __init__(self, img_size=128, n_slices=12, mode="train", backend="hip", prec="f16x3"):
        super().__init__()
        if prec not in ("f32", "f16x3"):
            raise ValueError("prec must be 'f32' or 'f16x3'")
        self.mode, self.img_size, self.n_slices, self.backend, self.prec = mode, img_size, n_slices, backend, prec
        self.img_encoder = VGG16BNFeats()
        self.att_layer = nn.TransformerEncoderLayer(d_model=128, nhead=4, batch_first=True)
        self.att_decoder = nn.TransformerEncoder(self.att_layer, num_layers=3)
        self.fc_out = nn.Sequential(nn.Linear(128, 1))
        self.pts_feat_extractor = nn.Sequential(nn.Linear(3, 32), nn.ReLU(), nn.Linear(32, 64), nn.ReLU(),
                                                nn.Linear(64, 128), nn.ReLU())
        self.fc_local = nn.Sequential(nn.Linear(1472, 128), nn.ReLU(), nn.Linear(128, 128), nn.ReLU())
        self.fc_global = nn.Sequential(nn.Linear(128 + 128, 128), nn.ReLU(), nn.Linear(128, 128), nn.ReLU())
        self._packed_key = None
        self._enc_packed = self._head_packed = None
        self._ws = {}
        self._lib = _lib.load() if backend == "hip" else None

    # ------------------------------------------------------------------------------------------
    def _require_lib(self):
        if self._lib is None:
            raise _lib.S3dError("Slices3DGTModel(backend=%r) cannot compute: the HIP library is required "
                                "(backend='hip'); there is no CPU fallback in the product path" % self.backend)
        return self._lib

    def _require_eval(self):
        if self.training:
            raise RuntimeError("Slices3DGTModel computes the eval-mode forward (running-stat BatchNorm, no dropout); "
                               "call model.eval()")

    def _device(self):
        return self.fc_out[0].weight.device

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self._device()).cuda_stream)

    def _workspace(self, key, nbytes):
        buf = self._ws.get(key)
        if buf is None or buf.numel() < nbytes or buf.device != self._device():
            buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self._device())
            self._ws[key] = buf
        return buf

    def _f32(self, t):
        return t.to(device=self._device(), dtype=torch.float32).contiguous()

    def _prec(self):
        return {"f32": _lib.PREC_F32, "f16x3": _lib.PREC_F16X3, "f16": _lib.PREC_F16}[self.prec]

    def _params_key(self):
        return tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def repack(self):
        lib = self._require_lib()
        dev = self._device()
        if dev.type != "cuda":
            raise _lib.S3dError("model parameters are on %s; move the model to the GPU (model.cuda())" % dev)
        for t in list(self.parameters()) + list(self.buffers()):
            if t.is_floating_point() and (t.dtype != torch.float32 or not t.is_contiguous()):
                raise _lib.S3dError("parameters must be contiguous fp32")
        e = self.img_encoder
        vp = _lib.S3dVgg16BnParams()
        for i, (idx, _, _) in enumerate(_VGG16_CFG):
            conv = getattr(getattr(e, _slice_of(idx, _GT_SLICES)), str(idx))
            bn = getattr(getattr(e, _slice_of(idx + 1, _GT_SLICES)), str(idx + 1))
            cp = vp.conv[i]
            cp.w, cp.b = conv.weight.data_ptr(), conv.bias.data_ptr()
            for k, t in enumerate((bn.weight, bn.bias, bn.running_mean, bn.running_var)):
                cp.bn[k] = t.data_ptr()
        nb = lib.s3d_gt_encoder_packed_bytes()
        self._enc_packed = torch.empty(nb, dtype=torch.uint8, device=dev)
        _lib.check(lib.s3d_gt_encoder_pack(C.byref(vp), self._enc_packed.data_ptr(), nb, self._stream()),
                   "s3d_gt_encoder_pack")
        hp = _lib.S3dGtHeadParams()
        for k, idx in enumerate((0, 2, 4)):
            hp.pts_w[k] = self.pts_feat_extractor[idx].weight.data_ptr()
            hp.pts_b[k] = self.pts_feat_extractor[idx].bias.data_ptr()
        for k, idx in enumerate((0, 2)):
            hp.local_w[k] = self.fc_local[idx].weight.data_ptr()
            hp.local_b[k] = self.fc_local[idx].bias.data_ptr()
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
        nb = lib.s3d_gt_head_packed_bytes()
        self._head_packed = torch.empty(nb, dtype=torch.uint8, device=dev)
        _lib.check(lib.s3d_gt_head_pack(C.byref(hp), self._head_packed.data_ptr(), nb, self._stream()),
                   "s3d_gt_head_pack")
        self._packed_key = self._params_key()

    def _ensure_packed(self):
        if self._packed_key is None or self._packed_key != self._params_key():
            self.repack()

    # ------------------------------------------------------------------------------------------
    def encode(self, feed_dict):
        """VGG16-BN pyramid of the slice images (model_gt.py:73-76) + folded latent maps; once per object."""
        lib = self._require_lib()
        self._require_eval()
        self._ensure_packed()
        sl = self._f32(feed_dict["img_slices"])
        b, ch, s, s2 = sl.shape
        ns = self.n_slices
        if ch != 3 * ns or s != s2 or s % 16 != 0:
            raise ValueError("img_slices must be (B,%d,S,S) with S a multiple of 16, got %s" % (3 * ns, tuple(sl.shape)))
        dev = sl.device
        code = GtLatentCode()
        code.batch, code.size, code.n_slices = b, s, ns
        n_img = b * ns
        code.pyramid = [torch.empty((n_img, s >> l, s >> l, GT_LEVEL_CHANNELS[l]), dtype=torch.float32, device=dev)
                        for l in range(5)]
        pyr = _lib.S3dGtPyramid()
        for l in range(5):
            pyr.level[l] = code.pyramid[l].data_ptr()
        pyr.n_img, pyr.size = n_img, s
        nb = lib.s3d_gt_encoder_workspace_bytes(n_img, s)
        ws = self._workspace("enc", nb)
        _lib.check(lib.s3d_gt_encode_fwd(self._enc_packed.data_ptr(), sl.data_ptr(), C.byref(pyr), n_img, s,
                                         self._prec(), ws.data_ptr(), nb, self._stream()), "s3d_gt_encode_fwd")
        code.proj = [torch.empty((n_img, s >> (4 - l), s >> (4 - l), 128), dtype=torch.float32, device=dev)
                     for l in range(4)]
        lat = code.latent_struct()
        _lib.check(lib.s3d_gt_latent_build(self._head_packed.data_ptr(), C.byref(pyr), C.byref(lat), self._prec(),
                                           self._stream()), "s3d_gt_latent_build")
        return code

    def decode_sdf(self, p, c, obj_rot_mat=None, trans_mat_wo_rot_tp=None, mode=None):
        """p (B,Q,3) un-rotated queries -> sdf (B,Q); mode 'test' negates y,z (model_gt.py:63-70)."""
        lib = self._require_lib()
        self._require_eval()
        self._ensure_packed()
        mode = self.mode if mode is None else mode
        qry, tm = self._f32(p), self._f32(trans_mat_wo_rot_tp)
        b, q, _ = qry.shape
        flip = 1 if mode == "test" else 0
        rot = None if flip else self._f32(obj_rot_mat)
        out = torch.empty((b, q), dtype=torch.float32, device=qry.device)
        nb = lib.s3d_gt_decode_workspace_bytes(b, q, self.n_slices)
        ws = self._workspace("dec", nb)
        lat = c.latent_struct()
        _lib.check(lib.s3d_gt_decode_points_fwd(self._head_packed.data_ptr(), C.byref(lat), qry.data_ptr(),
                                                rot.data_ptr() if rot is not None else None, tm.data_ptr(), flip,
                                                out.data_ptr(), b, q, self.n_slices, self._prec(), ws.data_ptr(), nb,
                                                self._stream()), "s3d_gt_decode_points_fwd")
        return out

    def decode_grid(self, c, nx, box=1.0, trans_mat_wo_rot_tp=None):
        """Occupancy logits (-sdf) on the dense nx^3 grid box*linspace(-.5,.5,nx)^3 (reconstruct.py:135-146)."""
        lib = self._require_lib()
        self._require_eval()
        self._ensure_packed()
        tm = self._f32(trans_mat_wo_rot_tp)
        n = nx ** 3
        out = torch.empty((n,), dtype=torch.float32, device=tm.device)
        nb = lib.s3d_gt_decode_workspace_bytes(1, n, self.n_slices)
        ws = self._workspace("dec", nb)
        lat = c.latent_struct()
        _lib.check(lib.s3d_gt_decode_grid_fwd(self._head_packed.data_ptr(), C.byref(lat), tm.data_ptr(), nx,
                                              C.c_float(box), out.data_ptr(), self.n_slices, self._prec(),
                                              ws.data_ptr(), nb, self._stream()), "s3d_gt_decode_grid_fwd")
        return out.view(nx, nx, nx)

    def forward(self, feed_dict):
        """model_gt.py:59-111 -> {'sdf_pred': (B,Q)}.  Unlike the reference, 'test' mode does not modify
        feed_dict['qry_norot'] in place."""
        code = self.encode(feed_dict)
        sdf = self.decode_sdf(feed_dict["qry_norot"], code, obj_rot_mat=feed_dict.get("obj_rot_mat"),
                              trans_mat_wo_rot_tp=feed_dict["trans_mat_wo_rot_tp"])
        return {"sdf_pred": sdf}
