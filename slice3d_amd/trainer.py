"""Training step on the HIP path — host side of reg_slices/train.py:41-53 (train_step) and :136 (Adam).

`HipTrainer.train_step(batch)` = zero_grad -> model(batch) in train mode -> the three losses -> backward
-> (all-reduce of gradients across ranks) -> Adam step, all inside libslice3d_hip.so
(s3d_train_fwd_bwd + s3d_adam_step).  PyTorch owns the parameter / gradient / optimiser-state memory and
runs the RCCL all-reduce; gradients are exposed as `param.grad` views into one flat buffer.
"""
import ctypes as C

import torch

from . import _lib
from .models import _VGG16_CFG, _VGG16_SLICES, _VGG19_CONVS, _VGG19_SLICES, _slice_of


class HipTrainer:
    def __init__(self, model, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, dropout=0.0, process_group=None, seed=0,
                 prec="f32"):
        self.model = model
        self.lib = _lib.load()
        self.lr, self.betas, self.eps = lr, betas, eps
        self.dropout = dropout
        self.prec = {"f32": _lib.PREC_F32, "f16x3": _lib.PREC_F16X3}[prec]   # conv / linear GEMMs; wgrad stays fp32
        self.group = process_group
        self.step = 0
        self.seed, self._calls, self.last_seed = seed, 0, 0
        dev = model.fc_out[0].weight.device
        if dev.type != "cuda":
            raise _lib.S3dError("move the model to the GPU before building a HipTrainer")
        self.names, self.params = [], []
        for k, p in model.named_parameters():
            if not self._trainable(k):
                continue
            self.names.append(k)
            self.params.append(p)
        n_total = sum(p.numel() for p in self.params)
        self.grad_flat = torch.zeros(n_total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros_like(self.grad_flat)
        self.exp_avg_sq = torch.zeros_like(self.grad_flat)
        self.offsets, off = {}, 0
        for k, p in zip(self.names, self.params):
            self.offsets[k] = off
            p.grad = self.grad_flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self._losses = torch.zeros(4, dtype=torch.float32, device=dev)
        self._ws = None

    @staticmethod
    def _trainable(k):
        """Trainable tensors = everything the forward touches (reference: 14 tensors never get a grad — the dead
        att_layer.* twin and down5_.41.* — and vggptlossfunc.* is frozen)."""
        return not (k.startswith("att_layer.") or k.startswith("vggptlossfunc.") or ".down5_." in k)

    # -- struct builders ------------------------------------------------------------------------
    def _gptr(self, t):
        return t.grad.data_ptr() if t.grad is not None else None

    def _conv(self, conv, bn, grad):
        cp = _lib.S3dConvParams()
        pick = (lambda t: self._gptr(t)) if grad else (lambda t: t.data_ptr())
        cp.w = pick(conv.weight)
        cp.b = pick(conv.bias) if conv.bias is not None else None
        if bn is not None:
            cp.bn[0], cp.bn[1] = pick(bn.weight), pick(bn.bias)
            if not grad:
                cp.bn[2], cp.bn[3] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        return cp

    def _unet_struct(self, grad):
        g = self.model.slices_generator
        up = _lib.S3dUNetParams()
        seqs = {name: getattr(g, name) for name, _, _ in _VGG16_SLICES}
        for i, (idx, _, _) in enumerate(_VGG16_CFG):
            conv = getattr(seqs[_slice_of(idx, _VGG16_SLICES)], str(idx))
            bn = getattr(seqs[_slice_of(idx + 1, _VGG16_SLICES)], str(idx + 1)) if i < 12 else None
            up.enc[i] = self._conv(conv, bn, grad)
        up.trans_c = self._conv(g.trans_c, None, grad)
        for i in range(4):
            u = getattr(g, "up%d" % (i + 1))
            dc = u.conv.double_conv
            up.trans_up[i] = self._conv(getattr(g, "trans_up%d" % (i + 1)), None, grad)
            up.up_t[i] = self._conv(u.up, None, grad)
            up.up_c1[i] = self._conv(getattr(dc, "0"), getattr(dc, "1"), grad)
            up.up_c2[i] = self._conv(getattr(dc, "3"), getattr(dc, "4"), grad)
        up.outc = self._conv(g.outc.conv, None, grad)
        up.emds = self._gptr(g.emds.weight) if grad else g.emds.weight.data_ptr()
        up.n_slices = self.model.n_slices
        return up

    def _fill_layers(self, hp, pick):
        m = self.model
        for i, layer in enumerate(m.att_decoder.layers):
            lp = hp.layer[i]
            lp.in_proj_w, lp.in_proj_b = pick(layer.self_attn.in_proj_weight), pick(layer.self_attn.in_proj_bias)
            lp.out_proj_w, lp.out_proj_b = pick(layer.self_attn.out_proj.weight), pick(layer.self_attn.out_proj.bias)
            lp.lin1_w, lp.lin1_b = pick(layer.linear1.weight), pick(layer.linear1.bias)
            lp.lin2_w, lp.lin2_b = pick(layer.linear2.weight), pick(layer.linear2.bias)
            lp.norm1_w, lp.norm1_b = pick(layer.norm1.weight), pick(layer.norm1.bias)
            lp.norm2_w, lp.norm2_b = pick(layer.norm2.weight), pick(layer.norm2.bias)
        hp.fc_out_w, hp.fc_out_b = pick(m.fc_out[0].weight), pick(m.fc_out[0].bias)

    def _head_struct(self, grad):
        m = self.model
        pick = (lambda t: self._gptr(t)) if grad else (lambda t: t.data_ptr())
        hp = _lib.S3dHeadParams()
        hp.fc_p_w, hp.fc_p_b = pick(m.fc_p.weight), pick(m.fc_p.bias)
        hp.fc_s_w, hp.fc_s_b = pick(m.fc_s.weight), pick(m.fc_s.bias)
        self._fill_layers(hp, pick)
        return hp

    def _vgg_struct(self):
        vp = _lib.S3dVggParams()
        vgg = self.model.vggptlossfunc.vgg
        for i, (idx, _, _) in enumerate(_VGG19_CONVS):
            conv = getattr(getattr(vgg, _slice_of(idx, _VGG19_SLICES)), str(idx))
            vp.conv[i].w, vp.conv[i].b = conv.weight.data_ptr(), conv.bias.data_ptr()
        self._mean = self.model.vggptlossfunc.mean.reshape(3).contiguous()
        self._std = self.model.vggptlossfunc.std.reshape(3).contiguous()
        vp.mean, vp.std = self._mean.data_ptr(), self._std.data_ptr()
        return vp

    def _next_seed(self):
        """Fresh dropout seed per call (rank-dependent so data-parallel ranks draw different masks)."""
        self.last_seed = (self.seed * 1000003 + self.step * 7919 + self._calls) & 0xFFFFFFFFFFFF
        self._calls += 1
        return self.last_seed

    # -- one step ---------------------------------------------------------------------------------
    def forward_backward(self, batch, want_outputs=False):
        """Train-mode forward + losses + backward; fills param.grad.  Returns the device tensor
        [loss_pred, loss_img, loss_vgg, acc] (and sdf_pred / slices_rec if asked)."""
        m, lib = self.model, self.lib
        dev = self.grad_flat.device
        f = lambda k: batch[k].to(device=dev, dtype=torch.float32).contiguous()
        img, sl, qry, rot, tm, sdf = (f(k) for k in ("img_input", "img_slices", "qry_norot", "obj_rot_mat",
                                                       "trans_mat_wo_rot_tp", "sdf"))
        b, _, s, _ = img.shape
        q, ns = qry.shape[1], m.n_slices
        tb = _lib.S3dTrainBatch()
        tb.img, tb.img_slices, tb.qry = img.data_ptr(), sl.data_ptr(), qry.data_ptr()
        tb.rot, tb.trans, tb.sdf = rot.data_ptr(), tm.data_ptr(), sdf.data_ptr()
        nb = lib.s3d_train_workspace_bytes(b, s, q, ns)
        if self._ws is None or self._ws.numel() < nb:
            self._ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        sdf_pred = torch.empty((b, q), dtype=torch.float32, device=dev) if want_outputs else None
        rec = torch.empty((b * ns, 3, s, s), dtype=torch.float32, device=dev) if want_outputs else None
        u, h, v = self._unet_struct(False), self._head_struct(False), self._vgg_struct()
        du, dh = self._unet_struct(True), self._head_struct(True)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.s3d_train_fwd_bwd(C.byref(u), C.byref(h), C.byref(v), C.byref(du), C.byref(dh), C.byref(tb),
                                         b, s, q, ns, float(self.dropout), self._next_seed(), self.prec,
                                         self._losses.data_ptr(),
                                         sdf_pred.data_ptr() if want_outputs else None,
                                         rec.data_ptr() if want_outputs else None,
                                         self._ws.data_ptr(), self._ws.numel(), stream), "s3d_train_fwd_bwd")
        m._packed_key = None   # BN running statistics changed in place: eval-mode packs are stale
        if want_outputs:
            return self._losses, sdf_pred, rec.view(b, ns * 3, s, s)
        return self._losses

    def all_reduce_grads(self):
        """Data-parallel exchange step: mean of the gradients over ranks (one flat 83 MB bucket)."""
        from .parallel import all_reduce_mean_
        all_reduce_mean_(self.grad_flat, self.group)

    def adam_step(self):
        self.step += 1
        stream = C.c_void_p(torch.cuda.current_stream(self.grad_flat.device).cuda_stream)
        n = len(self.params)
        ptrs = (C.c_void_p * n)(*[p.data_ptr() for p in self.params])
        offs = (C.c_long * n)(*[self.offsets[k] for k in self.names])
        sizes = (C.c_long * n)(*[p.numel() for p in self.params])
        _lib.check(self.lib.s3d_adam_step_multi(ptrs, offs, sizes, n, self.grad_flat.data_ptr(), self.exp_avg.data_ptr(),
                                                self.exp_avg_sq.data_ptr(), self.lr, self.betas[0], self.betas[1],
                                                self.eps, self.step, stream), "s3d_adam_step_multi")
        self.model._packed_key = None

    def state_dict(self):
        """Optimiser state in the checkpoint layout of the reference's `torch.optim.Adam(model.parameters())`
        ('opt' entry, train.py:136,174-176): parameters are numbered by their position in model.parameters() — ALL of
        them, frozen VGG19 and never-touched tensors included — and only the tensors that received a gradient carry a
        state entry, exactly what torch writes.  `torch.optim.Adam(model.parameters()).load_state_dict()` accepts it."""
        index = {id(p): i for i, p in enumerate(self.model.parameters())}
        state = {}
        for k, p in zip(self.names, self.params):
            if self.step == 0:
                break              # torch creates the state lazily at the first step
            off, n = self.offsets[k], p.numel()
            state[index[id(p)]] = {"step": torch.tensor(float(self.step)),
                                   "exp_avg": self.exp_avg[off:off + n].view_as(p).clone(),
                                   "exp_avg_sq": self.exp_avg_sq[off:off + n].view_as(p).clone()}
        group = dict(torch.optim.Adam([torch.zeros(1)]).state_dict()["param_groups"][0])   # this torch's key set
        group.update(lr=self.lr, betas=tuple(self.betas), eps=self.eps, weight_decay=0, amsgrad=False,
                     params=list(range(len(index))))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts the 'opt' entry of a reference checkpoint (or of state_dict() above): state keyed by the index in
        model.parameters().  Shapes are validated; a state entry for a tensor this trainer does not update is an error."""
        g = sd["param_groups"][0]
        all_params = list(self.model.named_parameters())
        if len(g["params"]) != len(all_params):
            raise _lib.S3dError("optimizer state numbers %d parameters, the model has %d (a checkpoint of another model?)"
                                % (len(g["params"]), len(all_params)))
        self.lr, self.betas, self.eps = g["lr"], tuple(g["betas"]), g["eps"]
        position = {idx: i for i, idx in enumerate(g["params"])}
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        steps = set()
        for idx, st in sd["state"].items():
            name, p = all_params[position[int(idx)]]
            if name not in self.offsets:
                raise _lib.S3dError("optimizer state for %s, which this trainer never updates" % name)
            if tuple(st["exp_avg"].shape) != tuple(p.shape) or tuple(st["exp_avg_sq"].shape) != tuple(p.shape):
                raise _lib.S3dError("optimizer state of %s has shape %s, the parameter %s"
                                    % (name, tuple(st["exp_avg"].shape), tuple(p.shape)))
            off, n = self.offsets[name], p.numel()
            self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(st["step"]))
        if len(steps) > 1:
            raise _lib.S3dError("optimizer state holds different step counts %s (one global step is kept)" % sorted(steps))
        self.step = steps.pop() if steps else 0

    def train_step(self, batch):
        """train.py:41-53 — returns python floats (loss_pred, loss_img, loss_img_vgg, acc)."""
        losses = self.forward_backward(batch)
        self.all_reduce_grads()
        self.adam_step()
        lp, li, lv, acc = losses.tolist()    # the reference's 4 .item() syncs, as one
        return lp, li, lv, acc


class HipGtTrainer(HipTrainer):
    """Training step of Slices3DGTModel — host side of reg_slices/train_gt.py:38-52 (train_step) and :115 (Adam):
    s3d_gt_train_fwd_bwd + s3d_adam_step.  `train_step(batch)` returns (loss_pred, acc)."""

    @staticmethod
    def _trainable(k):
        # never reached by the forward's gradient (model_gt.py:77 drops feat_global; fc_global and the att_layer
        # twin are dead): torch.optim.Adam skips them because their .grad stays None
        dead = ("att_layer.", "fc_global.", "img_encoder.classifier.", "img_encoder.conv_last.")
        return not k.startswith(dead)

    def _enc_struct(self, grad):
        from .models_gt import _GT_SLICES
        e = self.model.img_encoder
        vp = _lib.S3dVgg16BnParams()
        for i, (idx, _, _) in enumerate(_VGG16_CFG):
            conv = getattr(getattr(e, _slice_of(idx, _GT_SLICES)), str(idx))
            bn = getattr(getattr(e, _slice_of(idx + 1, _GT_SLICES)), str(idx + 1))
            if i < 12:
                vp.conv[i] = self._conv(conv, bn, grad)
            else:   # conv5_3's BatchNorm (conv_last.41): running statistics move, weight / bias get no gradient
                vp.conv[i] = self._conv(conv, None, grad)
                if not grad:
                    vp.conv[i].bn[2], vp.conv[i].bn[3] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        return vp

    def _gt_head_struct(self, grad):
        m = self.model
        pick = (lambda t: self._gptr(t)) if grad else (lambda t: t.data_ptr())
        hp = _lib.S3dGtHeadParams()
        for k, idx in enumerate((0, 2, 4)):
            hp.pts_w[k], hp.pts_b[k] = pick(m.pts_feat_extractor[idx].weight), pick(m.pts_feat_extractor[idx].bias)
        for k, idx in enumerate((0, 2)):
            hp.local_w[k], hp.local_b[k] = pick(m.fc_local[idx].weight), pick(m.fc_local[idx].bias)
        self._fill_layers(hp, pick)
        return hp

    def forward_backward(self, batch, want_outputs=False):
        """Train-mode forward + L1 loss + backward; fills param.grad.  Returns the device tensor
        [loss_pred, acc, 0, 0] (and sdf_pred if asked)."""
        m, lib = self.model, self.lib
        dev = self.grad_flat.device
        f = lambda k: batch[k].to(device=dev, dtype=torch.float32).contiguous()
        sl, qry, rot, tm, sdf = (f(k) for k in ("img_slices", "qry_norot", "obj_rot_mat", "trans_mat_wo_rot_tp",
                                                  "sdf"))
        b, c, s, _ = sl.shape
        q, ns = qry.shape[1], m.n_slices
        if c != 3 * ns:
            raise ValueError("img_slices has %d channels, expected 3*n_slices = %d" % (c, 3 * ns))
        tb = _lib.S3dTrainBatch()
        tb.img_slices, tb.qry = sl.data_ptr(), qry.data_ptr()
        tb.rot, tb.trans, tb.sdf = rot.data_ptr(), tm.data_ptr(), sdf.data_ptr()
        nb = lib.s3d_gt_train_workspace_bytes(b, s, q, ns)
        if self._ws is None or self._ws.numel() < nb:
            self._ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        sdf_pred = torch.empty((b, q), dtype=torch.float32, device=dev) if want_outputs else None
        e, h = self._enc_struct(False), self._gt_head_struct(False)
        de, dh = self._enc_struct(True), self._gt_head_struct(True)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.s3d_gt_train_fwd_bwd(C.byref(e), C.byref(h), C.byref(de), C.byref(dh), C.byref(tb),
                                            b, s, q, ns, float(self.dropout), self._next_seed(), self.prec,
                                            self._losses.data_ptr(),
                                            sdf_pred.data_ptr() if want_outputs else None,
                                            self._ws.data_ptr(), self._ws.numel(), stream), "s3d_gt_train_fwd_bwd")
        m._packed_key = None   # BN running statistics changed in place: eval-mode packs are stale
        if want_outputs:
            return self._losses, sdf_pred
        return self._losses

    def train_step(self, batch):
        """train_gt.py:38-52 — returns python floats (loss_pred, acc)."""
        losses = self.forward_backward(batch)
        self.all_reduce_grads()
        self.adam_step()
        lp, acc = losses[:2].tolist()
        return lp, acc
