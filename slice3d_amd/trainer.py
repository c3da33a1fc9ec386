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


def bucket_ranges(names, sizes):
    """Contiguous [lo, hi) ranges of the flat gradient buffer (parameters laid out in `names` order, which is
    model.named_parameters() order) for the four buckets the backward finishes in this order (include/slice3d_hip.h,
    S3dTrainBatch.ev_grad_ready): 0 transformer decoder + fc_*, 1 the U-Net's decoder half, 2 encoder convs 7..12
    (torchvision indices >= 24) with their BatchNorms, 3 the shallow encoder.  Returned in that order."""
    def bucket(k):
        if not k.startswith("slices_generator."):
            return 0
        part = k.split(".")[1]
        if not part.startswith("down"):
            return 1
        return 2 if int(k.split(".")[2]) >= 24 else 3
    ranges, off = {}, 0
    for k, n in zip(names, sizes):
        b = bucket(k)
        lo, hi = ranges.get(b, (off, off))
        if hi != off:
            raise AssertionError("bucket %d is not contiguous in the flat gradient buffer (at %s)" % (b, k))
        ranges[b] = (lo, off + n)
        off += n
    return [ranges[b] for b in range(4) if b in ranges]


class HipTrainer:
    def __init__(self, model, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, dropout=0.0, process_group=None, seed=0,
                 prec="f32", bind_grads=True, overlap_all_reduce=True, sync_bn=False):
        self.model = model
        self.lib = _lib.load()
        self.lr, self.betas, self.eps = lr, betas, eps
        self.dropout = dropout
        # "f32": exact fp32 MFMAs; "f16x3": split precision, fp32-class (the headline mode); "f16" (round 6): THROUGHPUT mode — the
        # decoder's GEMM kernels run one f16 MFMA per product, everything else as "f16x3" (fp32 master weights, fp32 accumulation,
        # the same power-of-two backward scale); not fp32-class, reported beside the headline, never instead of it
        self.prec = {"f32": _lib.PREC_F32, "f16x3": _lib.PREC_F16X3, "f16": _lib.PREC_F16}[prec]
        self.group = process_group
        self.overlap_all_reduce = overlap_all_reduce
        self.sync_bn = sync_bn          # BatchNorm statistics over the batches of all ranks (train.py --sync_bn)
        self._sync = None
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
        self._exp_avg = self._exp_avg_sq = None      # Adam moments: allocated on first use
        self.offsets, self._gmap, off = {}, {}, 0
        for k, p in zip(self.names, self.params):
            self.offsets[k] = off
            self._gmap[id(p)] = self.grad_flat.data_ptr() + 4 * off
            if bind_grads:   # param.grad are views into the flat buffer the library writes (optimisers see them)
                p.grad = self.grad_flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self._losses = torch.zeros(4, dtype=torch.float32, device=dev)
        self._ws = None
        self._buckets = None
        self._events = None
        self._comm_stream = None

    @property
    def exp_avg(self):
        if self._exp_avg is None:
            self._exp_avg = torch.zeros_like(self.grad_flat)
        return self._exp_avg

    @property
    def exp_avg_sq(self):
        if self._exp_avg_sq is None:
            self._exp_avg_sq = torch.zeros_like(self.grad_flat)
        return self._exp_avg_sq

    @staticmethod
    def _trainable(k):
        """Trainable tensors = everything the forward touches (reference: 14 tensors never get a grad — the dead
        att_layer.* twin and down5_.41.* — and vggptlossfunc.* is frozen)."""
        return not (k.startswith("att_layer.") or k.startswith("vggptlossfunc.") or ".down5_." in k)

    # -- struct builders ------------------------------------------------------------------------
    def _gptr(self, t):
        return self._gmap.get(id(t))

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
        if self.overlap_all_reduce and self._exchange_on():
            for k, e in enumerate(self._ddp_events()):
                tb.ev_grad_ready[k] = e.cuda_event
            self._events_armed = True
        self._attach_sync(tb)
        self._workspace(b, s, q, ns)
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

    # -- autograd-style halves of the step (s3d_train_fwd / s3d_train_bwd) ----------------------------
    def _batch_struct(self, batch, need_sdf):
        dev = self.grad_flat.device
        f = lambda k: batch[k].to(device=dev, dtype=torch.float32).contiguous()
        keys = ["img_input", "img_slices", "qry_norot", "obj_rot_mat", "trans_mat_wo_rot_tp"] + (["sdf"] if need_sdf else [])
        t = {k: f(k) for k in keys}
        tb = _lib.S3dTrainBatch()
        tb.img, tb.img_slices, tb.qry = t["img_input"].data_ptr(), t["img_slices"].data_ptr(), t["qry_norot"].data_ptr()
        tb.rot, tb.trans = t["obj_rot_mat"].data_ptr(), t["trans_mat_wo_rot_tp"].data_ptr()
        tb.sdf = t["sdf"].data_ptr() if need_sdf else None
        self._attach_sync(tb)
        return tb, t

    def _workspace(self, b, s, q, ns):
        nb = self.lib.s3d_train_workspace_bytes(b, s, q, ns)
        if self._ws is None or self._ws.numel() < nb:
            self._ws = None                      # release before growing (tens of GB at full size)
            self._ws = torch.empty(nb, dtype=torch.uint8, device=self.grad_flat.device)
        return self._ws

    def forward_only(self, batch):
        """Train-mode forward (models.py:48-94 with batch-statistics BatchNorm and dropout): returns
        (sdf_pred (B,Q), slices_rec (B,3*ns,S,S), vgg_loss ()) and a context for backward_from; every activation the
        backward needs stays in this trainer's workspace until the next forward."""
        m, lib, dev = self.model, self.lib, self.grad_flat.device
        tb, t = self._batch_struct(batch, need_sdf=False)
        b, _, s, _ = t["img_input"].shape
        q, ns = t["qry_norot"].shape[1], m.n_slices
        ws = self._workspace(b, s, q, ns)
        sdf_pred = torch.empty((b, q), dtype=torch.float32, device=dev)
        rec = torch.empty((b * ns, 3, s, s), dtype=torch.float32, device=dev)
        vgg = torch.empty((), dtype=torch.float32, device=dev)
        seed = self._next_seed()
        u, h, v = self._unet_struct(False), self._head_struct(False), self._vgg_struct()
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.s3d_train_fwd(C.byref(u), C.byref(h), C.byref(v), C.byref(tb), b, s, q, ns, float(self.dropout),
                                     seed, self.prec, vgg.data_ptr(), sdf_pred.data_ptr(), rec.data_ptr(),
                                     ws.data_ptr(), ws.numel(), stream), "s3d_train_fwd")
        m._packed_key = None   # BN running statistics changed in place: eval-mode packs are stale
        ctx = {"t": t, "dims": (b, s, q, ns), "seed": seed, "rec": rec, "dropout": float(self.dropout)}
        self._last_ctx = ctx
        return sdf_pred, rec.view(b, ns * 3, s, s), vgg, ctx

    _last_ctx = None

    def auto_grad_scale(self, d_sdf):
        """Power-of-two backward scale for the split-precision path from the incoming gradient itself: puts
        max|d sdf| in [8, 16) (one host sync; the fused step knows 1/n in advance).  0 = let the library choose."""
        if self.prec not in (_lib.PREC_F16X3, _lib.PREC_F16) or d_sdf is None:
            return 0.0
        mx = float(d_sdf.abs().max())
        if not (mx > 0.0) or mx != mx or mx == float("inf"):
            return 0.0
        import math
        return float(2.0 ** (3 - math.floor(math.log2(mx))))

    def backward_from(self, ctx, d_sdf=None, d_rec=None, d_vgg=0.0, grad_scale=0.0):
        """Backward of forward_only from the output gradients; WRITES the parameter gradients into grad_flat
        (param.grad views when bind_grads).  grad_scale: see s3d_train_bwd (0 = automatic)."""
        lib, dev = self.lib, self.grad_flat.device
        b, s, q, ns = ctx["dims"]
        t = ctx["t"]
        tb = _lib.S3dTrainBatch()
        tb.img, tb.img_slices, tb.qry = t["img_input"].data_ptr(), t["img_slices"].data_ptr(), t["qry_norot"].data_ptr()
        tb.rot, tb.trans = t["obj_rot_mat"].data_ptr(), t["trans_mat_wo_rot_tp"].data_ptr()
        self._attach_sync(tb)
        g = lambda x, shape: None if x is None else x.to(device=dev, dtype=torch.float32).reshape(shape).contiguous()
        d_sdf, d_rec = g(d_sdf, (b, q)), g(d_rec, (b * ns, 3, s, s))
        u, h, v = self._unet_struct(False), self._head_struct(False), self._vgg_struct()
        du, dh = self._unet_struct(True), self._head_struct(True)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.s3d_train_bwd(C.byref(u), C.byref(h), C.byref(v), C.byref(du), C.byref(dh), C.byref(tb), b, s, q,
                                     ns, ctx["dropout"], ctx["seed"], self.prec,
                                     d_sdf.data_ptr() if d_sdf is not None else None,
                                     d_rec.data_ptr() if d_rec is not None else None, float(d_vgg), float(grad_scale),
                                     ctx["rec"].data_ptr(), self._ws.data_ptr(), self._ws.numel(), stream),
                   "s3d_train_bwd")
        return self.grad_flat

    # -- cross-rank BatchNorm statistics ----------------------------------------------------------------
    def _sync_bn_struct(self):
        """S3dSyncBn descriptor (include/slice3d_hip.h) whose callback all-reduces a slice of a device scratch buffer
        with torch.distributed on the current stream — 60 small collectives per step (two per BatchNorm layer in the
        forward: means, then merged variances; one in the backward), issued from inside the library
        call between the kernel that produces the per-rank statistics and the one that consumes the global ones."""
        if not self.sync_bn or not self._exchange_on():
            return None
        if self._sync is None:
            import torch.distributed as dist
            buf = torch.zeros(2048, dtype=torch.float32, device=self.grad_flat.device)
            state = {"error": None}

            def all_reduce_sum(user, ptr, n, stream):
                try:
                    off = (ptr - buf.data_ptr()) // 4
                    dist.all_reduce(buf[off:off + n], op=dist.ReduceOp.SUM, group=self.group)
                    return 0
                except Exception as e:      # never let an exception cross the C frame
                    state["error"] = e
                    return 1
            cb = _lib.ALL_REDUCE_SUM_FN(all_reduce_sum)
            st = _lib.S3dSyncBn()
            st.all_reduce_sum, st.user, st.world_size, st.scratch = cb, None, self._world(), buf.data_ptr()
            self._sync = (st, cb, buf, state)          # keep the callback and the buffer alive
        return self._sync[0]

    def _attach_sync(self, tb):
        st = self._sync_bn_struct()
        if st is not None:
            tb.sync_bn = C.pointer(st)

    # -- data-parallel exchange step ------------------------------------------------------------------
    def _exchange_on(self):
        """True when the collectives of the step must be issued: more than one rank — or S3D_FORCE_COLLECTIVES=1 inside an
        initialised process group of ONE rank (TEST-ONLY switch, read here so that tests/test_gpu_rccl.py can drive the
        production step unchanged; it has no effect at world size > 1 and must not be set in a real job: pushes the bucketed gradient all-reduces through the
        event-ordered side stream and the sync-BN callback's collectives through RCCL on a single GPU; every sum over one
        rank is the identity, so the step's results do not change)."""
        if self._world() > 1:
            return True
        import os
        import torch.distributed as dist
        return (os.environ.get("S3D_FORCE_COLLECTIVES") == "1" and self.group is not False
                and dist.is_available() and dist.is_initialized())

    def _world(self):
        import torch.distributed as dist
        if self.group is False:          # process_group=False: a lone replica inside a distributed job (no exchange)
            return 1
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group)
        return 1

    def _ddp_events(self):
        """hipEvent handles the library records when a gradient bucket is final (S3dTrainBatch.ev_grad_ready)."""
        if self._events is None:
            dev = self.grad_flat.device
            self._events = [torch.cuda.Event() for _ in range(3)]
            for e in self._events:
                e.record(torch.cuda.current_stream(dev))      # instantiates the underlying hipEvent
            self._comm_stream = torch.cuda.Stream(device=dev)
            self._buckets = bucket_ranges(self.names, [p.numel() for p in self.params])
        return self._events

    def all_reduce_grads(self):
        """Data-parallel exchange step (replaces train.py:131-132): mean of the gradients over ranks.
        After a forward_backward that recorded the bucket events, buckets 0-2 (decoder 7.6 MB, U-Net decoder half,
        deep encoder 52 MB) are all-reduced on a side stream as the backward finishes them, under the remaining
        backward; the shallow-encoder bucket follows the last kernel.  Otherwise: one flat all-reduce."""
        import torch.distributed as dist
        world = self._world()
        if not self._exchange_on():
            return
        self.lib.s3d_range_push(b"s3d:train:grad_all_reduce")       # roctx: the exchange step as a trace range
        try:
            if not (self.overlap_all_reduce and self._events_armed):      # one flat all-reduce (parallel.all_reduce_mean_'s body;
                dist.all_reduce(self.grad_flat, op=dist.ReduceOp.SUM, group=self.group)   # issued at world size 1 too when forced)
                self.grad_flat.div_(world)
                return
            main = torch.cuda.current_stream(self.grad_flat.device)
            comm = self._comm_stream
            with torch.cuda.stream(comm):
                for k, (lo, hi) in enumerate(self._buckets):
                    if k < 3:
                        comm.wait_event(self._events[k])
                    else:
                        comm.wait_stream(main)
                    dist.all_reduce(self.grad_flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
            main.wait_stream(comm)
            self.grad_flat.div_(world)
            self._events_armed = False
        finally:
            self.lib.s3d_range_pop()

    _events_armed = False

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
        self._attach_sync(tb)
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
