"""UNetModel — MI355X-native counterpart of the latent-diffusion denoising U-Net
(gen_slices/ldm/modules/diffusionmodules/openaimodel.py:413-757, configured by
configs/latent-diffusion/objaverse-ldm-kl-8.yaml:22-34; BASELINE configs[4]).

Inference only.  Same constructor arguments for the options that configuration uses, the same module tree and
therefore the same state_dict keys as the reference (time_embed.*, input_blocks.N.M.{in_layers,emb_layers,
out_layers,skip_connection,norm,qkv,proj_out}.*, middle_block.*, output_blocks.*, out.*), and the same
forward(x, timesteps, c_fmaps=...) contract (NCHW in / NCHW out).  Every tensor op is an entry point of
libslice3d_hip.so ("Latent-diffusion denoising U-Net primitives" in include/slice3d_hip.h): activations live
channels-last, 3x3 convolutions run the LDS-staged split-precision kernel, the ResBlock residual add and the
AttentionBlock residual are conv epilogues.  There is no
CPU fallback.

Not built (the configuration does not use them): class conditioning, SpatialTransformer / cross-attention,
use_new_attention_order, conv_resample down/up-sampling, fp16 torso, dims != 2.
"""
import ctypes as C

import os

import torch
import torch.nn as nn

from . import _lib


def _gn(ch):
    return nn.GroupNorm(32, ch)


class ResBlock(nn.Module):
    """openaimodel.py:160-275 (use_scale_shift_norm / up / down variants)."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_scale_shift_norm=False, up=False,
                 down=False):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels or channels
        self.use_scale_shift_norm, self.up, self.down = use_scale_shift_norm, up, down
        self.in_layers = nn.Sequential(_gn(channels), nn.SiLU(), nn.Conv2d(channels, self.out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, 2 * self.out_channels
                                                             if use_scale_shift_norm else self.out_channels))
        self.out_layers = nn.Sequential(_gn(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        nn.Conv2d(self.out_channels, self.out_channels, 3, padding=1))
        self.skip_connection = (nn.Identity() if self.out_channels == channels
                                else nn.Conv2d(channels, self.out_channels, 1))
        self.cat_split = None     # (C_h, C_skip) when the block's input is th.cat([h, hs.pop()]) (output blocks)


class AttentionBlock(nn.Module):
    """openaimodel.py:278-331 with QKVAttentionLegacy."""

    def __init__(self, channels, num_heads):
        super().__init__()
        self.channels, self.num_heads = channels, num_heads
        self.norm = _gn(channels)
        self.qkv = nn.Conv1d(channels, channels * 3, 1)
        self.proj_out = nn.Conv1d(channels, channels, 1)


class UNetModel(nn.Module):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), num_heads=-1, use_scale_shift_norm=False, resblock_updown=False,
                 backend="hip", prec="f16x3", fuse_gn=True, branch_streams=False, defer_finish=False):
        super().__init__()
        if not resblock_updown:
            raise NotImplementedError("conv_resample down/up-sampling is not built (the Slice3D configuration uses "
                                      "resblock_updown=True)")
        if num_heads < 1:
            raise ValueError("num_heads must be set")
        self.image_size, self.in_channels, self.model_channels = image_size, in_channels, model_channels
        self.out_channels, self.num_heads, self.prec, self.backend = out_channels, num_heads, prec, backend
        # a ResBlock's 1x1 skip_connection does not depend on its GroupNorm -> conv chain: with branch_streams it runs on a side
        # stream (a parallel branch of the captured HIP graph) beside the chain's small kernels (openaimodel.py:240-246,
        # :272-275).  OFF by default: measured SLOWER on MI355X — 4.45 against 4.18 ms per step at batch 1, with runs of 7-8 ms
        # (tools/ldm_ab.py, profiles/r05_ldm_experiments.md): the graph's cross-branch dependencies cost more than the 17
        # overlapped 14 us kernels return
        self.branch_streams = branch_streams
        self._side = None
        self._ws_side = None
        # defer_finish: a fused-GroupNorm 3x3 convolution that ran split-K leaves its raw partial sums to the NEXT GroupNorm's
        # statistics kernel, which adds them up (+ bias + residual), stores the finished tensor and goes on with the values:
        # the convolution's own finish pass (one ~5 us launch behind ~60 of the step's 70 3x3 convolutions) is gone.  OFF by
        # default: bit-identical and measured 0.6 % SLOWER (4.203 against 4.179 ms per step, tools/ldm_ab.py,
        # profiles/r05_ldm_experiments.md) — the statistics kernels, latency-bound themselves, pay for the 5-14 partials per
        # element what the finish launches cost
        self.defer_finish = defer_finish
        self._pending = None      # (tensor, S3dConvPartial, (n, h, w), keep-alive refs) of the one deferred output, or None
        self.fuse_gn = fuse_gn    # GroupNorm -> SiLU -> conv3x3 as one operator where the kernel serves the shape (s3d_conv_gn_fwd)
        ted = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
        self.input_blocks = nn.ModuleList([nn.Sequential(nn.Conv2d(in_channels, model_channels, 3, padding=1))])
        chans, ch, ds = [model_channels], model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [ResBlock(ch, ted, dropout, mult * model_channels, use_scale_shift_norm)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(AttentionBlock(ch, num_heads))
                self.input_blocks.append(nn.Sequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(nn.Sequential(ResBlock(ch, ted, dropout, ch, use_scale_shift_norm, down=True)))
                chans.append(ch)
                ds *= 2
        self.middle_block = nn.Sequential(ResBlock(ch, ted, dropout, None, use_scale_shift_norm),
                                          AttentionBlock(ch, num_heads),
                                          ResBlock(ch, ted, dropout, None, use_scale_shift_norm))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [ResBlock(ch + ich, ted, dropout, model_channels * mult, use_scale_shift_norm)]
                layers[0].cat_split = (ch, ich)
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(AttentionBlock(ch, num_heads))
                if level and i == num_res_blocks:
                    layers.append(ResBlock(ch, ted, dropout, ch, use_scale_shift_norm, up=True))
                    ds //= 2
                self.output_blocks.append(nn.Sequential(*layers))
        self.out = nn.Sequential(_gn(ch), nn.SiLU(), nn.Conv2d(model_channels, out_channels, 3, padding=1))
        self._lib = _lib.load() if backend == "hip" else None
        self._attn_ws = {}   # device -> scratch of the long-sequence attention kernel (pre-split K / V block images)
        # 48-wide heads at >= 1 024 tokens on the key-split f16-MFMA kernel (ldm_attn.hip la_attention2_kernel); 0 = the
        # fp32-MFMA kernel of ldm_ops.hip they ran on before round 6 (kept for the A/B)
        self.wide_head_mfma = os.environ.get("S3D_LDM_WIDE_MFMA", "1") != "0"
        self._packed = {}
        self._packed_key = None
        self._ws = None

    # ------------------------------------------------------------------------------------------
    def _require_lib(self):
        if self._lib is None:
            raise _lib.S3dError("UNetModel(backend=%r) cannot compute: the HIP library is required; there is no CPU "
                                "fallback in the product path" % self.backend)
        return self._lib

    def _dev(self):
        return self.time_embed[0].weight.device

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self._dev()).cuda_stream)

    def _precv(self):
        return {"f16x3": _lib.PREC_F16X3, "f16": _lib.PREC_F16, "f32": _lib.PREC_F32}[self.prec]

    def _attn_precv(self):
        # prec='f16' (single-pass convolutions, the throughput mode): the attention operators keep their split-precision form
        return _lib.PREC_F32 if self.prec == "f32" else _lib.PREC_F16X3

    def _params_key(self):
        # fuse_gn / prec decide the two-source packing of the output blocks' first convolutions: toggling either after a
        # repack() must repack (a stale split made s3d_conv_gn_fwd refuse the call)
        return (self.fuse_gn, self.prec) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    @staticmethod
    def _pad16(c):
        return (c + 15) // 16 * 16

    def _pack_conv(self, conv, split=None):
        """packed weight image of a Conv2d / Conv1d(k=1); split = (cin0, cin1) for a two-source convolution."""
        lib = self._lib
        w = conv.weight
        cout, cin = w.shape[0], w.shape[1]
        ks = w.shape[2] if w.dim() == 4 else 1
        cin0, cin1 = split if split else (cin, 0)
        nb = lib.s3d_conv_packed_bytes(cout, cin0, cin1, ks)
        buf = torch.empty(nb, dtype=torch.uint8, device=w.device)
        _lib.check(lib.s3d_conv_pack(w.data_ptr(), conv.bias.data_ptr() if conv.bias is not None else None, cout, cin0,
                                     cin1, ks, buf.data_ptr(), nb, self._stream()), "s3d_conv_pack")
        return (buf, cout, cin0, cin1, ks)

    def repack(self):
        self._require_lib()
        if self._dev().type != "cuda":
            raise _lib.S3dError("move the model to the GPU (model.cuda())")
        self._packed = {}
        for mod in self.modules():
            if isinstance(mod, ResBlock):
                # an output block's first convolution reads cat([h, skip]): packed as a two-source K loop when the fused
                # GroupNorm + convolution operator serves the block (the sources are then normalised as they are staged),
                # as one source behind the stand-alone GroupNorm otherwise (its output is one tensor)
                cw = mod.in_layers[2].weight
                split = mod.cat_split
                if split is not None and not (not (mod.up or mod.down)
                                              and self._gn_conv_served(cw.shape[0], split[0], split[1], cw.shape[2])):
                    split = None
                self._packed[id(mod.in_layers[2])] = self._pack_conv(mod.in_layers[2], split=split)
                self._packed[id(mod.out_layers[3])] = self._pack_conv(mod.out_layers[3])
                if isinstance(mod.skip_connection, nn.Conv2d):   # two-source K loop over (h, skip) for the output blocks
                    self._packed[id(mod.skip_connection)] = self._pack_conv(mod.skip_connection, split=mod.cat_split)
            elif isinstance(mod, AttentionBlock):
                self._packed[id(mod.qkv)] = self._pack_conv(mod.qkv)
                self._packed[id(mod.proj_out)] = self._pack_conv(mod.proj_out)
        self._packed[id(self.input_blocks[0][0])] = self._pack_conv(self.input_blocks[0][0])
        self._packed[id(self.out[2])] = self._pack_conv(self.out[2])
        # all ResBlock.emb_layers read the same timestep embedding: one stacked GEMV instead of one launch per block
        res = [m for m in self.modules() if isinstance(m, ResBlock)]
        self._film_w = torch.cat([m.emb_layers[1].weight for m in res], 0).contiguous()
        self._film_b = torch.cat([m.emb_layers[1].bias for m in res], 0).contiguous()
        self._film_off, off = {}, 0
        for m in res:
            rows = m.emb_layers[1].weight.shape[0]
            self._film_off[id(m)] = (off, rows)
            off += rows
        self._packed_key = self._params_key()

    # ------------------------------------------------------------------------------------------
    # primitive wrappers (channels-last tensors)
    # ------------------------------------------------------------------------------------------
    # -- deferred split-K finish ------------------------------------------------------------------------
    def _take_pending(self, x):
        """The S3dConvPartial of x if x is the tensor whose finish pass is still owed (and hand the duty to the caller)."""
        p = self._pending
        if p is not None and p[0] is x:
            self._pending = None
            return p
        return None

    def _finish_pending(self):
        """Run the owed finish pass on its own (a consumer that is not a GroupNorm comes first, or the workspace is needed)."""
        p = self._pending
        if p is not None:
            self._pending = None
            _, desc, (n, h, w), _ = p
            _lib.check(self._lib.s3d_conv_finish_fwd(C.byref(desc), n, h, w, self._stream()), "s3d_conv_finish_fwd")

    def _conv(self, conv, x0, x1=None, residual=None, side=False):
        self._finish_pending()
        lib = self._lib
        buf, cout, cin0, cin1, ks = self._packed[id(conv)]
        n, h, w, _ = x0.shape
        out = torch.empty((n, h, w, cout), dtype=torch.float32, device=x0.device)
        if self._ws is None:
            self._ws = torch.empty(8 << 20, dtype=torch.float32, device=x0.device)     # split-K scratch
        ws = self._ws
        if side:     # a convolution on the side stream must not share the main chain's split-K scratch
            if self._ws_side is None:
                self._ws_side = torch.empty(2 << 20, dtype=torch.float32, device=x0.device)
            ws = self._ws_side
        _lib.check(lib.s3d_conv_fwd(buf.data_ptr(), x0.data_ptr(), x1.data_ptr() if x1 is not None else None,
                                    residual.data_ptr() if residual is not None else None, out.data_ptr(), n, h, w, cout,
                                    cin0, cin1, ks, self._precv(), ws.data_ptr(), ws.numel() * 4,
                                    self._stream()), "s3d_conv_fwd")
        return out

    def _skip_branch(self, blk, xs, skip):
        """The ResBlock's 1x1 skip_connection on the side stream (forked behind everything the current stream has queued,
        i.e. behind the producers of xs / skip); returns (tensor, join) — call join() on the main stream before the tensor is
        read there.  Inside a HIP-graph capture the fork / join become graph edges: the branch runs beside the chain."""
        cur = torch.cuda.current_stream(xs.device)
        if self._side is None:
            self._side = torch.cuda.Stream(xs.device)
        side = self._side
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            res = self._conv(blk.skip_connection, xs, x1=skip, side=True)
        for t in (xs, skip):                  # allocated on the main stream, read on the side stream
            if t is not None:
                t.record_stream(side)

        def join():
            cur.wait_stream(side)
            res.record_stream(cur)            # allocated on the side stream, read on the main stream
        return res, join

    def _gn_conv_served(self, cout, cin0, cin1, ks, h=0, w=0):
        """The C side's own statement of what s3d_conv_gn_fwd serves (s3d_conv_gn_supported) — the one source of both the
        two-source packing decision of repack() and the fused / unfused choice of forward()."""
        return bool(self.fuse_gn and self.prec != "f32" and
                    self._lib.s3d_conv_gn_supported(cout, cin0, cin1, ks, self._precv(), h, w, 1))

    def _gn_conv_fusable(self, conv, x):
        """GroupNorm -> [FiLM] -> SiLU -> conv3x3 as one operator (s3d_conv_gn_fwd)."""
        _, cout, cin0, cin1, ks = self._packed[id(conv)]
        return (self._gn_conv_served(cout, cin0, cin1, ks, x.shape[1], x.shape[2])
                and x.shape[-1] == cin0)   # (a concatenated input needs the two-source pack)

    def _gn_table(self, gn, x, x1=None, film=None):
        """Statistics of group_norm(cat([x, x1])) folded with gamma / beta / FiLM into the per-channel affine table a fused
        convolution applies (s3d_group_norm_table_fwd).  If x is the deferred output of a split-K convolution this launch is
        also its finish pass (it sums the partials, adds bias and residual and stores x)."""
        lib = self._lib
        n, h, w, c = x.shape
        c1 = x1.shape[-1] if x1 is not None else 0
        stats = torch.empty(lib.s3d_group_norm_stats_floats(n, gn.num_groups), dtype=torch.float32, device=x.device)
        table = torch.empty((n, 2, c + c1), dtype=torch.float32, device=x.device)
        fp, fld = (None, 0)
        if film is not None:
            ft, off = film
            fp, fld = ft.data_ptr() + 4 * off, ft.shape[1]
        pend = self._take_pending(x)
        if pend is None:
            self._finish_pending()
        _lib.check(lib.s3d_group_norm_table_fwd(x.data_ptr(), c, x1.data_ptr() if x1 is not None else None, c1,
                                                gn.weight.data_ptr(), gn.bias.data_ptr(), fp, fld, table.data_ptr(),
                                                stats.data_ptr(), n, h * w, gn.num_groups, C.c_float(gn.eps),
                                                C.byref(pend[1]) if pend is not None else None, self._stream()),
                   "s3d_group_norm_table_fwd")
        return table

    def _gn_conv_apply(self, conv, x, x1, table, residual=None):
        """conv(silu(x * A + B)) (+ residual) with the table of _gn_table: the normalised tensor is never written
        (s3d_conv_gn_fwd; openaimodel.py:188-194, :229-236, :262-270).  With defer_finish a split-K launch leaves its finish
        pass to the next consumer (self._pending)."""
        lib = self._lib
        buf, cout, cin0, cin1, ks = self._packed[id(conv)]
        n, h, w, c = x.shape
        c1 = x1.shape[-1] if x1 is not None else 0
        if (c, c1) != (cin0, cin1):
            raise _lib.S3dError("fused GroupNorm convolution: sources (%d, %d) do not match the packed split (%d, %d)" % (c, c1, cin0, cin1))
        self._finish_pending()           # (the split-K scratch is about to be overwritten)
        out = torch.empty((n, h, w, cout), dtype=torch.float32, device=x.device)
        if self._ws is None:
            self._ws = torch.empty(8 << 20, dtype=torch.float32, device=x.device)     # split-K scratch
        nsplit = C.c_int(1)
        _lib.check(lib.s3d_conv_gn_fwd(buf.data_ptr(), x.data_ptr(), x1.data_ptr() if x1 is not None else None,
                                       residual.data_ptr() if residual is not None else None, out.data_ptr(), n, h, w, cout,
                                       cin0, cin1, ks, self._precv(), table.data_ptr(), 1,
                                       self._ws.data_ptr(), self._ws.numel() * 4,
                                       C.byref(nsplit) if self.defer_finish else None, self._stream()), "s3d_conv_gn_fwd")
        if nsplit.value > 1:             # `out` is still to be written: by the next GroupNorm, or by _finish_pending()
            desc = _lib.S3dConvPartial(self._ws.data_ptr(), nsplit.value, buf.data_ptr(), cout, cin0, cin1, ks,
                                       residual.data_ptr() if residual is not None else None, out.data_ptr())
            self._pending = (out, desc, (n, h, w), (residual, buf))
        return out

    def _group_norm(self, gn, x, film=None, silu=True, x1=None):
        """GroupNorm(+FiLM)(+SiLU) of x, or of the channel concatenation [x, x1] (never materialised)."""
        lib = self._lib
        n, h, w, c = x.shape
        stats = torch.empty(lib.s3d_group_norm_stats_floats(n, gn.num_groups), dtype=torch.float32, device=x.device)
        pend = self._take_pending(x) if x1 is None else None
        if pend is not None:             # x's split-K partials: the statistics kernel of this GroupNorm is its finish pass
            y = torch.empty_like(x)
            fp, fld = (None, 0)
            if isinstance(film, tuple):
                fp, fld = film[0].data_ptr() + 4 * film[1], film[0].shape[1]
            elif film is not None:
                fp, fld = film.data_ptr(), film.shape[1]
            _lib.check(lib.s3d_group_norm_partial_fwd(C.byref(pend[1]), gn.weight.data_ptr(), gn.bias.data_ptr(), fp, fld,
                                                      y.data_ptr(), stats.data_ptr(), n, h * w, gn.num_groups,
                                                      C.c_float(gn.eps), 1 if silu else 0, self._stream()),
                       "s3d_group_norm_partial_fwd")
            return y
        self._finish_pending()
        if x1 is not None:
            c1 = x1.shape[-1]
            y = torch.empty((n, h, w, c + c1), dtype=torch.float32, device=x.device)
            _lib.check(lib.s3d_group_norm2_fwd(x.data_ptr(), c, x1.data_ptr(), c1, gn.weight.data_ptr(),
                                               gn.bias.data_ptr(), film.data_ptr() if film is not None else None,
                                               y.data_ptr(), stats.data_ptr(), n, h * w, gn.num_groups,
                                               C.c_float(gn.eps), 1 if silu else 0, self._stream()),
                       "s3d_group_norm2_fwd")
            return y
        y = torch.empty_like(x)
        if isinstance(film, tuple):     # (stacked film tensor, first column)
            ft, off = film
            _lib.check(lib.s3d_group_norm_film_fwd(x.data_ptr(), gn.weight.data_ptr(), gn.bias.data_ptr(),
                                                   ft.data_ptr() + 4 * off, ft.shape[1], y.data_ptr(), stats.data_ptr(), n, h * w, c,
                                                   gn.num_groups, C.c_float(gn.eps), 1 if silu else 0, self._stream()),
                       "s3d_group_norm_film_fwd")
            return y
        _lib.check(lib.s3d_group_norm_fwd(x.data_ptr(), gn.weight.data_ptr(), gn.bias.data_ptr(),
                                          film.data_ptr() if film is not None else None, y.data_ptr(), stats.data_ptr(),
                                          n, h * w, c, gn.num_groups, C.c_float(gn.eps), 1 if silu else 0,
                                          self._stream()), "s3d_group_norm_fwd")
        return y

    def _resample(self, x, up):
        self._finish_pending()
        lib = self._lib
        n, h, w, c = x.shape
        y = torch.empty((n, h * 2, w * 2, c) if up else (n, h // 2, w // 2, c), dtype=torch.float32, device=x.device)
        _lib.check(lib.s3d_resample2x_fwd(x.data_ptr(), y.data_ptr(), n, h, w, c, 1 if up else 0, self._stream()),
                   "s3d_resample2x_fwd")
        return y

    def _linear(self, lin, x, silu_in):
        lib = self._lib
        n, k = x.shape
        m = lin.weight.shape[0]
        out = torch.empty((n, m), dtype=torch.float32, device=x.device)
        _lib.check(lib.s3d_small_linear_fwd(x.data_ptr(), lin.weight.data_ptr(), lin.bias.data_ptr(), out.data_ptr(), n,
                                            k, m, 1 if silu_in else 0, self._stream()), "s3d_small_linear_fwd")
        return out

    def _add(self, a, b):
        self._finish_pending()
        lib = self._lib
        out = torch.empty_like(a)
        _lib.check(lib.s3d_add_fwd(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), self._stream()), "s3d_add_fwd")
        return out

    def _add_nchw(self, a, b):
        """h + c_fmap (openaimodel.py:735-746) with the feature map read in the reference's NCHW layout: one launch."""
        self._finish_pending()
        b = b.to(device=self._dev(), dtype=torch.float32).contiguous()
        n, c, h, w = b.shape
        if tuple(a.shape) != (n, h, w, c):
            return self._add(a, self._to_nhwc(b))
        out = torch.empty_like(a)
        _lib.check(self._lib.s3d_add_nchw_fwd(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, c, h, w, self._stream()), "s3d_add_nchw_fwd")
        return out

    def _to_nhwc(self, x, cpad=None):
        lib = self._lib
        x = x.to(device=self._dev(), dtype=torch.float32).contiguous()
        n, c, h, w = x.shape
        cpad = cpad or c
        out = torch.empty((n, h, w, cpad), dtype=torch.float32, device=x.device)
        _lib.check(lib.s3d_nchw_to_nhwc_pad(x.data_ptr(), out.data_ptr(), n, c, h, w, cpad, self._stream()),
                   "s3d_nchw_to_nhwc_pad")
        return out

    # ------------------------------------------------------------------------------------------
    # blocks
    # ------------------------------------------------------------------------------------------
    def _res_block(self, blk, x, emb, skip=None):
        """ResBlock._forward (openaimodel.py:253-275); x (and skip: the block input is cat([x, skip]))."""
        # th.cat([h, hs.pop()], dim=1) (openaimodel.py:750) is never built: the GroupNorm reads both tensors (its groups
        # straddle them) and the 1x1 skip_connection walks them as the two sources of its K loop
        if not blk.use_scale_shift_norm:
            raise NotImplementedError("ResBlock without use_scale_shift_norm is not built")
        resampling = blk.up or blk.down
        if resampling and skip is not None:
            raise NotImplementedError("resampling ResBlock on a concatenated input (not in this architecture)")
        # 1. the GroupNorm of the input FIRST: if x is a deferred split-K output its statistics launch finishes it
        fuse_in = not resampling and self._gn_conv_fusable(blk.in_layers[2], x)
        if fuse_in:
            table_in = self._gn_table(blk.in_layers[0], x, x1=skip)
        else:
            hn = self._group_norm(blk.in_layers[0], x, silu=True, x1=skip)
        # 2. the skip path: it only needs the block's (finished) input
        xs = self._resample(x, blk.up) if resampling else x
        join = None
        if isinstance(blk.skip_connection, nn.Conv2d):
            if self.branch_streams:
                res, join = self._skip_branch(blk, xs, skip)
            else:
                res = self._conv(blk.skip_connection, xs, x1=skip)
        elif skip is not None:
            raise NotImplementedError("identity skip connection on a concatenated input (not in this architecture)")
        else:
            res = xs
        # 3. in_layers convolution (GroupNorm -> SiLU -> resample -> conv: the resampling sits between, nothing to fuse)
        if fuse_in:
            h = self._gn_conv_apply(blk.in_layers[2], x, skip, table_in)
        else:
            h = self._conv(blk.in_layers[2], self._resample(hn, blk.up) if resampling else hn)
        # 4. out_layers; (N, 2*Cout) = scale | shift: columns [off, off + rows) of the stacked emb_layers output, read in place
        off, rows = self._film_off[id(blk)]
        if self._gn_conv_fusable(blk.out_layers[3], h):
            table_out = self._gn_table(blk.out_layers[0], h, film=(self._film_all, off))
            if join is not None:
                join()
            return self._gn_conv_apply(blk.out_layers[3], h, None, table_out, residual=res)
        h = self._group_norm(blk.out_layers[0], h, film=(self._film_all, off), silu=True)
        if join is not None:
            join()
        return self._conv(blk.out_layers[3], h, residual=res)

    def _attention_block(self, blk, x):
        """AttentionBlock._forward (openaimodel.py:309-315)."""
        lib = self._lib
        n, h, w, c = x.shape
        hn = self._group_norm(blk.norm, x, silu=False)
        qkv = self._conv(blk.qkv, hn)                                        # (N, H, W, 3C), heads x (q|k|v) x ch
        att = torch.empty((n, h, w, c), dtype=torch.float32, device=x.device)
        ch = c // blk.num_heads
        # long sequences of narrow heads: the f16-MFMA kernel with fp32-class logits and pre-split K / V (ldm_attn.hip)
        use_mfma = self._attn_precv() == _lib.PREC_F16X3
        ws_bytes = lib.s3d_qkv_attention_ws_bytes(n, h * w, blk.num_heads, ch) if use_mfma else 0
        if ws_bytes and h * w >= 1024 and (ch <= 32 or self.wide_head_mfma):
            ws = self._attn_ws.get(x.device)
            if ws is None or ws.numel() < ws_bytes:
                ws = self._attn_ws[x.device] = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
            _lib.check(lib.s3d_qkv_attention_ws_fwd(qkv.data_ptr(), att.data_ptr(), n, h * w, blk.num_heads, ch, ws.data_ptr(),
                                                    ws_bytes, self._stream()), "s3d_qkv_attention_ws_fwd")
        else:
            _lib.check(lib.s3d_qkv_attention_fwd(qkv.data_ptr(), att.data_ptr(), n, h * w, blk.num_heads, ch, self._attn_precv(),
                                                 self._stream()), "s3d_qkv_attention_fwd")
        return self._conv(blk.proj_out, att, residual=x)

    def _run(self, seq, h, emb, skip=None):
        for mod in seq:
            if isinstance(mod, ResBlock):
                h = self._res_block(mod, h, emb, skip)
                skip = None
            elif isinstance(mod, AttentionBlock):
                h = self._attention_block(mod, h)
            else:   # the stem convolution
                h = self._conv(mod, h)
        return h

    def forward(self, x, timesteps=None, context=None, y=None, c_fmaps=None, **kwargs):
        """openaimodel.py:710-757: x (N, in_channels, H, W), timesteps (N,), c_fmaps {'f1'..'f5'} NCHW feature maps
        added after input blocks 0, 4, 7, 10, 12 -> (N, out_channels, H, W)."""
        lib = self._require_lib()
        if self.training:
            raise RuntimeError("UNetModel computes the eval-mode forward; call model.eval()")
        if context is not None or y is not None:
            raise NotImplementedError("cross-attention context / class labels are not built")
        if self._packed_key is None or self._packed_key != self._params_key():
            self.repack()
        dev = self._dev()
        n = x.shape[0]
        self._pending = None
        t = timesteps.to(device=dev, dtype=torch.float32).contiguous()
        t_emb = torch.empty((n, self.model_channels), dtype=torch.float32, device=dev)
        _lib.check(lib.s3d_timestep_embedding_fwd(t.data_ptr(), t_emb.data_ptr(), n, self.model_channels,
                                                  C.c_float(10000.0), self._stream()), "s3d_timestep_embedding_fwd")
        emb = self._linear(self.time_embed[2], self._linear(self.time_embed[0], t_emb, silu_in=False), silu_in=True)
        # emb_layers of every ResBlock (SiLU + Linear on the same emb, openaimodel.py:222-228,262) in one launch
        m_all = self._film_w.shape[0]
        self._film_all = torch.empty((n, m_all), dtype=torch.float32, device=dev)
        _lib.check(lib.s3d_small_linear_fwd(emb.data_ptr(), self._film_w.data_ptr(), self._film_b.data_ptr(),
                                            self._film_all.data_ptr(), n, emb.shape[1], m_all, 1, self._stream()),
                   "s3d_small_linear_fwd")
        inject = {0: "f1", 4: "f2", 7: "f3", 10: "f4", 12: "f5"}
        h = self._to_nhwc(x, self._pad16(self.in_channels))
        hs = []
        for m_id, module in enumerate(self.input_blocks):
            h = self._run(module, h, emb)
            if c_fmaps is not None and m_id in inject:
                h = self._add_nchw(h, c_fmaps[inject[m_id]])
            hs.append(h)
        h = self._run(self.middle_block, h, emb)
        for module in self.output_blocks:
            h = self._run(module, h, emb, skip=hs.pop())
        hn = self._group_norm(self.out[0], h, silu=True)
        o = self._conv(self.out[2], hn)                                       # (N, H, W, out_channels)
        self._finish_pending()                                                # (nothing is owed here: the stem of `out` is not split-K)
        nn_, hh, ww, cc = o.shape
        out = torch.empty((nn_, cc, hh, ww), dtype=torch.float32, device=dev)
        _lib.check(lib.s3d_nhwc_to_nchw(o.data_ptr(), out.data_ptr(), nn_, cc, hh, ww, self._stream()), "s3d_nhwc_to_nchw")
        return out
