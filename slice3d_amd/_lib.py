"""ctypes binding of libslice3d_hip.so (C ABI declared in include/slice3d_hip.h).

The library is built in-tree by `make -C slice3d_amd/csrc` (see __graft_entry__.build()).  There is NO
fallback: if the shared object is missing or a call fails, this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("S3D_HIP_LIB") or os.path.join(_HERE, "csrc", "libslice3d_hip.so")   # override: kernel experiments

N_LEVELS = 5
N_LAYERS = 3
PREC_F32 = 0
PREC_F16X3 = 1
PREC_F16 = 2          # single-pass f16 MFMA: throughput mode, not fp32-class (inference only)
PREC_BF16 = 3         # the decoder's attention / FFN GEMMs on the bf16 MFMA, the rest as PREC_F16 (Slices3DRegModel inference only)
PROF_UNET, PROF_LATENT, PROF_SAMPLE, PROF_ATTN, PROF_FFN, PROF_FFN_FINAL, PROF_VGG, PROF_SAMPLE_PYR = range(8)
PROF_NAMES = ("unet_encode", "latent_build", "sample_tokens", "attn_layer", "ffn_layer", "ffn_final", "vgg_loss",
              "sample_pyramid")

c_float_p = C.POINTER(C.c_float)


class S3dConvParams(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("bn", C.c_void_p * 4)]


class S3dUNetParams(C.Structure):
    _fields_ = [("enc", S3dConvParams * 13), ("trans_c", S3dConvParams),
                ("trans_up", S3dConvParams * 4), ("up_t", S3dConvParams * 4),
                ("up_c1", S3dConvParams * 4), ("up_c2", S3dConvParams * 4),
                ("outc", S3dConvParams), ("emds", C.c_void_p), ("n_slices", C.c_int)]


class S3dPyramid(C.Structure):
    _fields_ = [("level", C.c_void_p * N_LEVELS), ("n_img", C.c_int), ("size", C.c_int)]


class S3dLayerParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "lin1_w", "lin1_b", "lin2_w",
                 "lin2_b", "norm1_w", "norm1_b", "norm2_w", "norm2_b")]


class S3dHeadParams(C.Structure):
    _fields_ = [("fc_p_w", C.c_void_p), ("fc_p_b", C.c_void_p), ("fc_s_w", C.c_void_p),
                ("fc_s_b", C.c_void_p), ("layer", S3dLayerParams * N_LAYERS),
                ("fc_out_w", C.c_void_p), ("fc_out_b", C.c_void_p)]


class S3dConvPartial(C.Structure):
    _fields_ = [("part", C.c_void_p), ("nsplit", C.c_int), ("conv_packed", C.c_void_p), ("cout", C.c_int), ("cin0", C.c_int),
                ("cin1", C.c_int), ("ks", C.c_int), ("residual", C.c_void_p), ("out", C.c_void_p)]


class S3dVggParams(C.Structure):
    _fields_ = [("conv", S3dConvParams * 14), ("mean", C.c_void_p), ("std", C.c_void_p)]


ALL_REDUCE_SUM_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p)


class S3dSyncBn(C.Structure):
    _fields_ = [("all_reduce_sum", ALL_REDUCE_SUM_FN), ("user", C.c_void_p), ("world_size", C.c_int),
                ("scratch", C.c_void_p)]


class S3dTrainBatch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("img", "img_slices", "qry", "rot", "trans", "sdf")] + \
               [("ev_grad_ready", C.c_void_p * 3), ("sync_bn", C.POINTER(S3dSyncBn))]


class S3dVgg16BnParams(C.Structure):
    _fields_ = [("conv", S3dConvParams * 13)]


class S3dGtPyramid(C.Structure):
    _fields_ = [("level", C.c_void_p * 5), ("n_img", C.c_int), ("size", C.c_int)]


class S3dGtHeadParams(C.Structure):
    _fields_ = [("pts_w", C.c_void_p * 3), ("pts_b", C.c_void_p * 3), ("local_w", C.c_void_p * 2),
                ("local_b", C.c_void_p * 2), ("layer", S3dLayerParams * N_LAYERS),
                ("fc_out_w", C.c_void_p), ("fc_out_b", C.c_void_p)]


class S3dGtLatent(C.Structure):
    _fields_ = [("proj", C.c_void_p * 4), ("fine", C.c_void_p), ("n_img", C.c_int), ("size", C.c_int)]


class S3dLatent(C.Structure):
    _fields_ = [("proj", C.c_void_p * 3), ("fine", C.c_void_p * 2), ("n_img", C.c_int),
                ("size", C.c_int)]


# name -> (restype, argtypes); every symbol include/slice3d_hip.h declares
_vp, _i, _l, _sz, _f = C.c_void_p, C.c_int, C.c_long, C.c_size_t, C.c_float
SYMBOLS = {
    "s3d_version": (_i, []),
    "s3d_last_error": (C.c_char_p, []),
    "s3d_unet_packed_bytes": (_sz, [_i]),
    "s3d_unet_pack": (_i, [C.POINTER(S3dUNetParams), _vp, _sz, _vp]),
    "s3d_unet_workspace_bytes": (_sz, [_i, _i, _i]),
    "s3d_unet_encode_fwd": (_i, [_vp, _vp, C.POINTER(S3dPyramid), _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "s3d_head_packed_bytes": (_sz, []),
    "s3d_head_pack": (_i, [C.POINTER(S3dHeadParams), _vp, _sz, _vp]),
    "s3d_latent_build": (_i, [_vp, C.POINTER(S3dPyramid), C.POINTER(S3dLatent), _i, _vp]),
    "s3d_decode_workspace_bytes": (_sz, [_i, _l, _i]),
    "s3d_decode_workspace_bytes_min": (_sz, [_i, _l, _i]),
    "s3d_decode_set_lanes": (_i, [_i]),
    "s3d_decode_set_last_fused": (_i, [_i]),
    "s3d_decode_set_shared_footprint": (_i, [_i]),
    "s3d_decode_points_fwd": (_i, [_vp, C.POINTER(S3dLatent), _vp, _vp, _vp, _i, _vp, _i, _l, _i, _i,
                                   _vp, _sz, _vp]),
    "s3d_decode_stages_floats": (_sz, [_i, _l, _i]),
    "s3d_decode_points_stages_fwd": (_i, [_vp, C.POINTER(S3dLatent), _vp, _vp, _vp, _i, _vp, _vp, _i, _l, _i, _i,
                                          _vp, _sz, _vp]),
    "s3d_decode_grid_fwd": (_i, [_vp, C.POINTER(S3dLatent), _vp, _i, _f, _vp, _i, _i, _vp, _sz, _vp]),
    "s3d_decode_grid_slab_fwd": (_i, [_vp, C.POINTER(S3dLatent), _vp, _i, _f, _l, _l, _vp, _i, _i, _vp, _sz, _vp]),
    "s3d_gt_encoder_packed_bytes": (_sz, []),
    "s3d_gt_encoder_pack": (_i, [C.POINTER(S3dVgg16BnParams), _vp, _sz, _vp]),
    "s3d_gt_encoder_workspace_bytes": (_sz, [_i, _i]),
    "s3d_gt_encode_fwd": (_i, [_vp, _vp, C.POINTER(S3dGtPyramid), _i, _i, _i, _vp, _sz, _vp]),
    "s3d_gt_head_packed_bytes": (_sz, []),
    "s3d_gt_head_pack": (_i, [C.POINTER(S3dGtHeadParams), _vp, _sz, _vp]),
    "s3d_gt_latent_build": (_i, [_vp, C.POINTER(S3dGtPyramid), C.POINTER(S3dGtLatent), _i, _vp]),
    "s3d_gt_decode_workspace_bytes": (_sz, [_i, _l, _i]),
    "s3d_gt_decode_points_fwd": (_i, [_vp, C.POINTER(S3dGtLatent), _vp, _vp, _vp, _i, _vp, _i, _l, _i, _i,
                                      _vp, _sz, _vp]),
    "s3d_gt_decode_grid_fwd": (_i, [_vp, C.POINTER(S3dGtLatent), _vp, _i, _f, _vp, _i, _i, _vp, _sz, _vp]),
    "s3d_conv_packed_bytes": (_sz, [_i, _i, _i, _i]),
    "s3d_conv_pack": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "s3d_conv_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "s3d_group_norm_stats_floats": (_sz, [_i, _i]),
    "s3d_group_norm_table_fwd": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, C.c_long, _vp, _vp, _i, _i, _i, _f, _vp, _vp]),
    "s3d_group_norm_partial_fwd": (_i, [_vp, _vp, _vp, _vp, C.c_long, _vp, _vp, _i, _i, _i, _f, _i, _vp]),
    "s3d_conv_gn_supported": (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "s3d_conv_gn_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _sz, _vp, _vp]),
    "s3d_conv_finish_fwd": (_i, [_vp, _i, _i, _i, _vp]),
    "s3d_group_norm_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "s3d_group_norm_film_fwd": (_i, [_vp, _vp, _vp, _vp, C.c_long, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "s3d_group_norm2_fwd": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp]),
    "s3d_qkv_attention_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "s3d_qkv_attention_ws_bytes": (_sz, [_i, _i, _i, _i]),
    "s3d_qkv_attention_ws_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "s3d_resample2x_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "s3d_small_linear_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "s3d_timestep_embedding_fwd": (_i, [_vp, _vp, _i, _i, _f, _vp]),
    "s3d_add_fwd": (_i, [_vp, _vp, _vp, _l, _vp]),
    "s3d_nchw_to_nhwc_pad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "s3d_add_nchw_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "s3d_vgg_packed_bytes": (_sz, []),
    "s3d_vgg_pack": (_i, [C.POINTER(S3dVggParams), _vp, _sz, _vp]),
    "s3d_vgg_workspace_bytes": (_sz, [_i, _i]),
    "s3d_vgg_loss_fwd": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "s3d_train_workspace_bytes": (_sz, [_i, _i, _l, _i]),
    "s3d_train_fwd_bwd": (_i, [C.POINTER(S3dUNetParams), C.POINTER(S3dHeadParams), C.POINTER(S3dVggParams),
                               C.POINTER(S3dUNetParams), C.POINTER(S3dHeadParams), C.POINTER(S3dTrainBatch),
                               _i, _i, _l, _i, _f, C.c_ulonglong, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "s3d_train_fwd": (_i, [C.POINTER(S3dUNetParams), C.POINTER(S3dHeadParams), C.POINTER(S3dVggParams),
                           C.POINTER(S3dTrainBatch), _i, _i, _l, _i, _f, C.c_ulonglong, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "s3d_train_bwd": (_i, [C.POINTER(S3dUNetParams), C.POINTER(S3dHeadParams), C.POINTER(S3dVggParams),
                           C.POINTER(S3dUNetParams), C.POINTER(S3dHeadParams), C.POINTER(S3dTrainBatch),
                           _i, _i, _l, _i, _f, C.c_ulonglong, _i, _vp, _vp, _f, _f, _vp, _vp, _sz, _vp]),
    "s3d_gt_train_workspace_bytes": (_sz, [_i, _i, _l, _i]),
    "s3d_gt_train_fwd_bwd": (_i, [C.POINTER(S3dVgg16BnParams), C.POINTER(S3dGtHeadParams),
                                  C.POINTER(S3dVgg16BnParams), C.POINTER(S3dGtHeadParams), C.POINTER(S3dTrainBatch),
                                  _i, _i, _l, _i, _f, C.c_ulonglong, _i, _vp, _vp, _vp, _sz, _vp]),
    "s3d_dropout_mask": (_i, [C.c_ulonglong, _i, C.c_ulonglong, _l, _f, _vp, _vp]),
    "s3d_adam_step": (_i, [_vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _i, _vp]),
    "s3d_adam_step_multi": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _f, _f, _f, _f, _i, _vp]),
    "s3d_prof_enable": (_i, [_i]),
    "s3d_range_push": (_i, [C.c_char_p]),
    "s3d_range_pop": (_i, []),
    "s3d_prof_read": (_i, [_i, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
    "s3d_project_coord_fwd": (_i, [_vp, _vp, _vp, _i, _l, _vp]),
    "s3d_query_sort_workspace_bytes": (_sz, [_i, _l]),
    "s3d_query_sort": (_i, [_vp, _vp, _vp, _i, _i, _l, _vp, _vp, _sz, _vp]),
    "s3d_sample_pyramid_workspace_bytes": (_sz, [_i, _l]),
    "s3d_sample_pyramid_fwd": (_i, [C.POINTER(S3dPyramid), _vp, _vp, _i, _i, _l, _vp, _sz, _vp]),
    "s3d_sample_planes_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _l, _vp]),
    "s3d_mise_dev_workspace_bytes": (_sz, [_i, _i]),
    "s3d_mise_dev_create": (_vp, [_vp, _sz, _i, _i, C.c_double, _vp]),
    "s3d_mise_dev_destroy": (None, [_vp]),
    "s3d_mise_dev_resolution": (_i, [_vp]),
    "s3d_mise_dev_query": (_i, [_vp, _vp, _l, C.POINTER(C.c_long), _vp]),
    "s3d_mise_dev_points": (_i, [_vp, _vp, _l, _f, _vp, _vp]),
    "s3d_mise_dev_update": (_i, [_vp, _vp, _vp, _l, _vp]),
    "s3d_mise_dev_update_f64": (_i, [_vp, _vp, _vp, _l, _vp]),
    "s3d_mise_dev_to_dense": (_i, [_vp, _vp, _vp]),
    "s3d_mc_dev_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "s3d_mc_dev_count": (_i, [_vp, _i, _i, _i, _i, _i, C.c_double, C.c_double, _vp, _sz, C.POINTER(C.c_long),
                              C.POINTER(C.c_long), _vp]),
    "s3d_mc_dev_emit": (_i, [_vp, _i, _i, _i, _i, _i, C.c_double, C.c_double, _vp, _sz, _vp, _vp, _vp]),
    "s3d_dataset_images_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "s3d_dataset_points_fwd": (_i, [_vp, _vp, _l, _vp, _vp, _vp, _vp]),
    "s3d_nchw_to_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "s3d_nhwc_to_nchw": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
}

_lib = None


class S3dError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises if the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise S3dError(
            "libslice3d_hip.so not found at %s — build it with `make -C slice3d_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.s3d_version() < 110:
        raise S3dError("libslice3d_hip.so too old: %d" % lib.s3d_version())
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().s3d_last_error()
        raise S3dError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))


def ptr(t):
    """Device pointer of a contiguous fp32 torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous() and t.dtype.is_floating_point and t.element_size() == 4, \
        "expected a contiguous fp32 tensor"
    return t.data_ptr()
