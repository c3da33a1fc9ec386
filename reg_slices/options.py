"""Command-line surface of the reg_slices scripts — same flag names, types, defaults and choices as the
reference's single shared parser (reference: reg_slices/options.py:3-53), declared as a table.
Additions for this build are listed after the reference flags."""
import argparse

_BOOL = argparse.BooleanOptionalAction

# (flag, kwargs) — grouped as in the reference
_REFERENCE_FLAGS = [
    ("name_model", dict(type=str, default="slicenet", choices=["slicenet", "disn", "gtslice"])),
    # dataset
    ("dir_data", dict(type=str, default="../data")),
    ("name_dataset", dict(type=str, default="shapenet",
                          choices=["objaverse", "shapenet", "custom", "custom_sin_img", "synthetic"])),
    ("name_single", dict(type=str, default="fertility", help="name of the single shape")),
    ("n_wk", dict(type=int, default=16, help="number of workers in dataloader")),
    ("categories_train", dict(type=str, default="objaverse,", help="training / validation categories (ShapeNet)")),
    ("categories_test", dict(type=str, default="objaverse,", help="testing categories (ShapeNet)")),
    ("add_noise", dict(type=float, default=0, help="std of noise added to the point clouds")),
    ("gt_source", dict(type=str, default="imnet", choices=["imnet", "occnet"], help="query-occ ground truth source")),
    ("img_size", dict(type=int, default=128, help="img_size")),
    ("n_qry", dict(type=int, default=256, help="number of query points per shape when training")),
    ("n_slices", dict(type=int, default=12, help="number of slices for each shape")),
    ("n_views", dict(type=int, default=12, help="number of views for each shape")),
    ("pred_type", dict(type=str, default="sdf", choices=["occ", "sdf"], help="occupancy or signed distance")),
    # common hyper-parameters
    ("name_exp", dict(type=str, default="2023_07_04_chairs_vggptloss")),
    ("name_exp_cam", dict(type=str, default="2023_1107_airplanes_est")),
    ("mode", dict(type=str, default="train", choices=["train", "val", "test"])),
    ("n_bs", dict(type=int, default=16, help="batch size")),
    ("n_epochs", dict(type=int, default=600, help="number of epochs")),
    ("lr", dict(type=float, default=3e-4, help="init learning rate")),
    ("n_dim", dict(type=int, default=128, help="dimension of hidden layer features")),
    ("multi_gpu", dict(type=bool, default=False)),
    ("freq_ckpt", dict(type=int, default=4, help="checkpoint every freq_ckpt epochs")),
    ("freq_log", dict(type=int, default=200, help="log every freq_log iterations")),
    ("freq_decay", dict(type=int, default=100, help="decay the lr every freq_decay epochs")),
    ("weight_decay", dict(type=float, default=0.5, help="lr decay factor (the reference's name)")),
    ("tboard", dict(type=bool, default=True, help="use tensorboard if it is installed")),
    ("resume", dict(action=_BOOL, help="resume training")),
    ("est_campose", dict(action=_BOOL, help="use estimated camera poses")),
    ("back_bone_cam_est", dict(type=str, default="vgg16_bn", choices=["vgg16_bn", "resnet50"])),
    ("use_white_bg", dict(action=_BOOL, help="composite RGBA inputs on white")),
    # marching cubes
    ("mc_chunk_size", dict(type=int, default=3000, help="query points per chunk during mesh extraction")),
    ("mc_res0", dict(type=int, default=64, help="start resolution for MISE")),
    ("mc_up_steps", dict(type=int, default=2, help="number of upsampling steps")),
    ("mc_threshold", dict(type=float, default=0.5, help="threshold for network output values")),
    # testing
    ("name_ckpt", dict(type=str, default="10_5511_0.0876_0.9612.ckpt")),
    ("name_ckpt_cam", dict(type=str, default="570_225545_1.969e-05.ckpt")),
    ("from_which_slices", dict(type=str, default="gt", choices=["gt", "gt_rec", "gen"], help="which slices to use")),
    ("overwrite_res", dict(action=_BOOL, help="overwrite existing results")),
]

_BUILD_FLAGS = [
    ("synthetic_len", dict(type=int, default=64, help="[build] samples per epoch of --name_dataset synthetic")),
    ("dropout", dict(type=float, default=0.1, help="[build] transformer dropout in training (the reference trains with "
                                                   "nn.TransformerEncoderLayer's default 0.1, models.py:18)")),
    ("seed", dict(type=int, default=0, help="[build] base seed of the dropout streams (mixed with the rank)")),
    ("synthetic_weights", dict(action=_BOOL, help="[build] reconstruct*.py: run on name-seeded random weights when "
                                                  "no checkpoint is given (smoke tests only; meshes are meaningless)")),
    ("shards", dict(type=str, default="", help="[build] directory of pre-packed uint8 shards (reg_slices/pack_shards.py): "
                                               "batches are staged on the GPU instead of decoding 13 PNGs per sample")),
    ("shards_in_hbm", dict(action=_BOOL, help="[build] keep the packed split resident in GPU memory")),
    ("sync_bn", dict(action=_BOOL, help="[build] data-parallel training: BatchNorm statistics over all ranks")),
    ("prec", dict(type=str, default="f32", choices=["f32", "f16x3", "f16"],
                  help="[build] train.py: arithmetic of the step's GEMMs — f32 (exact fp32 MFMAs), f16x3 (split precision, fp32-class, "
                       "~1.5x faster) or f16 (single-pass throughput mode of the decoder, a further 1.3x, not fp32-class)")),
]


def get_parser():
    parser = argparse.ArgumentParser()
    for name, kw in _REFERENCE_FLAGS + _BUILD_FLAGS:
        parser.add_argument("--" + name, **kw)
    return parser
