"""reconstruct_slices.py — host entry point kept from the reference (reg_slices/reconstruct_slices.py:54-138):
run the slice-generating model on every test object and write its 12 regressed slice images as
experiments/<name_exp>/img_slices/<shape>/{X,Z,Y}_{1..4}.png (256x256, the reference's naming: X_1..4, Z_4..1,
Y_1..4 for slice indices 0..11).

    python reg_slices/reconstruct_slices.py --name_exp demo --name_ckpt x.ckpt --name_dataset synthetic --mode test

Only the forward of Slices3DRegModel is needed (the U-Net slice generator: s3d_unet_encode_fwd); camera-pose
estimation (--est_campose, CameraNet) is out of scope.
"""
import os
import sys

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from options import get_parser  # noqa: E402
from slice3d_amd.models import Slices3DRegModel  # noqa: E402
from slice3d_amd.synth import SyntheticSlice3DDataset  # noqa: E402


def slice_name(idx):
    """reconstruct_slices.py:33-52: slices 0-3 -> X_1..4, 4-7 -> Z_4..1, 8-11 -> Y_1..4."""
    if idx < 4:
        return "X_%d" % (idx + 1)
    if idx < 8:
        return "Z_%d" % (8 - idx)
    return "Y_%d" % (idx - 7)


def save_slices(slices_rec, dir_tgt):
    """slices_rec (3*n_slices, S, S) in [-1, 1] -> 256x256 PNGs (denorm, bilinear resize as cv2.resize's default)."""
    os.makedirs(dir_tgt, exist_ok=True)
    n = slices_rec.shape[0] // 3
    for i in range(n):
        img = (slices_rec[3 * i:3 * i + 3] * 0.5 + 0.5).clamp(0, 1).permute(1, 2, 0).cpu().numpy()
        im = Image.fromarray((img * 255.0).astype(np.uint8)).resize((256, 256), Image.BILINEAR)
        im.save(os.path.join(dir_tgt, slice_name(i) + ".png"))


def main():
    args = get_parser().parse_args()
    if args.name_model != "slicenet":
        raise SystemExit("reconstruct_slices needs the slice-generating model (--name_model slicenet)")
    if getattr(args, "est_campose", False):
        raise SystemExit("--est_campose (CameraNet) is out of scope (SURVEY.md section 2)")
    model = Slices3DRegModel(img_size=args.img_size, n_slices=args.n_slices, mode=args.mode)
    path_ckpt = os.path.join("experiments", args.name_exp, "ckpt", args.name_ckpt)
    if os.path.isfile(path_ckpt):
        model.load_state_dict(torch.load(path_ckpt, map_location="cpu")["model"])
    elif args.synthetic_weights:   # smoke tests without a trained checkpoint; never silently
        print("checkpoint %s not found: --synthetic_weights -> name-seeded random weights" % path_ckpt)
        from slice3d_amd.weights import load_seeded
        load_seeded(model, 0)
    else:   # the reference fails here too (reconstruct.py:343 torch.load)
        raise FileNotFoundError("checkpoint %s not found (check --name_exp / --name_ckpt; --synthetic_weights runs on "
                                "random weights for smoke tests)" % path_ckpt)
    model = model.cuda().eval()
    if args.name_dataset != "synthetic":
        from slice3d_amd.datasets import Slice3DDataset
        dataset = Slice3DDataset(split="test", args=args)
    else:
        dataset = SyntheticSlice3DDataset(args.synthetic_len, args.img_size, 16, args.n_slices, split="test")
    dir_output = os.path.join("experiments", args.name_exp, "img_slices")
    with torch.no_grad():
        for idx in range(len(dataset)):
            shape = dataset.files[idx][1] if hasattr(dataset, "files") else "synthetic_%04d" % idx
            data = {k: v.unsqueeze(0).cuda() for k, v in dataset[idx].items()}
            code = model.encode(data, want_slices=True, build_latent=False)
            rec = code.slices_rec_flat.view(3 * args.n_slices, args.img_size, args.img_size)
            save_slices(rec, os.path.join(dir_output, shape))
            print(os.path.join(dir_output, shape))


if __name__ == "__main__":
    main()
