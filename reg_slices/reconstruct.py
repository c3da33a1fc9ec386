"""reconstruct.py — host entry point kept from the reference (reg_slices/reconstruct.py:334-416):
load a checkpoint into Slices3DRegModel(mode='test'), run Generator3D (MISE or dense grid -> eval_points
-> marching cubes) per test object and export `<shape>.obj`, with the reference's flags.

    python reg_slices/reconstruct.py --name_exp demo --name_ckpt x.ckpt --name_dataset synthetic \
        --mode test --img_size 128 --mc_res0 64 --mc_up_steps 2
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from options import get_parser  # noqa: E402
from slice3d_amd.generator import Generator3D  # noqa: E402
from slice3d_amd.models import Slices3DRegModel  # noqa: E402
from slice3d_amd.synth import SyntheticSlice3DDataset  # noqa: E402


def main():
    args = get_parser().parse_args()
    if args.name_model != "slicenet":
        raise SystemExit("only --name_model slicenet is built (DISN / GT-slices models: SURVEY.md section 2)")
    model = Slices3DRegModel(img_size=args.img_size, n_slices=args.n_slices, mode=args.mode)
    path_ckpt = os.path.join("experiments", args.name_exp, "ckpt", args.name_ckpt)
    if os.path.isfile(path_ckpt):
        model.load_state_dict(torch.load(path_ckpt, map_location="cpu")["model"])     # strict, as the reference
    else:
        print("checkpoint %s not found: using name-seeded synthetic weights" % path_ckpt)
        from slice3d_amd.weights import load_seeded
        load_seeded(model, 0)
    model = model.cuda().eval()
    path_res = os.path.join("experiments", args.name_exp, "results", args.name_dataset)
    os.makedirs(path_res, exist_ok=True)
    generator = Generator3D(model, threshold=args.mc_threshold, resolution0=args.mc_res0,
                            upsampling_steps=args.mc_up_steps, chunk_size=args.mc_chunk_size,
                            pred_type=args.pred_type)
    if args.name_dataset != "synthetic":
        raise SystemExit("on-disk datasets (SURVEY.md 8(f-3)) are not built yet; use --name_dataset synthetic")
    dataset = SyntheticSlice3DDataset(args.synthetic_len, args.img_size, 16, args.n_slices, split="test")
    with torch.no_grad():
        for idx in range(len(dataset)):
            path_mesh = os.path.join(path_res, "synthetic_%04d.obj" % idx)
            if not args.overwrite_res and os.path.exists(path_mesh):
                continue
            data = {k: v.unsqueeze(0).cuda() for k, v in dataset[idx].items()}
            mesh, stats = generator.generate_mesh(data)
            mesh.export(path_mesh)
            print(path_mesh, "%d verts %d faces" % (len(mesh.vertices), len(mesh.faces)), stats)


if __name__ == "__main__":
    main()
