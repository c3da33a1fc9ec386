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
    if args.name_model == "slicenet":
        model = Slices3DRegModel(img_size=args.img_size, n_slices=args.n_slices, mode=args.mode)
    elif args.name_model == "gtslice":   # regression from given slices (model_gt.py; --from_which_slices gt|gen|gt_rec)
        from slice3d_amd.models_gt import Slices3DGTModel
        model = Slices3DGTModel(img_size=args.img_size, n_slices=args.n_slices, mode=args.mode)
    else:
        raise SystemExit("--name_model %s is not built (DISN: SURVEY.md section 2, out of scope)" % args.name_model)
    path_ckpt = os.path.join("experiments", args.name_exp, "ckpt", args.name_ckpt)
    if os.path.isfile(path_ckpt):
        model.load_state_dict(torch.load(path_ckpt, map_location="cpu")["model"])     # strict, as the reference
    elif args.synthetic_weights:   # smoke tests without a trained checkpoint; never silently
        print("checkpoint %s not found: --synthetic_weights -> name-seeded random weights" % path_ckpt)
        from slice3d_amd.weights import load_seeded
        load_seeded(model, 0)
    else:   # the reference fails here too (reconstruct.py:343 torch.load)
        raise FileNotFoundError("checkpoint %s not found (check --name_exp / --name_ckpt; --synthetic_weights runs on "
                                "random weights for smoke tests)" % path_ckpt)
    model = model.cuda().eval()
    path_res = os.path.join("experiments", args.name_exp, "results", args.name_dataset)
    os.makedirs(path_res, exist_ok=True)
    generator = Generator3D(model, threshold=args.mc_threshold, resolution0=args.mc_res0,
                            upsampling_steps=args.mc_up_steps, chunk_size=args.mc_chunk_size,
                            pred_type=args.pred_type)
    if args.name_dataset != "synthetic":
        from slice3d_amd.datasets import Slice3DDataset
        dataset = Slice3DDataset(split="test", args=args)
    else:
        dataset = SyntheticSlice3DDataset(args.synthetic_len, args.img_size, 16, args.n_slices, split="test")
    with torch.no_grad():
        for idx in range(len(dataset)):
            shape = dataset.files[idx][1] if hasattr(dataset, "files") else "synthetic_%04d" % idx
            path_mesh = os.path.join(path_res, shape + ".obj")
            if not args.overwrite_res and os.path.exists(path_mesh):
                continue
            data = {k: v.unsqueeze(0).cuda() for k, v in dataset[idx].items()}
            mesh, stats = generator.generate_mesh(data)
            mesh.export(path_mesh)
            print(path_mesh, "%d verts %d faces" % (len(mesh.vertices), len(mesh.faces)), stats)


if __name__ == "__main__":
    main()
