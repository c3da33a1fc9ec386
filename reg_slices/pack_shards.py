"""pack_shards.py — one-off conversion of a dataset in the reference's on-disk layout (README.md:37-47) into the
uint8 shards slice3d_amd.shards.ShardLoader streams (train.py / train_gt.py --shards DIR).  Takes the options of
reg_slices/options.py that decide the image contents (--dir_data --name_dataset --img_size --n_views
--from_which_slices --use_white_bg) plus --shards for the output directory.

    python reg_slices/pack_shards.py --dir_data data --name_dataset objaverse --img_size 256 --n_views 12 --shards packed/
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from options import get_parser  # noqa: E402
from slice3d_amd.shards import pack_dataset  # noqa: E402

if __name__ == "__main__":
    args = get_parser().parse_args()
    if not args.shards:
        raise SystemExit("--shards OUT_DIR is required")
    print("packed:", pack_dataset(args, args.shards))
