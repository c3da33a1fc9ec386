"""Developer script (GPU box): stage-by-stage comparison of the HIP path with the CPU oracle."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slice3d_amd.models import Slices3DRegModel
from slice3d_amd.weights import load_seeded
from slice3d_amd.synth import make_feed_dict
from oracle import ref_cpu

def run(B, S, Q, ns, mode, seed=5):
    torch.manual_seed(0)
    model = Slices3DRegModel(n_slices=ns, mode=mode)
    load_seeded(model, 0)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.cuda().eval()
    fd = make_feed_dict(B, S, Q, ns, seed=seed)
    fdg = {k: v.cuda() for k, v in fd.items() if k != 'img_slices'}
    t0 = time.time()
    code = model.encode(fdg, want_slices=True)
    torch.cuda.synchronize(); t1 = time.time()
    sdf = model.decode_sdf(fdg['qry_norot'], code, obj_rot_mat=fdg['obj_rot_mat'], trans_mat_wo_rot_tp=fdg['trans_mat_wo_rot_tp'])
    torch.cuda.synchronize(); t2 = time.time()
    print(f'[B={B} S={S} Q={Q} ns={ns} {mode}] encode {t1-t0:.3f}s decode {t2-t1:.3f}s')
    with torch.no_grad():
        feats, rec = ref_cpu.unet_forward(sd, fd['img_input'], ns)
        for l, f in enumerate(feats):
            mine = code.pyramid[l].permute(0, 3, 1, 2).cpu()
            print(f'  pyramid[{l}] max|diff| {float((mine-f).abs().max()):.3e}  (|ref|max {float(f.abs().max()):.2f})')
        print(f'  slices_rec max|diff| {float((code.slices_rec_flat.cpu()-rec).abs().max()):.3e}')
        qr = ref_cpu.rotate_queries(fd, mode)
        ref_sdf = ref_cpu.decode_points(sd, feats, qr, fd['trans_mat_wo_rot_tp'], ns)
        d = (sdf.cpu() - ref_sdf).abs()
        print(f'  sdf max|diff| {float(d.max()):.3e} mean {float(d.mean()):.3e} (|ref|max {float(ref_sdf.abs().max()):.2f})')
        # decode with the ORACLE pyramid uploaded -> isolates the decoder
    return float(d.max())

if __name__ == '__main__':
    print(torch.cuda.get_device_name(0))
    run(1, 32, 200, 12, 'test')
    run(1, 64, 1000, 4, 'train')
    run(2, 32, 300, 12, 'train')
    run(1, 64, 2048, 12, 'test')
    if len(sys.argv) > 1:
        run(1, 256, 4096, 12, 'test')
