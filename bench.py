#!/usr/bin/env python
"""bench.py — occupancy query-points/s of the Slice3D regression hot path on MI355X.

Workload (BASELINE.json configs[1]): one object per step = reg_slices U-Net encode of a 256x256 image
into the 12-slice feature pyramid + decode of 100 000 query points (project -> sample 12x5 planes ->
fc_p/fc_s -> 3-layer transformer -> fc_out), inputs resident in HBM.  --prec f16x3 (default): fp32 operands
split into f16 hi+lo and multiplied with 3 f16 MFMAs per product (fp32-class, passes the 1e-4 gate);
--prec f32: exact fp32 MFMA.
N GPUs = N independent objects (one process per GPU, no data-path collective): weak scaling.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

UNET_GFLOP_256 = 241.97   # encoder 40.09 + 12-slice decoder 201.88 GFLOP at S=256
F32_MFMA_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
F16_MFMA_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak (spec); 16x16x32 f16 measures 1955
FFN_FLOP_PER_ROW = 2 * 2 * 128 * 2048   # two 128x2048 GEMMs, 2 FLOP/MAC (FFN = 88 %)
F_MIN_PER_QUERY = 35.96e6           # exact decoder FLOPs/query with last-layer pruning
ATTN_FLOP_PER_TOKEN = 2 * (3 * 128 * 128 + 128 * 128) + 2 * 2 * 13 * 128   # in_proj + out_proj + (QK^T, PV) over 13 keys


def _r(x, nd=4):
    """Round every float of a JSON-able value to `nd` significant decimals (the driver keeps ~2 KB of the line's tail)."""
    if isinstance(x, float):
        return float("%.*g" % (nd + 2, x))
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


class _ClockSampler:
    """Samples every GPU's shader clock (MHz) and socket power (W) from sysfs every 50 ms on a host thread while a leg runs and
    reports the card that drew the most power — the one under load: the sysfs card order is not HIP's device order
    (this part's clock depends on the data the matrix pipes see: profiles/r02_ffn_data_power.md)."""

    def __init__(self):
        import glob
        self.cards = []
        for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
            hw = sorted(glob.glob(os.path.dirname(f) + "/hwmon/hwmon*/power1_*"))
            fp = next((h for h in hw if h.endswith("power1_input")), None) or next((h for h in hw if h.endswith("power1_average")), None)
            self.cards.append({"sclk_file": f, "power_file": fp, "sclk": [], "power": []})
        self._stop = False
        self._th = None

    def _read(self):
        for c in self.cards:
            try:
                for line in open(c["sclk_file"]):
                    if "*" in line:
                        c["sclk"].append(float(line.split(":")[1].lower().split("mhz")[0]))
                if c["power_file"]:
                    c["power"].append(float(open(c["power_file"]).read()) * 1e-6)
            except (OSError, ValueError, IndexError):
                pass

    def __enter__(self):
        import threading
        if self.cards:
            def loop():
                while not self._stop:
                    self._read()
                    time.sleep(0.05)
            self._th = threading.Thread(target=loop, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        if self._th:
            self._th.join()

    def result(self):
        drop = lambda v: v[len(v) // 4:] if len(v) >= 8 else v      # the first quarter is the ramp
        m = lambda v: sum(v) / len(v) if v else None
        best = max(self.cards, key=lambda c: m(drop(c["power"])) or 0.0, default=None)
        if best is None:
            return None, None
        return m(drop(best["sclk"])), m(drop(best["power"]))


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _median_time(fn, runs):
    """One warm-up call, then the median wall time of `runs` calls."""
    fn()
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def cpu_baseline(sd, fd, n_slices, n_sample, gpu_sdf, runs=5):
    """Oracle (CPU restatement of the reference path, oracle/ref_cpu.py) timed on the host cores for a bounded
    sample, the way BASELINE.md section 4 prescribes: same synthetic inputs, stage split (U-Net / sample / decoder
    tokens), one warm-up + median of `runs`, at the best thread count of a short probe that includes ALL host cores
    (ATen's CPU kernels get slower beyond ~16-32 threads on many-core hosts; the probe records by how much)."""
    from oracle import ref_cpu
    fd_cpu = {k: v.cpu() for k, v in fd.items()}
    trans = fd_cpu["trans_mat_wo_rot_tp"]
    n_qry = fd_cpu["qry_norot"].shape[1]
    qry = ref_cpu.rotate_queries(fd_cpu, "test")[:, :n_sample]
    host = os.cpu_count() or 1
    chunk = 4096
    with torch.no_grad():
        # thread-count probe on a small problem (64^2 U-Net pyramid, 256 queries; one warm-up + one timed call);
        # stops climbing once a count is 3x slower than the best so far
        small_img = fd_cpu["img_input"][:, :, :64, :64]
        probe = {}
        for n in sorted({min(c, host) for c in (8, 16, 32, 64, host)}):
            torch.set_num_threads(n)
            feats_small, _ = ref_cpu.unet_forward(sd, small_img, n_slices)
            probe[n] = _median_time(lambda: ref_cpu.decode_points(sd, feats_small, qry[:, :256], trans, n_slices), 1)
            if probe[n] > 3 * min(probe.values()) and n < host:
                torch.set_num_threads(host)      # the all-core point is always recorded
                probe[host] = _median_time(lambda: ref_cpu.decode_points(sd, feats_small, qry[:, :256], trans, n_slices), 1)
                break
        best_n = min(probe, key=probe.get)
        torch.set_num_threads(best_n)
        box = {}
        t_unet = _median_time(lambda: box.__setitem__("f", ref_cpu.unet_forward(sd, fd_cpu["img_input"], n_slices)[0]), runs)
        feats = box["f"]

        def sample_all():
            box["t"] = [ref_cpu.sample_pyramid(feats, ref_cpu.project_coord(qry[:, s:s + chunk], trans), n_slices)
                        for s in range(0, n_sample, chunk)]

        def tokens_all():
            box["s"] = torch.cat([ref_cpu.decode_tokens(sd, t, qry[:, s:s + chunk])
                                  for t, s in zip(box["t"], range(0, n_sample, chunk))], 1)
        t_sample = _median_time(sample_all, runs)
        t_tokens = _median_time(tokens_all, runs)
        sdf = box["s"]
    per_q = (t_sample + t_tokens) / n_sample
    err = float((gpu_sdf[:, :n_sample].cpu() - sdf).abs().max())
    # (how the figure is formed: DESIGN.md section 5 "bench line glossary")
    return {
        "value": n_qry / (t_unet + per_q * n_qry), "unit": "query-points/s", "cores": best_n, "kind": "port",
        "cpu_model": _cpu_model(),
        "sample": "oracle/ref_cpu.py: U-Net at %d^2 + %d of %d queries/object; median of %d; best of a thread probe"
                  % (fd_cpu["img_input"].shape[-1], n_sample, n_qry, runs),
        "stages": {"unet_s": t_unet, "sample_us_per_query": t_sample / n_sample * 1e6,
                   "decoder_tokens_us_per_query": t_tokens / n_sample * 1e6},
        "thread_probe_s": {k: float("%.3g" % v) for k, v in probe.items()},   # seconds per 256 queries at each thread count tried
    }, err


def _pmc_traffic(args, kname):
    """(2 * FETCH_SIZE + WRITE_SIZE) KiB -> bytes per launch of the kernel whose name contains `kname`, from two
    rocprofv3 --pmc child runs of this script's timed inference loop (counters need their own passes; gfx950 reports
    half of the bytes of a wide streaming read, hence the factor — MI355X_MICROARCH.md, HBM section)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, None
    vals = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="s3d_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
               os.path.abspath(__file__), "--pmc-child", "--steps", "2", "--warmup", "1", "--img-size", str(args.img_size),
               "--n-qry", str(args.n_qry), "--n-slices", str(args.n_slices), "--batch", str(args.batch), "--prec", args.prec]
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
            per_inst = {}   # every instantiation of the kernel template (mangled or demangled name) -> its launches' values
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == counter and kname in row["Kernel_Name"]:
                        per_inst.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
            if not per_inst:
                print("bench: no '%s' rows in the rocprofv3 %s pass" % (kname, counter), file=sys.stderr)
                return None, None
            acc = max(per_inst.values(), key=sum)    # the dominant instantiation (the non-final layers' kernel)
            vals[counter] = sum(acc) / len(acc)
        except Exception as e:
            print("bench: rocprofv3 --pmc %s pass failed (%s)" % (counter, e), file=sys.stderr)
            return None, None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, (
        "in-run rocprofv3 --pmc: (2*FETCH %.0f + WRITE %.0f) KiB" % (vals["FETCH_SIZE"], vals["WRITE_SIZE"]))


def _self_launch(n):
    """Re-run this command as an n-rank torch.distributed.run job on this node (127.0.0.1, a free port)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--img-size", type=int, default=256)
    ap.add_argument("--n-qry", type=int, default=100000)
    ap.add_argument("--n-slices", type=int, default=12)
    ap.add_argument("--batch", type=int, default=4, help="objects per GPU per step (BASELINE C2: B = 1..4)")
    ap.add_argument("--cpu-sample", type=int, default=8192, help="queries timed on the CPU baseline (0 = skip)")
    ap.add_argument("--cpu-runs", type=int, default=5, help="timed runs per stage of the CPU baseline (median; BASELINE.md section 4)")
    ap.add_argument("--prec", default="f16x3", choices=["f32", "f16x3"], help="arithmetic mode of the decoder GEMMs")
    ap.add_argument("--f16-steps", type=int, default=5, help="timed steps of the single-pass f16 throughput mode, reported "
                                                              "separately with its error (0 = skip)")
    ap.add_argument("--f32-steps", type=int, default=3, help="timed steps of the exact fp32-MFMA mode, reported beside the headline "
                                                              "(0 = skip)")
    ap.add_argument("--noise-steps", type=int, default=30, help="timed steps on white-noise images with the clock sampled "
                                                                "(SURVEY 8(d)'s inputs; 0 = skip)")
    ap.add_argument("--c4-steps", type=int, default=2, help="timed dense 256^3 grid evaluations (BASELINE configs[3]; 0 = skip)")
    ap.add_argument("--c4-res", type=int, default=256)
    ap.add_argument("--mesh-steps", type=int, default=2, help="timed reconstruct.py-default mesh extractions (MISE 64 -> 256 + "
                    "marching cubes on the device; 0 = skip)")
    ap.add_argument("--ldm-steps", type=int, default=20, help="timed LDM denoising steps (BASELINE configs[4]; 0 = skip)")
    ap.add_argument("--pmc", type=int, default=1, help="1: measure roofline.traffic in this run (two rocprofv3 --pmc child passes "
                    "of the timed inference loop, FETCH_SIZE and WRITE_SIZE); 0: quote the committed profiles/pmc_traffic.json")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # the child pass: inference loop only
    ap.add_argument("--train-steps", type=int, default=10, help="timed training steps for train_samples_per_s (0 = skip)")
    ap.add_argument("--gt-train-steps", type=int, default=5, help="timed Slices3DGTModel training steps (0 = skip)")
    ap.add_argument("--train-f16-steps", type=int, default=8,
                    help="timed steps of the training step's single-pass f16 throughput mode (prec='f16'; reported beside "
                         "train_ms_per_step with its gradient deviation, never instead of it; 0 = skip)")
    ap.add_argument("--infer-only", action="store_true", help="the headline loop alone (A/B scripts): every secondary leg off")
    args = ap.parse_args()
    if args.infer_only:
        for k in ("cpu_sample", "f16_steps", "f32_steps", "noise_steps", "c4_steps", "mesh_steps", "ldm_steps", "train_steps",
                  "gt_train_steps", "train_f16_steps", "pmc"):
            setattr(args, k, 0)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: spawn the N ranks ourselves (one process per GPU under torch.distributed.run on
        # a free local port); rank 0 of the child job prints the one JSON line, this process only forwards the exit code
        sys.exit(_self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run, or without it: bench.py spawns "
                         "its own ranks)" % (args.gpus, world))
    # S3D_BENCH_BACKEND=gloo: smoke-test the N-rank code path with several processes on ONE GPU (no RCCL between ranks
    # that share a device); never set by the driver — a real run is one rank per GPU over RCCL
    backend = os.environ.get("S3D_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:     # under a launcher even ONE rank forms its process group (RCCL communicator,
        import torch.distributed as dist      # barriers and the MAX-over-ranks reduction run as they do at N > 1)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from slice3d_amd import _lib
    from slice3d_amd.models import Slices3DRegModel
    from slice3d_amd.synth import make_feed_dict
    from slice3d_amd.weights import load_seeded

    model = Slices3DRegModel(img_size=args.img_size, n_slices=args.n_slices, mode="test", prec=args.prec)
    load_seeded(model, 0)
    sd_cpu = {k: v.clone() for k, v in model.state_dict().items()} if rank == 0 else None
    model.cuda().eval()
    fd = make_feed_dict(args.batch, args.img_size, args.n_qry, args.n_slices, seed=1234 + rank, with_slices=False,
                        device="cuda")
    lib = _lib.load()

    def step():
        code = model.encode(fd)                       # U-Net + latent maps, once per object
        return model.decode_sdf(fd["qry_norot"], code)  # 100k queries

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    lib.s3d_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    if args.pmc_child:      # counter pass under rocprofv3: only the timed loop's launches are wanted
        return
    rank_ms = [dt / args.steps * 1e3]
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)                       # per-rank times: a straggler shows in the first curve
        rank_ms = [float(v.item()) / args.steps * 1e3 for v in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stage_ms, counts = {}, {}
    for i, name in enumerate(_lib.PROF_NAMES):
        ms, n = C.c_double(), C.c_long()
        lib.s3d_prof_read(i, C.byref(ms), C.byref(n))
        stage_ms[name] = ms.value / args.steps
        counts[name] = n.value
    lib.s3d_prof_enable(0)

    # ---- secondary rooflines (north_star): stand-alone feature-sample op (HBM bound) ----
    sample_roof = None
    if rank == 0:
        fd1 = {k: v[:1].contiguous() for k, v in fd.items()}      # one object: the (12, Q, 992) tensor is 4.8 GB
        code = model.encode(fd1, build_latent=False)
        g = model.project_coord(fd1["qry_norot"] * torch.tensor([1.0, -1.0, -1.0], device="cuda"),   # mode='test' flip
                                fd1["trans_mat_wo_rot_tp"])
        feats = model.sample_pyramid(code.pyramid, g)   # the result tensor is allocated ONCE, outside the timed loop (round 5's
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # 25 ms was hipMalloc of 4.76 GB per call)
        torch.cuda.synchronize()
        lib.s3d_prof_enable(1)
        e0.record()
        for _ in range(5):
            model.sample_pyramid(code.pyramid, g, out=feats)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        kms, kn = C.c_double(), C.c_long()
        lib.s3d_prof_read(_lib.PROF_SAMPLE_PYR, C.byref(kms), C.byref(kn))
        lib.s3d_prof_enable(0)
        k_ms = kms.value / max(kn.value, 1)
        ch = sum(p.shape[-1] for p in code.pyramid)
        alg = args.n_slices * args.n_qry * (ch * 4 + 8) + sum(p.numel() * 4 for p in code.pyramid)
        sample_roof = {"kernel": "sample_pyramid_kernel (sample_from_planes x5 + cat)",
                       "bound": "hbm", "achieved": alg / (k_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                       "frac": alg / (k_ms * 1e-3) / 1e9 / 8000.0, "kernel_ms": k_ms,
                       "op_ms_incl_sort": ms,
                       }
        del code, feats

    # ---- throughput mode (NOT the headline): single-pass f16 MFMA, what BASELINE configs[1]'s "bf16" names; fails the
    #      1e-4 gate by construction, so it is reported beside the headline with its measured error ----
    def time_mode(prec, steps, feed):
        """ms per step of the headline workload in another arithmetic mode / on another feed (2 warm-up steps)."""
        m = Slices3DRegModel(img_size=args.img_size, n_slices=args.n_slices, mode="test", prec=prec)
        load_seeded(m, 0)
        m.cuda().eval()
        for _ in range(2):
            o = m.decode_sdf(feed["qry_norot"], m.encode(feed))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            o = m.decode_sdf(feed["qry_norot"], m.encode(feed))
        torch.cuda.synchronize()
        return (time.perf_counter() - t1) / steps * 1e3, o

    qps = lambda ms: args.n_qry * args.batch / (ms * 1e-3)
    f16_mode = bf16_mode = f32_mode = noise_leg = None
    if args.f16_steps > 0 and rank == 0:
        ms16, o16 = time_mode("f16", args.f16_steps, fd)
        f16_mode = {"ms_per_step": ms16, "query_points_per_s": qps(ms16),
                    "max_abs_diff_vs_headline_mode": float((o16 - out).abs().max())}
        del o16
        # the precision BASELINE configs[1] literally names: attention / FFN GEMMs on the bf16 MFMA (S3D_PREC_BF16)
        msb, ob = time_mode("bf16", args.f16_steps, fd)
        bf16_mode = {"ms_per_step": msb, "query_points_per_s": qps(msb),
                     "max_abs_diff_vs_headline_mode": float((ob - out).abs().max())}
        del ob
    # ---- the cost of the f16x3 choice: the same workload with exact fp32 MFMAs (v_mfma_f32_16x16x4_f32) ----
    if args.f32_steps > 0 and rank == 0 and args.prec != "f32":
        ms32, o32 = time_mode("f32", args.f32_steps, fd)
        f32_mode = {"ms_per_step": ms32, "query_points_per_s": qps(ms32),
                    "max_abs_diff_vs_headline_mode": float((o32 - out).abs().max())}
        del o32
    # ---- SURVEY 8(d)'s white-noise images (uniform(-1,1)): this part's clock depends on the operands' bit activity, so
    #      the same step is timed on both feeds with the shader clock / socket power sampled from sysfs ----
    if args.noise_steps > 0 and rank == 0:
        fdn = make_feed_dict(args.batch, args.img_size, args.n_qry, args.n_slices, seed=1234, smooth=False,
                             with_slices=False, device="cuda")
        noise_leg = {}
        for tag, feed in (("smooth", fd), ("noise", fdn)):
            for _ in range(2):
                step_on = model.decode_sdf(feed["qry_norot"], model.encode(feed))
            torch.cuda.synchronize()
            with _ClockSampler() as cs:
                t1 = time.perf_counter()
                for _ in range(args.noise_steps):
                    step_on = model.decode_sdf(feed["qry_norot"], model.encode(feed))
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t1) / args.noise_steps * 1e3
            mhz, watts = cs.result()
            noise_leg[tag] = {"ms_per_step": ms, "sclk_mhz": mhz, "power_w": watts}
        noise_leg["query_points_per_s"] = qps(noise_leg["noise"]["ms_per_step"])
        noise_leg["steps"] = args.noise_steps
        del fdn, step_on

    # ---- BASELINE configs[3]: reconstruct.py --mc_res0 256 --mc_up_steps 0 — the dense 256^3 logit grid of ONE object
    #      (16.7 M queries, coordinates generated in-kernel), copied to the host as Generator3D does.  With N ranks the
    #      grid's linear index is split into N contiguous slabs (every rank encodes the object itself) and the logits
    #      are all-gathered over RCCL: strong scaling of one object.
    c4 = None
    if args.c4_steps > 0:
        from slice3d_amd.generator import Generator3D
        fd1 = {k: v[:1].contiguous() for k, v in make_feed_dict(1, args.img_size, 16, args.n_slices, seed=1234,
                                                                 with_slices=False, device="cuda").items()}
        gen = Generator3D(model, resolution0=args.c4_res, upsampling_steps=0, pred_type="sdf")
        host = torch.empty((args.c4_res,) * 3, dtype=torch.float32).pin_memory()
        n_grid = args.c4_res ** 3

        def c4_once():
            code = gen.encode(fd1)
            grid = gen.decode_dense_grid(code, args.c4_res, 1.0, fd1["trans_mat_wo_rot_tp"])
            return grid

        c4_once()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.c4_steps):
            grid = c4_once()
        barrier()
        t_dev = (time.perf_counter() - t1) / args.c4_steps
        t1 = time.perf_counter()
        for _ in range(args.c4_steps):
            grid = c4_once()
            host.copy_(grid, non_blocking=True)
        barrier()
        t_host = (time.perf_counter() - t1) / args.c4_steps
        if dist is not None:
            t = torch.tensor([t_dev, t_host], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            t_dev, t_host = (float(v) for v in t.tolist())
        # BASELINE configs[3]; N ranks: one slab of the grid's linear index per rank + all_gather of the logits (strong scaling)
        c4 = {"res": args.c4_res, "seconds_device": t_dev, "seconds_incl_d2h": t_host, "query_points_per_s": n_grid / t_dev,
              "query_points_per_s_incl_d2h": n_grid / t_host, "n_gpus": world, "scaling": "strong"}
        if world > 1:   # the exchange step of the split alone: the all_gather of the slabs' logits (what the N-rank time pays beside 1/N of the decode)
            from slice3d_amd.parallel import gather_slabs, shard_range
            lo, hi = shard_range(n_grid, rank, world)
            slab = torch.zeros(hi - lo, dtype=torch.float32, device="cuda")
            gather_slabs(slab, n_grid)
            barrier()
            t1 = time.perf_counter()
            for _ in range(5):
                gather_slabs(slab, n_grid)
            barrier()
            t = torch.tensor([(time.perf_counter() - t1) / 5 * 1e3], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            c4["c4_all_gather_ms"] = float(t.item())
            del slab
        del grid, host

    # ---- reconstruct.py at its default options (mc_res0 64, two upsampling steps): MISE refinement on the
    #      device + HIP marching cubes, one object, rank 0 ----
    mesh_leg = None
    if args.mesh_steps > 0 and rank == 0 and args.img_size == 256:
        try:
            from slice3d_amd.generator import Generator3D
            fdm = {k: v[:1].contiguous() for k, v in make_feed_dict(1, args.img_size, 16, args.n_slices, seed=3, with_slices=False,
                                                                     device="cuda").items()}
            gm = Generator3D(model, threshold=0.5, resolution0=64, upsampling_steps=2, pred_type="sdf", mesh_backend="device")
            gm.generate_mesh(fdm)
            tms = []
            for _ in range(args.mesh_steps):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                mesh, mst = gm.generate_mesh(fdm)
                torch.cuda.synchronize()
                tms.append(time.perf_counter() - t1)
            mesh_leg = {"seconds_per_mesh": sorted(tms)[len(tms) // 2], "seconds_eval_points": mst.get("time (eval points)"),
                        "seconds_marching_cubes": mst.get("time (marching cubes)"), "vertices": int(len(mesh.vertices)),
                        "faces": int(len(mesh.faces))}
            del gm, mesh
        except Exception as e:   # the mesh library is a separate .so (csrc_mesh); the headline does not depend on it
            mesh_leg = {"error": repr(e)[:200]}

    # ---- BASELINE configs[4]: one denoising step of the gen_slices latent-diffusion U-Net (295 M parameters,
    #      64x64x4 latent mosaic + 4 conditioning channels, 21 attention blocks), batch 1 per GPU.  EVERY rank runs it
    #      on its own latents (configs[4] "1 -> 8 MI355X": independent samples, no collective); the reported time is the
    #      slowest rank's, the rate the aggregate over all ranks ----
    ldm = None
    if args.ldm_steps > 0:
        from slice3d_amd.ldm_unet import UNetModel
        cfg = dict(image_size=64, in_channels=8, out_channels=4, model_channels=192, attention_resolutions=[1, 2, 4, 8],
                   num_res_blocks=2, channel_mult=[1, 2, 2, 4, 4], num_heads=8, use_scale_shift_norm=True,
                   resblock_updown=True)
        um = load_seeded(UNetModel(prec=args.prec, **cfg), 0).cuda().eval()
        g = torch.Generator().manual_seed(rank)
        lx = torch.randn(1, 8, 64, 64, generator=g).cuda()
        lt = torch.tensor([500]).cuda()
        lc = {k: (torch.randn(1, c, r, r, generator=g) * 0.5).cuda()
              for k, (c, r) in (("f1", (192, 64)), ("f2", (384, 32)), ("f3", (384, 16)), ("f4", (768, 8)), ("f5", (768, 4)))}
        def time_ldm(x, t, cf):
            """ms per DDIM step of slice3d_amd.ldm_sampler.DDIMSampler (ddim.py:56-203: 200-step schedule, eta = 1, the
            conditioning concatenated on the channel axis): the loop captures the step's ~480 launches once into a HIP
            graph and replays it per step, then does the x_{t-1} update — what is timed is that loop itself, `ldm_steps`
            consecutive steps from x_T after a 3-step warm-up run (which includes the capture)."""
            from slice3d_amd.ldm_sampler import DDIMSampler
            smp = DDIMSampler(um)
            x_T, cc = x[:, :4].contiguous(), x[:, 4:].contiguous()
            gen = torch.Generator(device="cuda").manual_seed(rank)
            smp.sample(200, x_T, cc, cf, eta=1.0, generator=gen, n_steps=3)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            smp.sample(200, x_T, cc, cf, eta=1.0, generator=gen, n_steps=args.ldm_steps)
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / args.ldm_steps * 1e3, ("DDIM sampler loop, hip-graph replay" if smp._graph
                                                                        else "DDIM sampler loop, eager launches")

        lms, lmode = time_ldm(lx, lt, lc)
        # gen_slices LDM UNetModel denoise step, objaverse-ldm-kl-8.yaml, batch 1 = 222 GFLOP
        ldm = {"ms_per_step": lms, "tflops_algorithmic": 0.222 / lms * 1e3, "dtype": args.prec,
               "graph_replay": "graph" in lmode}
        # the same step on a batch of 4 latents (the sampler's classifier-free pair x 2 objects): what the small kernels of
        # the batch-1 step cost in utilisation
        lx4, lt4 = lx.repeat(4, 1, 1, 1).contiguous(), lt.repeat(4)
        lc4 = {k: v.repeat(4, 1, 1, 1).contiguous() for k, v in lc.items()}
        lms4, _ = time_ldm(lx4, lt4, lc4)
        ldm["batch4_ms_per_step"] = lms4
        # the precision configs[4] names: single-pass f16 convolutions (prec='f16'), with its deviation from the headline mode
        if args.prec == "f16x3":
            with torch.no_grad():
                ref_out = um(lx, lt, c_fmaps=lc)
            del um
            um = load_seeded(UNetModel(prec="f16", **cfg), 0).cuda().eval()
            lms16, _ = time_ldm(lx, lt, lc)
            with torch.no_grad():
                o16 = um(lx, lt, c_fmaps=lc)
            ldm["f16_ms_per_step"] = lms16
            ldm["f16_max_abs_diff_vs_headline_mode"] = float((o16 - ref_out).abs().max())
            del o16, ref_out
        del um, lx, lc, lx4, lc4
        # the 128 x 128 latent of 256^2 slice generation (configs[4]): convolutions x4, the 16 384-token attention x16
        um = load_seeded(UNetModel(prec=args.prec, **dict(cfg, image_size=128)), 0).cuda().eval()
        lx = torch.randn(1, 8, 128, 128, generator=g).cuda()
        lc = {k: (torch.randn(1, c, 2 * r, 2 * r, generator=g) * 0.5).cuda()
              for k, (c, r) in (("f1", (192, 64)), ("f2", (384, 32)), ("f3", (384, 16)), ("f4", (768, 8)), ("f5", (768, 4)))}
        lms128, _ = time_ldm(lx, lt, lc)
        del um, lx, lc
        if dist is not None:
            t = torch.tensor([lms, lms4, lms128], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            lms, lms4, lms128 = (float(v) for v in t.tolist())
            ldm["ms_per_step"], ldm["batch4_ms_per_step"] = lms, lms4
            ldm["tflops_algorithmic"] = 0.222 / lms * 1e3
        ldm["latent128_ms_per_step"] = lms128
        ldm["n_gpus"] = world
        ldm["steps_per_s_all_gpus"] = world / (lms * 1e-3)
        ldm["latent128_steps_per_s_all_gpus"] = world / (lms128 * 1e-3)

    # ---- secondary metric: Slices3DGTModel training (train_gt.py:38-52) at the reference's default options
    #      (options.py: n_bs 16, img_size 128, n_qry 256), rank 0 only (the other ranks wait at the next leg's barrier) ----
    gt_train = None
    if args.gt_train_steps > 0 and rank == 0:
        from slice3d_amd.models_gt import Slices3DGTModel
        from slice3d_amd.trainer import HipGtTrainer
        gm = load_seeded(Slices3DGTModel(img_size=128, n_slices=args.n_slices, mode="train"), 0).cuda()
        gtr = HipGtTrainer(gm, dropout=0.1, seed=0, prec=args.prec, process_group=False)   # rank 0 alone: no exchange
        gfd = make_feed_dict(16, 128, 256, args.n_slices, seed=99, device="cuda")
        gtr.train_step(gfd)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.gt_train_steps):
            gtr.train_step(gfd)
        torch.cuda.synchronize()
        gms = (time.perf_counter() - t1) / args.gt_train_steps * 1e3
        # Slices3DGTModel train_step (fwd + L1 + bwd + Adam), 16 objects x 12 slices at 128^2, 256 queries each, dropout 0.1
        gt_train = {"ms_per_step": gms, "samples_per_s": 16 / gms * 1e3}
        del gtr, gm, gfd

    # ---- secondary metric: training samples/s (train.py:41-53 train_step, B = 1 object per GPU) ----
    train_ms = train_ms_local = None
    if args.train_steps > 0:
        from slice3d_amd.trainer import HipTrainer
        tmodel = Slices3DRegModel(img_size=args.img_size, n_slices=args.n_slices, mode="train")
        load_seeded(tmodel, 0)
        tmodel.cuda()
        trainer = HipTrainer(tmodel, dropout=0.1, seed=rank, prec=args.prec)   # dropout: the reference's default
        tfd = make_feed_dict(args.batch, args.img_size, args.n_qry, args.n_slices, seed=4321 + rank, device="cuda")
        trainer.train_step(tfd)                      # warm-up (allocates the ~25 GB activation workspace)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.train_steps):
            tl = trainer.train_step(tfd)
        barrier()
        tdt = time.perf_counter() - t1
        if dist is not None:
            t = torch.tensor([tdt], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tdt = float(t.item())
        train_ms = tdt / args.train_steps * 1e3
        del trainer, tmodel
        if world > 1:   # the same step with the exchange switched off (process_group=False: a lone replica per rank): what the
                        # bucketed gradient all-reduce leaves exposed beside the backward it overlaps with
            tmodel = Slices3DRegModel(img_size=args.img_size, n_slices=args.n_slices, mode="train")
            load_seeded(tmodel, 0)
            tmodel.cuda()
            trainer = HipTrainer(tmodel, dropout=0.1, seed=rank, prec=args.prec, process_group=False)
            trainer.train_step(tfd)
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.train_steps):
                trainer.train_step(tfd)
            barrier()
            t = torch.tensor([time.perf_counter() - t1], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            train_ms_local = float(t.item()) / args.train_steps * 1e3
            del trainer, tmodel

    # ---- throughput mode of the training step (NOT the reported train_samples_per_s): prec="f16" — the decoder's GEMM kernels run one
    #      f16 MFMA per product (configs[1] names bf16: this is the reduced-precision leg of the step), everything else as f16x3.
    #      Rank 0 alone, no exchange; with the gradient deviation of one small step from the split-precision step (same weights,
    #      batch and dropout masks): median and 95th percentile of the per-tensor relative L2 distance ----
    train_f16 = None
    if args.train_f16_steps > 0 and args.train_steps > 0 and rank == 0:
        from slice3d_amd.trainer import HipTrainer

        def _grads(prec):
            tm = load_seeded(Slices3DRegModel(img_size=64, n_slices=args.n_slices, mode="train"), 0).cuda()
            tt = HipTrainer(tm, dropout=0.1, seed=11, prec=prec, process_group=False)
            tt.forward_backward(make_feed_dict(2, 64, 8192, args.n_slices, seed=17, device="cuda"))
            return {k: p.grad.detach().clone() for k, p in tm.named_parameters() if p.grad is not None}
        g3, g1 = _grads("f16x3"), _grads("f16")
        # (the eight pre-BatchNorm conv biases have an exactly-zero gradient — rounding noise in every mode — and are left out)
        pre_bn = {"slices_generator.%s.bias" % k for k in ("down1.0", "down2.7", "down3.14", "down3.17", "down4.24", "down4.27", "down5.34", "down5.37")}
        dev = sorted(float((g1[k] - g).norm() / g.norm()) for k, g in g3.items() if k not in pre_bn and float(g.norm()) > 0)
        del g3, g1
        tmodel = load_seeded(Slices3DRegModel(img_size=args.img_size, n_slices=args.n_slices, mode="train"), 0).cuda()
        trainer = HipTrainer(tmodel, dropout=0.1, seed=0, prec="f16", process_group=False)
        tfd16 = make_feed_dict(args.batch, args.img_size, args.n_qry, args.n_slices, seed=4321, device="cuda")
        trainer.train_step(tfd16)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.train_f16_steps):
            trainer.train_step(tfd16)
        torch.cuda.synchronize()
        ms16 = (time.perf_counter() - t1) / args.train_f16_steps * 1e3
        train_f16 = {"ms_per_step": ms16, "samples_per_s": args.batch / (ms16 * 1e-3), "dtype": "f16 decoder, f16x3 rest",
                     "grad_rel_dev_vs_f16x3": {"median": dev[len(dev) // 2], "p95": dev[int(len(dev) * 0.95)]}}
        del trainer, tmodel, tfd16

    if rank == 0:
        q_total = args.n_qry * args.batch * world * args.steps
        n_tok = args.n_slices + 1
        ffn_launches = max(counts["ffn_layer"], 1)
        ffn_ms = stage_ms["ffn_layer"] * args.steps / ffn_launches
        # algorithmic FLOPs of the average launch (a step's 2 full FFN layers are launched on <= 262 144 queries at a time)
        ffn_flops = 2.0 * n_tok * args.n_qry * args.batch * FFN_FLOP_PER_ROW * args.steps / ffn_launches
        achieved = ffn_flops / (ffn_ms * 1e-3) / 1e12
        peak = F32_MFMA_PEAK_TFLOPS if args.prec == "f32" else F16_MFMA_PEAK_TFLOPS
        # HBM bytes per launch of the dominant kernel: measured in this run by two counter passes (FETCH_SIZE, WRITE_SIZE:
        # separate rocprofv3 --pmc runs of the same timed loop, MI355X_MICROARCH.md's recipe and gfx950 correction);
        # only if that is impossible the figure of the committed pass is quoted, and the source field says which
        kname = "ffn_layer_kernel" if args.prec == "f32" else "ffn_layer_f16x3_pipe_kernel"   # base name: any template arguments
        traffic, traffic_src = (None, None)
        if args.pmc and world == 1:
            traffic, traffic_src = _pmc_traffic(args, kname)
        if traffic is None:
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
                wl = pmc["workload"]
                if (wl["img_size"], wl["n_slices"], wl["n_qry"], wl.get("batch", 1)) == (args.img_size, args.n_slices,
                                                                                          args.n_qry, args.batch):
                    jk = "ffn_layer_kernel<false>" if args.prec == "f32" else "ffn_layer_f16x3_pipe_kernel<false>"
                    traffic = pmc["kernels"][args.prec][jk]["hbm_bytes_per_launch"]
                    traffic_src = "committed pass profiles/pmc_traffic.json (not measured in this run: %s)" % (
                        "disabled" if not args.pmc else "multi-rank run" if world > 1 else "rocprofv3 pass failed")
                    if args.pmc and world == 1:
                        print("bench: roofline.traffic falls back to the committed figure", file=sys.stderr)
            except (OSError, KeyError, ValueError):
                pass
        decode_ms = sum(stage_ms[k] for k in ("sample_tokens", "attn_layer", "ffn_layer", "ffn_final"))
        unet_tf = args.batch * UNET_GFLOP_256 * (args.img_size / 256.0) ** 2 / stage_ms["unet_encode"]
        attn_tf = args.n_qry * args.batch * 2 * n_tok * ATTN_FLOP_PER_TOKEN / (stage_ms["attn_layer"] * 1e-3) / 1e12
        # algorithmic bytes of the token builder per step: the (queries x 13) token rows written once + every pyramid level of the
        # batch x n_slices slice images read once (folded levels 0-2: 128 channels at S/16, S/8, S/4; raw levels: 64 at S/2, 32 at S)
        S_ = args.img_size
        pyr_px_ch = 128 * ((S_ // 16) ** 2 + (S_ // 8) ** 2 + (S_ // 4) ** 2) + 64 * (S_ // 2) ** 2 + 32 * S_ ** 2
        tok_bytes = 4.0 * (args.batch * args.n_qry * n_tok * 128 + args.batch * args.n_slices * pyr_px_ch)
        # Key order: the driver keeps ~2 KB of the line's tail, so the contract keys and the long objects come first and
        # every secondary result sits at the end, numbers only (what each key means: DESIGN.md section 5, "bench line glossary").
        res = {
            "metric": "occupancy query-points/sec (U-Net encode + per-query decode, 256^2 x 12 slices)",
            "value": q_total / dt, "unit": "query-points/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.prec, "data": "synthetic-smooth",
            "config": {"workload": "reg_slices inference %d^2 x %d slices, %d queries/object, %d objects/GPU/step (configs[1])"
                                   % (args.img_size, args.n_slices, args.n_qry, args.batch),
                       "img_size": args.img_size, "n_slices": args.n_slices, "n_qry": args.n_qry,
                       "objects_per_step": world * args.batch, "parallelism": "objects x%d (no collective)" % world},
            "ms_per_step_rank_min_max": [min(rank_ms), max(rank_ms)],
            "roofline": {"kernel": ("ffn_layer_kernel<false>" if args.prec == "f32" else "ffn_layer_f16x3_pipe_kernel<0>")
                                   + " (decoder FFN + residual + LN2)",
                         "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "note": "f16x3: pipe executes 3x" if args.prec != "f32" else "exact fp32 MFMA",
                         "mfma_pipe_tflops": achieved * (3 if args.prec != "f32" else 1),
                         "traffic_source": traffic_src,
                         "avg_launch_ms": ffn_ms, "launches": counts["ffn_layer"],
                         "alg_flop_per_launch": ffn_flops},
            "secondary_rooflines": [
                sample_roof,
                {"kernel": "U-Net conv stack", "bound": "mfma", "achieved": unet_tf, "peak": peak, "unit": "TFLOP/s",
                 "frac": unet_tf / peak},
                {"kernel": "attention stage (2 layers + fused last layer)", "bound": "mfma",
                 "achieved": attn_tf, "peak": peak, "unit": "TFLOP/s", "frac": attn_tf / peak},
                # the fused feature-sample + fc_s kernel of the decode path (north_star's "feature-sample kernel" inside the pipeline): token
                # tensor written once (rows x 512 B) + the five pyramid levels of the B x n_slices images read once, over its stage time
                {"kernel": "sample_tokens_kernel (fused feature-sample + fc_s)", "bound": "hbm",
                 "achieved": tok_bytes / (stage_ms["sample_tokens"] * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                 "frac": tok_bytes / (stage_ms["sample_tokens"] * 1e-3) / 1e9 / 8000.0, "kernel_ms": stage_ms["sample_tokens"]},
            ],
            "decode_tflops_fmin": args.n_qry * args.batch * F_MIN_PER_QUERY / (decode_ms * 1e-3) / 1e12,
        }
        err = None
        if world == 1 and args.cpu_sample > 0:
            base, err = cpu_baseline(sd_cpu, {k: v[:1] for k, v in fd.items()}, args.n_slices,
                                     min(args.cpu_sample, args.n_qry), out[:1], runs=args.cpu_runs)
            res["cpu_baseline"] = base
        # ---- the tail: every secondary number of the run ----
        res["stage_ms_per_step"] = {k: v for k, v in stage_ms.items() if v}
        if err is not None:
            res["parity_vs_oracle"] = {"max_abs_err": err, "n": min(args.cpu_sample, args.n_qry), "tol": 1e-4}
        res["exact_f32_mode"] = f32_mode
        res["white_noise"] = noise_leg
        res["throughput_mode_f16"] = f16_mode
        res["throughput_mode_bf16"] = bf16_mode
        res["c4_dense_grid"] = c4
        res["ldm_denoise_step"] = ldm
        res["mesh_extraction"] = mesh_leg
        res["gt_train_step"] = gt_train
        res["train_throughput_mode_f16"] = train_f16
        res["train_ms_per_step"] = train_ms
        res["train_samples_per_s"] = (world * args.batch / (train_ms * 1e-3)) if train_ms else None
        if train_ms_local is not None:   # N > 1 only (tools/scaling_table.py reads these)
            res["train_ms_per_step_no_exchange"] = train_ms_local
            res["train_allreduce_ms_exposed"] = train_ms - train_ms_local
        exact = ("value", "ms_per_step", "roofline")      # contract numbers stay unrounded (value == queries / time exactly)
        res = {k: (v if k in exact else _r(v)) for k, v in res.items()}
        res["roofline"] = {k: (v if k in ("achieved", "peak", "frac") else _r(v)) for k, v in res["roofline"].items()}
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
