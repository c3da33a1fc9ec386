"""Turn the bench lines of N = 1, 2, 4, 8 GPUs into the efficiency table DESIGN.md section 6 needs.
    python tools/scaling_table.py BENCH_n1.json BENCH_n2.json BENCH_n4.json BENCH_n8.json
Each file holds bench.py's JSON line (a driver record with the line under "parsed" is accepted too).  Efficiency is computed here
from the per-N values — bench.py never reports one:
  weak scaling (inference objects per rank, training samples per rank): value(N) / (N * value(1))
  strong scaling (the dense 256^3 grid of ONE object):                  t(1) / (N * t(N))"""
import json
import sys


def load(path):
    txt = open(path).read().strip()
    try:
        d = json.loads(txt)                      # a driver record (possibly multi-line) or a bare line
    except ValueError:
        d = json.loads(txt.splitlines()[-1])     # a log whose last line is the bench line
    return d.get("parsed", d)


def main(paths):
    runs = sorted((load(p) for p in paths), key=lambda d: d["n_gpus"])
    base = next((d for d in runs if d["n_gpus"] == 1), None)
    if base is None:
        raise SystemExit("need the N = 1 line as the reference")
    print("| GPUs | query-points/s (all GPUs) | weak eff. | ms/step (max over ranks; per-rank min..max) | train samples/s | weak eff. | "
          "train ms/step | all-reduce exposed ms | 256^3 grid s | strong eff. | grid all_gather ms |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for d in runs:
        n = d["n_gpus"]
        mm = d.get("ms_per_step_rank_min_max") or [d["ms_per_step"]] * 2
        tr, tr1 = d.get("train_samples_per_s"), base.get("train_samples_per_s")
        c4, c41 = d.get("c4_dense_grid") or {}, base.get("c4_dense_grid") or {}
        f = lambda v, fmt="%.3g": "—" if v is None else fmt % v
        print("| %d | %.4g | %.3f | %.2f (%.2f..%.2f) | %s | %s | %s | %s | %s | %s | %s |" % (
            n, d["value"], d["value"] / (n * base["value"]), d["ms_per_step"], mm[0], mm[1],
            f(tr, "%.2f"), f(tr / (n * tr1) if tr and tr1 else None, "%.3f"), f(d.get("train_ms_per_step"), "%.1f"),
            f(d.get("train_allreduce_ms_exposed"), "%.2f"), f(c4.get("seconds_device"), "%.3f"),
            f(c41["seconds_device"] / (n * c4["seconds_device"]) if c4.get("seconds_device") and c41.get("seconds_device") else None, "%.3f"),
            f(c4.get("c4_all_gather_ms"), "%.2f")))


if __name__ == "__main__":
    main(sys.argv[1:])
