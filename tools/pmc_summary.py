"""Summarise two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE counter_collection CSVs) per kernel.

usage: python tools/pmc_summary.py <fetch.csv> <write.csv> <prec> <out.md> [--json profiles/pmc_traffic.json]

Counter unit is KiB.  Corrected HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 this
rocprofv3 reports half of the bytes of a wide streaming read (MI355X_MICROARCH.md, "HBM"); for narrow
accesses the doubled figure is an upper bound."""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.match(r"_Z\d+(\w+?_kernel)IL[bi](\d)E", name)
    if m:
        flag = m.group(2)
        if "ffn_layer_f16x3" in name or "attn_layer_f16x3" in name:
            return "%s<%s>" % (m.group(1), {"0": "false", "1": "true"}.get(flag, flag))
    m = re.match(r"(?:void )?([\w:]+(?:<[^>]*>)?)\(", name)
    return m.group(1) if m else name


def load(path, counter):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


def main():
    fetch_csv, write_csv, prec, out_md = sys.argv[1:5]
    jpath = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    rows = []
    for k in f:
        # skip the warm-up dispatches' first launch of each kernel? keep all: the workload is stationary
        fa = sum(f[k]) / len(f[k])
        wa = sum(w[k]) / len(w[k]) if k in w and w[k] else 0.0
        rows.append((k, len(f[k]), fa, wa, (2 * fa + wa) * 1024))
    rows.sort(key=lambda r: -r[4] * r[1])
    with open(out_md, "w") as o:
        o.write("# rocprofv3 PMC passes (HBM traffic), --prec %s\n\n" % prec)
        o.write("`rocprofv3 --pmc FETCH_SIZE --output-format csv` and `rocprofv3 --pmc WRITE_SIZE --output-format csv` "
                "(separate passes) of\n`python bench.py --steps 3 --warmup 1 --cpu-sample 0 --train-steps 0 --ldm-steps 0`.  "
                "Counter unit KiB; corrected bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024\n(gfx950 FETCH_SIZE halving per "
                "MI355X_MICROARCH.md; an upper bound for narrow loads).\n\n")
        o.write("| kernel | launches | FETCH_SIZE avg KiB | WRITE_SIZE avg KiB | corrected HBM bytes / launch |\n|---|---|---|---|---|\n")
        for k, n, fa, wa, b in rows[:24]:
            o.write("| `%s` | %d | %.0f | %.0f | %.3e |\n" % (k[:90], n, fa, wa, b))
    if jpath:
        try:
            J = json.load(open(jpath))
        except OSError:
            J = {"workload": {"img_size": 256, "n_slices": 12, "n_qry": 100000, "batch": 4}, "source": {}, "kernels": {}}
        J["workload"]["batch"] = 4      # bench.py default: 4 objects per step
        J["source"][prec] = out_md
        J["kernels"][prec] = {short(k): {"fetch_kib": fa, "write_kib": wa, "hbm_bytes_per_launch": b}
                              for k, n, fa, wa, b in rows[:24]}
        json.dump(J, open(jpath, "w"), indent=1)


if __name__ == "__main__":
    main()
