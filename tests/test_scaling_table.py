"""tools/scaling_table.py on synthetic bench lines (no GPU): the efficiency table DESIGN.md section 6 is waiting for."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(n, value, ms, train=None, c4=None, extra=None):
    d = {"metric": "m", "value": value, "unit": "query-points/s", "n_gpus": n, "ms_per_step": ms, "ms_per_step_rank_min_max": [ms * 0.99, ms],
         "train_samples_per_s": train, "train_ms_per_step": (4 * n / train * 1e3) if train else None,
         "c4_dense_grid": ({"seconds_device": c4} if c4 else None)}
    d.update(extra or {})
    return d


def test_scaling_table_computes_weak_and_strong_efficiency(tmp_path):
    lines = [_line(1, 1.0e7, 40.0, train=20.0, c4=1.2),
             _line(2, 1.9e7, 42.1, train=38.0, c4=0.65, extra={"train_allreduce_ms_exposed": 3.5}),
             # a driver record: the line under "parsed", spread over several lines of a JSON file
             {"rc": 0, "parsed": _line(8, 7.2e7, 44.4, train=144.0, c4=0.2, extra={"train_allreduce_ms_exposed": 9.25})}]
    lines[1]["c4_dense_grid"]["c4_all_gather_ms"] = 1.5
    paths = []
    for i, d in enumerate(lines):
        p = tmp_path / ("b%d.json" % i)
        p.write_text(json.dumps(d, indent=1 if i == 2 else None))
        paths.append(str(p))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scaling_table.py")] + paths[::-1], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = [ln.split("|") for ln in r.stdout.splitlines() if ln.startswith("| ") and ln[2].isdigit()]
    assert [int(c[1]) for c in rows] == [1, 2, 8]                       # sorted by GPU count whatever the argument order
    eff = {int(c[1]): float(c[3]) for c in rows}
    assert eff == {1: 1.0, 2: 0.95, 8: 0.9}                             # value(N) / (N value(1))
    tr = {int(c[1]): c[6].strip() for c in rows}
    assert tr[2] == "0.950" and tr[8] == "0.900"
    strong = {int(c[1]): c[10].strip() for c in rows}
    assert strong[2] == "0.923" and strong[8] == "0.750"                # t(1) / (N t(N))
    assert rows[1][8].strip() == "3.50" and rows[1][11].strip() == "1.50" and rows[0][8].strip() == "—"
    # without the N = 1 line there is no reference
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scaling_table.py"), paths[1]], capture_output=True, text=True)
    assert r.returncode != 0 and "N = 1" in (r.stderr + r.stdout)
