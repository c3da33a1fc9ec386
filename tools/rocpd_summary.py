"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a markdown/CSV-like table:
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/xxx_kernel_stats.md
Equivalent of `rocprofv3 --kernel-trace --stats`'s kernel_stats.csv (this ROCm writes a .db by default)."""
import sqlite3
import sys


def main(path, skip_first=0):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count "
                      "from kernels order by start").fetchall()
    stats = {}
    for name, st, en, gx, wx, lds, vg, ag in rows:
        s = stats.setdefault(name, {"n": 0, "tot": 0, "min": 1 << 62, "max": 0, "grid": gx, "wg": wx, "lds": lds,
                                    "vgpr": vg, "agpr": ag})
        d = en - st
        s["n"] += 1; s["tot"] += d; s["min"] = min(s["min"], d); s["max"] = max(s["max"], d)
        s["grid"] = max(s["grid"], gx)
    total = sum(s["tot"] for s in stats.values())
    print("| kernel | calls | total ms | avg us | min us | max us | % | max grid | wg | LDS B | VGPR | AGPR |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, s in sorted(stats.items(), key=lambda kv: -kv[1]["tot"]):
        short = name if len(name) < 90 else name[:87] + "..."
        print("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f | %d | %d | %d | %d | %d |" % (
            short, s["n"], s["tot"] / 1e6, s["tot"] / s["n"] / 1e3, s["min"] / 1e3, s["max"] / 1e3,
            100.0 * s["tot"] / total, s["grid"], s["wg"], s["lds"], s["vgpr"], s["agpr"]))
    print("\ntotal kernel time %.3f ms over %d dispatches" % (total / 1e6, len(rows)))


if __name__ == "__main__":
    main(sys.argv[1])
