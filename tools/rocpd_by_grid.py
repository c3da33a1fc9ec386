"""Kernel durations of a rocprofv3 rocpd database grouped by (kernel name, grid): python tools/rocpd_by_grid.py DB [substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start").fetchall()
st = {}
for name, s, e, gx, gy, gz, wx in rows:
    if sub not in name:
        continue
    k = (name[:70], gx // max(wx, 1), gy, gz)
    v = st.setdefault(k, [0, 0, 1 << 62, 0])
    d = e - s
    v[0] += 1; v[1] += d; v[2] = min(v[2], d); v[3] = max(v[3], d)
print("| kernel | workgroups x | y | z | calls | avg us | min us | max us |")
print("|---|---|---|---|---|---|---|---|")
for k, v in sorted(st.items(), key=lambda kv: -kv[1][1]):
    print("| `%s` | %d | %d | %d | %d | %.1f | %.1f | %.1f |" % (k[0], k[1], k[2], k[3], v[0], v[1] / v[0] / 1e3, v[2] / 1e3, v[3] / 1e3))
