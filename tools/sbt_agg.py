import sys, re
rows=[]
for l in sys.stdin:
    m=re.match(r'SBT nq (\d+) init (\d+) raw (\d+) folded (\d+) fflush (\d+) rflush (\d+)', l)
    if m: rows.append([int(x) for x in m.groups()])
n=len(rows)//8  # first step's prints only if many
print("WGs sampled", len(rows))
import statistics
tot=[sum(r[i] for r in rows) for i in range(6)]
print("sum queries %d; cycles: init %.3g raw %.3g folded %.3g fflush %.3g rflush %.3g" % tuple(tot))
rows.sort()
for q in (0, len(rows)//4, len(rows)//2, 3*len(rows)//4, len(rows)-1): print(rows[q])
