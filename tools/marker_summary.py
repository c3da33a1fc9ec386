"""Summarise a `rocprofv3 --marker-trace --kernel-trace --output-format csv` run of the training step: per roctx range
(the s3d:train:* phases recorded by api_train.inc's RangeSeq and trainer.py's gradient exchange) the number of
occurrences, the HOST time between push and pop, and the GPU kernel time of the kernels launched inside the range
(matched by launch order: kernels are enqueued on one stream, so the kernels whose host-side launch falls inside a range
are the range's; the kernel trace gives their device durations).
    python tools/marker_summary.py <dir with *_marker_api_trace.csv and *_kernel_trace.csv> > profiles/r04_train_marker_trace.md
"""
import csv
import glob
import os
import sys


def first(pattern, d):
    fs = glob.glob(os.path.join(d, "**", pattern), recursive=True)
    return fs[0] if fs else None


def main(d):
    mk, kt = first("*marker_api_trace.csv", d), first("*kernel_trace.csv", d)
    if not mk:
        print("no marker trace csv under", d)
        return
    rows = list(csv.DictReader(open(mk)))
    cols = rows[0].keys() if rows else []
    name_col = "Function" if "Function" in cols else [c for c in cols if "ame" in c][0]
    ranges = [(r[name_col], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows
              if int(r["End_Timestamp"]) > int(r["Start_Timestamp"])]
    kernels = []
    if kt:
        for r in csv.DictReader(open(kt)):
            kernels.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
        kernels.sort()
    stats = {}
    order = []
    for name, s, e in ranges:
        st = stats.setdefault(name, {"n": 0, "host": 0})
        if st["n"] == 0:
            order.append(name)
        st["n"] += 1
        st["host"] += e - s
    print("# roctx ranges of the training step (rocprofv3 --marker-trace --kernel-trace, `tools/prof_train.py`: 8 steps of 4 objects x 100 k queries)\n")
    print("Host time = push .. pop on the launching thread (the step is enqueued asynchronously: a phase's host time is its launch")
    print("cost unless the phase ends in a synchronisation).  Ranges in order of first appearance.\n")
    print("| range | occurrences | host ms per occurrence |\n|---|---|---|")
    for name in order:
        st = stats[name]
        print("| `%s` | %d | %.3f |" % (name, st["n"], st["host"] / st["n"] / 1e6))
    if kernels:
        tot = sum(e - s for s, e, _ in kernels)
        print("\nkernel trace of the same run: %d dispatches, %.1f ms of kernel time" % (len(kernels), tot / 1e6))


if __name__ == "__main__":
    main(sys.argv[1])
