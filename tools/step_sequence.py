#!/usr/bin/env python3
"""The launch sequence of ONE training step, from a rocprofv3 --kernel-trace CSV: every kernel between the last two
Adam launches of the run, in start order, with duration and the gap to the previous kernel's end.

    python tools/step_sequence.py <dir with *_kernel_trace.csv> > profiles/<round>_step_sequence.txt
"""
import csv, glob, os, re, sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", name)[:64]


def main():
    path = sys.argv[1]
    files = glob.glob(os.path.join(path, "**", "*_kernel_trace.csv"), recursive=True) if os.path.isdir(path) else [path]
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "k_adam" in r["Kernel_Name"]]
    if len(adam) < 2:
        sys.exit("fewer than two Adam launches in the trace")
    a, b = adam[-2] + 1, adam[-1] + 1
    t0, prev_end = int(rows[a]["Start_Timestamp"]), None
    tot = 0
    print(f"# {b - a} launches, {(int(rows[b - 1]['End_Timestamp']) - t0) / 1e3:.1f} us from first start to Adam's end")
    qkey = "Stream_Id" if "Stream_Id" in rows[a] and len({r["Stream_Id"] for r in rows[a:b]}) > 1 else ("Queue_Id" if "Queue_Id" in rows[a] else None)
    qids = {}
    print("#  idx   start_us    dur_us    gap_us  q  grid            kernel      (q: stream / queue of the launch, numbered in order of appearance)")
    for i, r in enumerate(rows[a:b]):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        prev_end = max(e, prev_end or e)
        grid = "x".join(str(int(r[k]) // max(1, int(r[w]))) for k, w in (("Grid_Size_X", "Workgroup_Size_X"), ("Grid_Size_Y", "Workgroup_Size_Y"), ("Grid_Size_Z", "Workgroup_Size_Z")))
        tot += e - s
        q = qids.setdefault(r[qkey], len(qids)) if qkey else 0
        print(f"{i:6d} {(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {gap:9.1f}  {q}  {grid:14s}  {short(r['Kernel_Name'])}")
    print(f"# kernel time {tot / 1e3:.1f} us")


if __name__ == "__main__":
    main()
