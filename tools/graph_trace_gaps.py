#!/usr/bin/env python3
"""Where a replayed hipGraph step loses time against the eager step: from ONE rocprofv3 --kernel-trace
of tools/graph_step.py, per training step (= the launches up to and including an Adam launch): wall time from the first
kernel's start to Adam's end, the time the device runs at least one kernel (union of the kernel intervals), the idle time
inside the step, the summed kernel time and how many kernels overlap another one.  The eager steps come first in that
tool's run, the replays last; steps are told apart by the launch count (capture itself launches nothing).

    python tools/graph_trace_gaps.py <dir with *_kernel_trace.csv> [--last N]
"""
import csv, glob, os, sys


def main():
    path = sys.argv[1]
    files = glob.glob(os.path.join(path, "**", "*_kernel_trace.csv"), recursive=True) if os.path.isdir(path) else [path]
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    steps, cur = [], []
    for r in rows:
        cur.append(r)
        if "k_adam" in r["Kernel_Name"]:
            steps.append(cur)
            cur = []
    out = []
    for k, st in enumerate(steps):
        iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in st)
        t0, t1 = iv[0][0], max(e for _, e in iv)
        busy, ce, overl, gaps = 0, iv[0][0], 0, []
        cs = iv[0][0]
        for s, e in iv:
            if s > ce:
                busy += ce - cs
                gaps.append(s - ce)
                cs = s
            elif s < ce and (s, e) != iv[0]:
                overl += 1
            ce = max(ce, e)
        busy += ce - cs
        queues = sorted({r.get("Queue_Id", "?") for r in st})
        out.append(dict(step=k, n=len(st), wall_us=(t1 - t0) / 1e3, busy_us=busy / 1e3, idle_us=(t1 - t0 - busy) / 1e3,
                        kernel_us=sum(e - s for s, e in iv) / 1e3, overlapped=overl, queues=len(queues),
                        gaps_over_2us=sum(1 for g in gaps if g > 2000), max_gap_us=max(gaps) / 1e3 if gaps else 0.0))
    print("# step  launches  wall_us  busy_us  idle_us  kernel_us  overlapped  queues  gaps>2us  max_gap_us")
    for o in out:
        print(f"{o['step']:6d} {o['n']:9d} {o['wall_us']:8.1f} {o['busy_us']:8.1f} {o['idle_us']:8.1f} {o['kernel_us']:10.1f} "
              f"{o['overlapped']:10d} {o['queues']:7d} {o['gaps_over_2us']:9d} {o['max_gap_us']:10.1f}")


if __name__ == "__main__":
    main()
