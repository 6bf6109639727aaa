#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters (one row per kernel class, counters as columns).
    python tools/pmc_kernels.py <dir-or-csv> [name-substring ...]"""
import csv, glob, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.trace_summary import short

src = sys.argv[1]
files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
subs = sys.argv[2:] or ["k_gemm<planes", "k_segreduce"]
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        name = short(r["Kernel_Name"])
        grid = int(r.get("Grid_Size", 0) or 0)
        if any(s in name for s in subs):
            acc[(name, grid)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, grid), d in sorted(acc.items()):
    n = len(next(iter(d.values())))
    print(f"{k}  grid={grid}  (n={n})")
    for c, v in sorted(d.items()):
        print(f"    {c:28s} {sum(v) / len(v):16.1f}")
