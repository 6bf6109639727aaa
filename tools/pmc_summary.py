"""Average rocprofv3 --pmc counters per kernel (name filter) from a *_counter_collection.csv.
    python tools/pmc_summary.py <csv> [substring]"""
import csv
import sys
from collections import defaultdict

sub = sys.argv[2] if len(sys.argv) > 2 else "k_gemm"
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if sub in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} n={len(v):3d} avg={sum(v) / len(v):16.1f}")
