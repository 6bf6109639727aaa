#!/usr/bin/env python3
"""Development helper: average duration (us) of the kernels whose name contains a pattern, from a rocprofv3 kernel_stats.csv.
    python tools/kernel_avg.py <dir with *kernel_stats.csv> <pattern> [<pattern> ...]"""
import csv, glob, os, sys
def main():
    d, pats = sys.argv[1], sys.argv[2:]
    f = [p for p in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)]
    if not f:
        print("no kernel_stats.csv under", d); return
    rows = list(csv.DictReader(open(f[0])))
    for p in pats:
        for r in rows:
            if p in r["Name"]:
                print(f"{p:28s} calls {r['Calls']:>6s}  avg {float(r['AverageNs']) / 1e3:8.2f} us  {r['Name'][:60]}")
if __name__ == "__main__":
    main()
