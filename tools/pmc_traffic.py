#!/usr/bin/env python3
"""Memory-side bytes per launch of the bench's roofline kernels, from two separate rocprofv3 PMC passes over bench.py
(FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950: MI355X_MICROARCH.md, "rocprofv3 PMC slots"):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write --workload B256_d256_nb2_L8 > profiles/pmc_traffic.json

Corrections, as the guide's HBM section prescribes: the counters are in KB; on gfx950 FETCH_SIZE tallies the 128-byte
requests of wide (16 B per lane) streaming reads at 64 bytes, so reads are doubled; WRITE_SIZE is exact for 16-byte
streaming stores and float atomics.  Infinity-Cache hits are included (memory-side of the L2).  bench.py reports these
numbers as `roofline.traffic` only while the digest of the kernel sources and the workload match.
"""
import argparse
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.trace_summary import short  # noqa: E402


def klass(kernel_name):
    """bench.py's class name of a kernel (gemm_<layout>_<tile> / segreduce_fwd / segreduce_bwd / gcl_fwd) or None."""
    s = short(kernel_name)
    if s.startswith("k_gemm<"):
        body = s[len("k_gemm<"):-1].split(",")
        return f"gemm_{body[1]}_{body[0]}"
    if s.startswith("k_segreduce_fwd"):
        return "segreduce_fwd"
    if s.startswith("k_segreduce_bwd"):
        return "segreduce_bwd"
    for k in ("gcl_fwd", "gcl_dagg", "gcl_dw"):
        if s.startswith("k_" + k):
            return k
    if s.startswith("k_rows_w"):
        return "gemm_NN_rows_w"
    if s.startswith("k_wide<"):                    # wide.hip: k_wide<VAR, DROP, NPW, BKIND>, VAR 0/1 forward, 2 input gradient, 3/4 chord products
        var = int(s[len("k_wide<"):].split(",")[0])
        return {0: "gcl_fwd", 1: "gcl_fwd", 2: "gcl_dagg", 3: "gemm_NN_rows_w", 4: "gemm_NN_rows_w"}.get(var)
    return None


def averages(directory, counter):
    acc = defaultdict(list)
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {directory}")
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = klass(r["Kernel_Name"])
                if k:
                    acc[k].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--workload", required=True)
    a = ap.parse_args()
    from bench import kernel_source_digest
    rd, n = averages(a.fetch_dir, "FETCH_SIZE")
    wr, _ = averages(a.write_dir, "WRITE_SIZE")
    out = {"workload": a.workload, "source_digest": kernel_source_digest(),
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), KB -> bytes, reads x2 (gfx950 "
                     "FETCH_SIZE rule), averaged over the launches of each kernel class; Infinity-Cache hits included",
           "bytes_per_launch": {}, "detail": {}}
    for k in sorted(set(rd) | set(wr)):
        r, w = 2.0 * 1024.0 * rd.get(k, 0.0), 1024.0 * wr.get(k, 0.0)
        out["bytes_per_launch"][k] = round(r + w)
        out["detail"][k] = {"read_bytes_corrected": round(r), "write_bytes": round(w), "launches_averaged": n.get(k, 0)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
