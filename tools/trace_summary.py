"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) launch count, average and total duration.

    python tools/trace_summary.py gpurun_out/prof/<pid>_kernel_trace.csv|<name>_results.db [--steps K] [--skip-frac F] [--md]

`--skip-frac` drops the leading fraction of the trace (warm-up steps); `--steps` divides totals into per-step figures;
`--steps 0` takes the training steps from the trace itself: the launches between the first and the last Adam launch
(model set-up, the batch's host-to-device copies and whatever follows the last step are then not counted).
"""
import argparse
import csv
import re
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    m = re.match(r"k_gemm<(\d+), (\d+), (\d+), \d+, \d+, (\w+), (\w+), (\w+), (\w+)(?:, (\w+))?>", name)
    if m:
        bm, bn, bk, ta, tb, va, vb, x6 = m.groups()
        kind = "TN" if ta == "true" else ("NT" if tb == "true" else "NN")
        mode = {"1": "x6:", "true": "x6:", "2": "planes:", "3": "planesB:"}.get(x6, "")
        return (f"k_gemm<{mode}{bm}x{bn}x{bk},{kind}"
                f"{'' if va == 'true' and vb == 'true' else ',scalar'}>")
    return re.sub(r"\(.*$", "", name)[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--skip-frac", type=float, default=0.0)
    ap.add_argument("--md", action="store_true")
    ap.add_argument("--top", type=int, default=45)
    a = ap.parse_args()
    if a.csv.endswith(".db"):                                   # rocprofv3's default rocpd (SQLite) output
        import sqlite3
        cur = sqlite3.connect(a.csv).cursor()
        rows = [dict(Kernel_Name=n, Start_Timestamp=s, End_Timestamp=e, Grid_Size_X=gx, Grid_Size_Y=gy, Grid_Size_Z=gz,
                     Workgroup_Size_X=wx)
                for n, s, e, gx, gy, gz, wx in cur.execute(
                    "select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels")]
    else:
        rows = list(csv.DictReader(open(a.csv)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[int(len(rows) * a.skip_frac):]
    if a.steps == 0:
        adam = [i for i, r in enumerate(rows) if "k_adam" in r["Kernel_Name"]]
        if len(adam) < 2:
            raise SystemExit("--steps 0 needs at least two Adam launches in the trace")
        rows = rows[adam[0] + 1: adam[-1] + 1]
        a.steps = len(adam) - 1
    agg = defaultdict(lambda: [0, 0.0])
    for r in rows:
        grid = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        k = (short(r["Kernel_Name"]), grid)
        agg[k][0] += 1
        agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot = sum(v[1] for v in agg.values())
    items = sorted(agg.items(), key=lambda kv: -kv[1][1])
    sep = " | " if a.md else "  "
    if a.md:
        print("| kernel | grid (workgroups x,y,z) | launches/step | avg us | us/step | % |")
        print("|---|---|---|---|---|---|")
    for (name, grid), (n, us) in items[:a.top]:
        cells = [name, "x".join(map(str, grid)), f"{n / a.steps:.1f}", f"{us / n:.1f}", f"{us / a.steps:.0f}",
                 f"{100 * us / tot:.1f}"]
        print(("| " + " | ".join(cells) + " |") if a.md else
              f"{cells[0]:58s} {cells[1]:>14s} {cells[2]:>7s} {cells[3]:>8s} {cells[4]:>8s} {cells[5]:>5s}")
    print(f"{'| ' if a.md else ''}total kernel time per step: {tot / a.steps / 1e3:.3f} ms over {len(rows) / a.steps:.0f} launches")


if __name__ == "__main__":
    main()
