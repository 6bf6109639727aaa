#!/usr/bin/env python3
"""Which kernel set carries the full-size gradient error (GPU box): the native HIP step at one BASELINE configuration
under each A/B switch of the step, R repetitions each (same inputs: repetitions differ only in the order of the float
atomics), against ONE fp64 run of oracle/vae_cpu.py.

    python tools/parity_attribution.py [--config configs1_lmd2_b256_d256] [--reps 5] > gpurun_out/.../attribution.json

The switches are read once per process, so every setting runs in a child process; the oracle's gradients are computed
once by the parent and handed over through a file in TMPDIR.  One JSON line per setting: worst tensor error / relative
L2 of every repetition, the per-tensor error of the tensors that were worst anywhere, and the spread between
repetitions (max |g_rep - g_rep0| in the same per-tensor metric).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SETTINGS = [
    ("default", {}),
    ("PM_GCL_FUSED=0", {"PM_GCL_FUSED": "0"}),
    ("PM_GCL_NO_DW=1", {"PM_GCL_NO_DW": "1"}),
    ("PM_NO_ROWS_W=1", {"PM_NO_ROWS_W": "1"}),
    ("PM_EMBED_MFMA=0", {"PM_EMBED_MFMA": "0"}),
    ("PM_GCL_NO_CLASSES=1", {"PM_GCL_NO_CLASSES": "1"}),
    ("PM_GCL_PLANES=0", {"PM_GCL_PLANES": "0"}),
    ("PM_GCL_DW_SPLIT=8", {"PM_GCL_DW_SPLIT": "8"}),        # d = 512 only: K slices of the weight-gradient kernel (default 2)
]
WATCH = ["decoder.c_decoder.bars_decoder.weight", "encoder.c_encoder.graph_encoder.layers.4.root",
         "encoder.c_encoder.chord_encoder.weight", "decoder.lin_decoder.weight", "encoder.c_encoder.graph_encoder.layers.1.root"]


def child(config, reps, oracle_file):
    import torch
    from util import FULLSIZE, grad_errors, hip_fullsize_step
    spec = FULLSIZE[config]
    g64 = torch.load(oracle_file)
    out = {"reps": [], "watch": {w: [] for w in WATCH}, "spread": []}
    first = None
    for r in range(reps):
        run = hip_fullsize_step(spec)
        ge = grad_errors(run["names"], run["grads"], g64)
        out["reps"].append({"worst_tensor_err": ge["hip_vs_o64"]["worst_tensor_err"], "worst_tensor": ge["hip_vs_o64"]["worst_tensor"],
                            "rel_l2": ge["hip_vs_o64"]["rel_l2"]})
        for w in WATCH:
            out["watch"][w].append(round(ge["per_tensor"].get(w, float("nan")), 6))
        if first is None:
            first = run["grads"]
            out["info"] = run["info"]
        else:
            sp = grad_errors(run["names"], run["grads"], {n: (first[n] if g64[n] is not None else None) for n in run["names"]})
            out["spread"].append({"worst_tensor_err": sp["hip_vs_o64"]["worst_tensor_err"], "worst_tensor": sp["hip_vs_o64"]["worst_tensor"],
                                  "rel_l2": sp["hip_vs_o64"]["rel_l2"]})
    print("ATTR " + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="configs1_lmd2_b256_d256")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--child", default="")
    ap.add_argument("--only", default="")
    ap.add_argument("--seeds", default="", help="comma-separated batch seeds: also report the default kernels on other realisations")
    ap.add_argument("--seeds-only", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a.config, a.reps, a.child)
    import torch
    from bench import host_cores
    from util import FULLSIZE, grad_errors, hip_fullsize_step, oracle_fullsize
    torch.set_num_threads(host_cores())
    spec = FULLSIZE[a.config]
    run = hip_fullsize_step(spec)
    res, times = oracle_fullsize(spec, run)
    g64, g32 = res["o64"][2], res["o32"][2]
    ge32 = grad_errors(run["names"], g32, g64)                       # the fp32 reference arithmetic against fp64
    ref = ge32["hip_vs_o64"]
    ref["watch"] = {w: round(ge32["per_tensor"].get(w, float("nan")), 6) for w in WATCH}
    ref["worst_tensors"] = ge32["hip_worst_tensors"][:5]
    print(json.dumps({"config": a.config, "oracle_seconds": times, "o32_vs_o64": ref}), flush=True)
    if a.seeds:
        # how much of a tensor's error is the realisation (batch, dropout mask): the same step on other synthetic batches
        import copy
        for sd in [int(x) for x in a.seeds.split(",")]:
            for mp in (spec["msg_p"], 0.0):
                sp = copy.deepcopy(spec)
                sp["seed"], sp["msg_p"] = sd, mp
                r2 = hip_fullsize_step(sp)
                res2, _ = oracle_fullsize(sp, r2)
                gh = grad_errors(r2["names"], r2["grads"], res2["o64"][2], res2["o32"][2])
                print(json.dumps({"realisation": {"batch_seed": sd, "msg_p": mp},
                                  "hip_vs_o64": gh["hip_vs_o64"], "o32_vs_o64": gh["o32_vs_o64"],
                                  "hip_worst_tensors": gh["hip_worst_tensors"][:4],
                                  "watch_hip": {w: round(gh["per_tensor"].get(w, float("nan")), 6) for w in WATCH}}), flush=True)
        if a.seeds_only:
            return
    f = os.path.join(tempfile.gettempdir(), f"pm_attr_{os.getpid()}.pt")
    torch.save(g64, f)
    try:
        for name, env in SETTINGS:
            if a.only and a.only not in name:
                continue
            e = dict(os.environ)
            e.update(env)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", a.config, "--reps", str(a.reps), "--child", f],
                               env=e, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("ATTR ")]
            if r.returncode != 0 or not line:
                print(json.dumps({"setting": name, "error": r.stderr[-2000:]}), flush=True)
                continue
            print(json.dumps({"setting": name, **json.loads(line[0][5:])}), flush=True)
    finally:
        os.remove(f)


if __name__ == "__main__":
    main()
