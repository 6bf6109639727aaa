"""Race or ReLU kink?  Repeats the smoke step (defaults = `__graft_entry__.SMOKE`; round 3's red smoke was --d 64 --seed 7) in FRESH
processes on the GPU under several PM_* settings and attributes every gradient to the fp64 oracle's near-kink ReLU
decisions (oracle/kinks.py).  Test infrastructure: imports oracle/.

    python tools/smoke_repeat.py --n 100 --inproc 4 --settings default side0 [det] --out gpurun_out/r04_smoke_repeat.json

Parent process: never touches the GPU; computes the oracle once, spawns the children one at a time, aggregates.
Child (`--child`): builds the model exactly as smoke() does, runs `inproc` + 1 independent first steps (fresh model and
trainer each; the first one is the process's first GPU work) and stores the flat gradients."""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SETTINGS = {"default": {}, "side0": {"PM_SIDE_STREAM": "0"}, "det": {"PM_DETERMINISTIC": "1"},
            "det_side0": {"PM_DETERMINISTIC": "1", "PM_SIDE_STREAM": "0"}}


def case_cfg(a):
    return dict(dropout=0, batch_norm=True, gnn_n_layers=a.layers, d=a.d, n_bars=2, resolution=8)


def child(a):
    import numpy as np
    import torch
    from polyphemus_amd.model import VAE
    from polyphemus_amd.synthetic import synthetic_batch
    from polyphemus_amd.trainer import HipTrainer
    dev = torch.device("cuda", 0)
    cfg = case_cfg(a)
    cpu_batch = synthetic_batch(a.batch, 2, p=0.25, seed=a.seed)
    grads, losses = [], []
    used = None
    if a.case and os.path.exists(a.case):
        used = torch.load(a.case, weights_only=False)["used"]
    for rep in range(a.inproc + 1):
        torch.manual_seed(0)
        vae = VAE(**cfg, device=dev).to(dev)
        vae.train()
        vae.msg_dropout = 0.0
        eps = torch.randn(a.batch, cfg["d"])
        if a.probe and rep == 0:
            sd = {k: v.detach().cpu().clone() for k, v in vae.state_dict().items()}
            names = [n for n, _ in vae.named_parameters()]
            torch.save({"sd": sd, "names": names, "eps": eps}, a.probe)
        trainer = HipTrainer(vae)
        out = trainer.train_step(cpu_batch.to(dev), eps.to(dev))
        losses.append(out.tolist())
        names = used if used is not None else [n for n, _ in vae.named_parameters()]
        grads.append(torch.cat([trainer._G[n].detach().reshape(-1) for n in names]).cpu().numpy())
        del trainer, vae
    np.savez(a.result, grads=np.stack(grads), losses=np.array(losses))


def parent(a):
    import numpy as np
    import torch
    from oracle import kinks
    from polyphemus_amd.synthetic import synthetic_batch
    tmp = a.tmp
    os.makedirs(tmp, exist_ok=True)
    base = [sys.executable, os.path.abspath(__file__), "--child", "--batch", str(a.batch), "--d", str(a.d), "--layers",
            str(a.layers), "--seed", str(a.seed)]
    probe = os.path.join(tmp, "probe.pt")
    subprocess.run(base + ["--probe", probe, "--inproc", "0", "--result", os.path.join(tmp, "probe.npz")], check=True)
    case = torch.load(probe, weights_only=False)
    cfg = case_cfg(a)
    cpu_batch = synthetic_batch(a.batch, 2, p=0.25, seed=a.seed)
    t0 = time.time()
    ref = kinks.kink_gradients(cpu_batch, case["sd"], case["names"], cfg, case["eps"], tau=a.tau)
    den = float(ref["f0"].norm())
    case_path = os.path.join(tmp, "case.pt")
    torch.save({"used": ref["used"]}, case_path)
    report = {"case": {"batch": a.batch, "d": a.d, "layers": a.layers, "seed": a.seed, "tau": a.tau,
                       "oracle_s": round(time.time() - t0, 1), "n_grad_elements": int(ref["f0"].numel())},
              "near_kinks": [{"site": s, "index": j, "abs_pre": ab, "rel_margin": ab / r,
                              "single_flip_rel_l2": float(ref["deltas"][k].norm()) / den}
                             for k, (s, j, ab, r) in enumerate(ref["kinks"])],
              "smallest_rel_margin_per_site": [m for _, m in ref["margins"]], "settings": {}}
    for name in a.settings:
        env = dict(os.environ)
        env.update(SETTINGS[name])
        rows = []
        hashes = {}
        t0 = time.time()
        for i in range(a.n):
            res = os.path.join(tmp, f"run_{name}_{i}.npz")
            r = subprocess.run(base + ["--case", case_path, "--inproc", str(a.inproc), "--result", res], env=env,
                               capture_output=True, text=True)
            if r.returncode != 0:
                rows.append({"run": i, "error": r.stderr[-400:]})
                continue
            z = np.load(res)
            for rep, g in enumerate(z["grads"]):
                h = hashlib.sha256(g.tobytes()).hexdigest()[:16]
                hashes[h] = hashes.get(h, 0) + 1
                ex = kinks.explain(torch.from_numpy(g).double(), ref)
                rows.append({"run": i, "rep": rep, "raw": ex["raw"], "residual": ex["residual"], "ok": ex["ok"],
                             "flips": [(s, j) for s, j, _ in ex["flips"]], "sha": h,
                             "losses": [float(v) for v in z["losses"][rep]]})
            os.remove(res)
        ok = [r for r in rows if "raw" in r]
        raws = np.array([r["raw"] for r in ok])
        edges = [0, 1e-6, 3e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1e9]
        hist = {f"<{edges[k + 1]:g}": int(((raws >= edges[k]) & (raws < edges[k + 1])).sum()) for k in range(len(edges) - 1)}
        first = np.array([r["raw"] for r in ok if r["rep"] == 0])
        flipsets = {}
        for r in ok:
            key = ";".join(f"{s}:{j}" for s, j in r["flips"]) or "none"
            flipsets[key] = flipsets.get(key, 0) + 1
        report["settings"][name] = {
            "env": SETTINGS[name], "processes": a.n, "steps": len(ok), "errors": len(rows) - len(ok),
            "wall_s": round(time.time() - t0, 1),
            "raw_rel_l2_hist": hist, "raw_max": float(raws.max()) if len(raws) else None,
            "first_step_of_process_over_1e-4": int((first > 1e-4).sum()), "all_steps_over_1e-4": int((raws > 1e-4).sum()),
            "residual_after_kink_fit_max": max((r["residual"] for r in ok), default=None),
            "all_explained_by_binary_kink_decisions": all(r["ok"] for r in ok), "flip_sets": flipsets,
            "distinct_gradient_bit_patterns": len(hashes),
            "loss_bit_patterns": len({tuple(r["losses"]) for r in ok}),
            "bad_examples": [r for r in ok if r["raw"] > 1e-4][:6]}
        print(name, json.dumps(report["settings"][name])[:1500], flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--probe", default="")
    ap.add_argument("--case", default="")
    ap.add_argument("--result", default="")
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--inproc", type=int, default=4)
    import __graft_entry__ as _ge
    ap.add_argument("--batch", type=int, default=_ge.SMOKE["batch"])
    ap.add_argument("--d", type=int, default=_ge.SMOKE["d"])
    ap.add_argument("--layers", type=int, default=_ge.SMOKE["layers"])
    ap.add_argument("--seed", type=int, default=_ge.SMOKE["batch_seed"])
    ap.add_argument("--tau", type=float, default=2e-5)
    ap.add_argument("--settings", nargs="+", default=["default", "side0"])
    ap.add_argument("--tmp", default="/tmp/pm_smoke_repeat")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r04_smoke_repeat.json"))
    a = ap.parse_args()
    (child if a.child else parent)(a)
