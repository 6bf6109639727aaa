#!/usr/bin/env python3
"""Diagnostic (GPU box): per-tensor gradient error of the native HIP step and of the fp32 golden (the reference's own
fp32 run) against the fp64 oracle, on a golden case.  Justifies the gradient tolerances with measured numbers."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import vae_cpu
from polyphemus_amd.model import VAE
from polyphemus_amd.trainer import HipTrainer
from util import _as_dtype, batch_from_golden, load_case, state_dict_from_golden

torch.set_num_threads(1)
for case in sys.argv[1:] or ["lmd2_tiny", "nb3_tiny", "d128_l2"]:
    z, cfg = load_case(case)
    sd = state_dict_from_golden(z)
    names = [str(n) for n in z["param_names"]]
    cpu = batch_from_golden(z, cfg)
    eps = torch.from_numpy(z["in/eps"])
    vae = VAE(**cfg, device="cuda").to("cuda"); vae.load_state_dict(sd); vae.train(); vae.msg_dropout = 0.0
    tr = HipTrainer(vae)
    tr.train_step(cpu.to("cuda"), eps.cuda())
    P, _ = vae_cpu.split_state({k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}, names)
    opt = torch.optim.SGD([P[n] for n in names], lr=0.0)
    _, _, g64 = vae_cpu.train_step(_as_dtype(cpu, torch.float64), P, names, cfg, opt, eps.double(), msg_dropout=0.0)
    live = [n for n in names if g64[n] is not None]
    gmax = max(float(g64[n].abs().max()) for n in live)
    rows = []
    for n in live:
        h, g, o = tr._G[n].detach().cpu().double(), torch.from_numpy(z[f"train1/grad/{n}"]).double(), g64[n]
        den = max(float(o.abs().max()), 1e-2 * gmax)
        rows.append((float((h - g).abs().max()) / den, float((h - o).abs().max()) / den, float((g - o).abs().max()) / den, n))
    rows.sort(reverse=True)
    print(f"== {case}: gmax {gmax:.3e}; columns: hip-vs-golden32, hip-vs-o64, golden32-vs-o64")
    for r in rows[:12]:
        print(f"  {r[0]:.2e} {r[1]:.2e} {r[2]:.2e}  {r[3]}")
    print("  max hip-vs-o64 %.2e   max golden32-vs-o64 %.2e" % (max(r[1] for r in rows), max(r[2] for r in rows)))
