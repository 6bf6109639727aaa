#!/usr/bin/env python3
"""Where a step of the reference's loop shape goes when `model(graph)` is the drop-in module (bench.py's
`reference_loop_workload`): GPU time between events around each phase, and the host's wall clock for the whole step.

    python tools/dropin_breakdown.py [--active-slots-only] [--steps 10]
phases: inputs = one-hot -> ids conversion (+ its host read); forward = vae(graph); loss = the reference's `_losses` in torch
(seven .item() reads); backward = scaler.scale(loss).backward() (torch's loss backward + the native backward); optim =
scaler.step(torch Adam over 152 tensors) + update + zero_grad."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import reference_format, reference_losses  # noqa: E402
from polyphemus_amd.model import VAE  # noqa: E402
from polyphemus_amd.native import prepare_inputs  # noqa: E402
from polyphemus_amd.synthetic import synthetic_batch  # noqa: E402
from polyphemus_amd.trainer import HipTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--active-slots-only", action="store_true")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--profile", action="store_true", help="torch.profiler table of one step (GPU kernels by total time)")
    a = ap.parse_args()
    dev = torch.device("cuda")
    cfg = dict(dropout=0, batch_norm=True, gnn_n_layers=8, d=256, n_bars=2, resolution=8)
    torch.manual_seed(0)
    vae = VAE(**cfg, device=dev).to(dev)
    vae.train()
    vae.active_slots_only = a.active_slots_only
    opt = torch.optim.Adam(vae.parameters(), lr=5e-6, betas=(0.9, 0.98), eps=1e-9)
    scaler = torch.amp.GradScaler("cuda")
    b = synthetic_batch(256, 2, p=0.25, seed=1234).to(dev)
    g = reference_format(b)
    names = ["inputs", "forward", "loss", "backward", "optim"]
    acc = {k: 0.0 for k in names}
    wall = 0.0
    for it in range(a.steps + 3):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.__dict__.pop("_pm_inputs", None)
        ev[0].record()
        prepare_inputs(g)
        ev[1].record()
        with torch.autocast("cuda", dtype=torch.float16):
            (s_logits, c_logits), mu, lv = vae(g)
            ev[2].record()
            tot, parts = reference_losses(g.s_tensor, s_logits, g.c_tensor, c_logits, mu, lv)
        ev[3].record()
        scaler.scale(tot).backward()
        ev[4].record()
        scaler.step(opt)
        scaler.update()
        opt.zero_grad()
        ev[5].record()
        torch.cuda.synchronize()
        if it >= 3:
            wall += time.perf_counter() - t0
            for i, k in enumerate(names):
                acc[k] += ev[i].elapsed_time(ev[i + 1])
    if a.profile:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            g.__dict__.pop("_pm_inputs", None)
            with torch.autocast("cuda", dtype=torch.float16):
                (s_logits, c_logits), mu, lv = vae(g)
                tot, parts = reference_losses(g.s_tensor, s_logits, g.c_tensor, c_logits, mu, lv)
            scaler.scale(tot).backward()
            scaler.step(opt)
            scaler.update()
            opt.zero_grad()
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=70))
        w = "encoder.c_encoder.graph_encoder.layers.1.weight"
    n = a.steps
    print(f"reference loop shape, configs[1], decoder head slots {vae._native_step().info()['n_slots']}: wall {1e3 * wall / n:.3f} ms per step; "
          + ", ".join(f"{k} {acc[k] / n:.3f}" for k in names) + " ms (GPU time between events)")
    # the fused trainer on the same batch for comparison
    torch.manual_seed(0)
    vae2 = VAE(**cfg, device=dev).to(dev)
    vae2.train()
    tr = HipTrainer(vae2)
    for _ in range(3):
        tr.train_step(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.train_step(b)
    torch.cuda.synchronize()
    print(f"HipTrainer.train_step on the same batch: {1e3 * (time.perf_counter() - t0) / n:.3f} ms per step")


if __name__ == "__main__":
    main()
