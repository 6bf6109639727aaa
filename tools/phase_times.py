#!/usr/bin/env python3
"""GPU time of the four native calls of a training step (forward + losses, decoder backward, encoder backward, its tail)
and of the rest (Adam, bookkeeping), from events on the step's stream; host time per step beside it.

    [PM_SIDE_STREAM=m] python tools/phase_times.py [--d 256] [--batch 256] [--steps 30]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import polyphemus_amd.trainer as T
from polyphemus_amd.model import VAE
from polyphemus_amd.synthetic import synthetic_batch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1234)
    a = ap.parse_args()
    torch.manual_seed(0)
    vae = VAE(dropout=0, batch_norm=True, gnn_n_layers=8, d=a.d, n_bars=2, resolution=8, device="cuda").to("cuda")
    vae.train(); vae.msg_dropout = 0.1
    tr = T.HipTrainer(vae, lr=5e-6)
    batch = synthetic_batch(a.batch, 2, p=0.25, seed=a.seed).to("cuda")
    marks = []
    orig = T.call

    def call(name, *args):
        if name.startswith("pm_vae_step_"):
            e0 = torch.cuda.Event(enable_timing=True); e0.record()
            orig(name, *args)
            e1 = torch.cuda.Event(enable_timing=True); e1.record()
            marks.append((name, e0, e1))
        else:
            orig(name, *args)
    T.call = call
    hi = torch.cuda.Stream(priority=-1) if os.environ.get("HI_PRIO") else None
    if hi is not None:
        torch.cuda.synchronize()
        torch.cuda.set_stream(hi)
    for _ in range(5):
        tr.train_step(batch)
    torch.cuda.synchronize()
    marks.clear()
    s0 = torch.cuda.Event(enable_timing=True); s0.record()
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        h0 = time.perf_counter()
        tr.train_step(batch)
        host += time.perf_counter() - h0
    s1 = torch.cuda.Event(enable_timing=True); s1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    acc = {}
    for name, e0, e1 in marks:
        acc[name] = acc.get(name, 0.0) + e0.elapsed_time(e1)
    tot = s0.elapsed_time(s1) / a.steps
    print(f"PM_SIDE_STREAM={os.environ.get('PM_SIDE_STREAM', 'default')}  step {tot:.3f} ms (wall {wall / a.steps * 1e3:.3f}), host issue time {host / a.steps * 1e3:.3f} ms/step")
    for k, v in acc.items():
        print(f"  {k[12:]:24s} {v / a.steps:.3f} ms")
    print(f"  {'rest':24s} {tot - sum(acc.values()) / a.steps:.3f} ms")


if __name__ == "__main__":
    main()
