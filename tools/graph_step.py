#!/usr/bin/env python3
"""Does a captured (hipGraph) training step beat the eagerly issued one?  (task 4.)

The native step is ~230 launches that the host issues in ~2.9 ms for ~5.0 ms of GPU work.  This tool captures one whole
step — `HipTrainer.train_step`: the four C calls on the caller's stream, the library's second stream between fork / join
events, the fused Adam — into ONE graph with torch.cuda.CUDAGraph (hipStreamBeginCapture underneath) and replays it on
the SAME batch (fixed shapes: the case most favourable to a graph; a real loader changes N and E every step, which would
need hipGraphExecUpdate of every kernel node's grid and arguments per step), then times eager against replay on the same box.

    python tools/graph_step.py [--d 256] [--batch 256] [--steps 30]

Prints one JSON line.  Run under `timeout`: a capture that goes wrong must not hold the box."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from polyphemus_amd.model import VAE
from polyphemus_amd.synthetic import synthetic_batch
from polyphemus_amd.trainer import HipTrainer


def timed(fn, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(steps):
        h0 = time.perf_counter()
        fn()
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, host / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    torch.manual_seed(0)
    vae = VAE(dropout=0, batch_norm=True, gnn_n_layers=8, d=a.d, n_bars=2, resolution=8, device="cuda").to("cuda")
    vae.train()
    tr = HipTrainer(vae, lr=5e-6)
    batch = synthetic_batch(a.batch, 2, p=0.25, seed=1234).to("cuda")
    eps = torch.randn(a.batch, a.d, device="cuda")             # fixed: a captured randn would need a graph-safe generator
    out = {"workload": f"B={a.batch} d={a.d} L=8", "steps": a.steps}
    for _ in range(5):
        tr.train_step(batch, eps)
    out["eager_ms"], out["eager_host_issue_ms"] = (round(v, 3) for v in timed(lambda: tr.train_step(batch, eps), a.steps))
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                           # capture needs a non-default stream
            for _ in range(3):
                tr.train_step(batch, eps)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            t0 = time.perf_counter()
            with torch.cuda.graph(g, stream=side):
                tr.train_step(batch, eps)
            out["capture_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        out["replay_ms"], out["replay_host_issue_ms"] = (round(v, 3) for v in timed(g.replay, a.steps))
        out["replay_over_eager"] = round(out["replay_ms"] / out["eager_ms"], 4)
        lossvec = tr.loss_buf.tolist()
        out["losses_after_replays_finite"] = all(v == v for v in lossvec)
    except Exception as e:                                      # a capture error is a result too
        out["capture_error"] = f"{type(e).__name__}: {str(e)[:300]}"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
