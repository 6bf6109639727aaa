#!/usr/bin/env python3
"""What leaving training.json's defaults costs (GPU box): the native C++ step covers `batch_norm=True, dropout=0` (what
bench.py measures); the two other constructor switches of the reference (model.py:176-188,199) and synchronised
BatchNorm run the same kernels through the Python orchestration (polyphemus_amd/engine.py).  Times, at BASELINE
configs[1] (B = 256, d = 256, L = 8), the step of: the native path; the Python orchestration of the SAME configuration;
`dropout=0.1`; `batch_norm=False`.  One JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from polyphemus_amd.model import VAE
from polyphemus_amd.synthetic import synthetic_batch
from polyphemus_amd.trainer import HipTrainer


def run(cfg, native, steps=10, warm=3):
    torch.manual_seed(0)
    vae = VAE(**cfg, device="cuda").to("cuda")
    vae.train()
    tr = HipTrainer(vae, lr=5e-6, native=native)
    batch = synthetic_batch(256, 2, p=0.25, seed=1234).to("cuda")
    for _ in range(warm):
        tr.train_step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(batch)
    torch.cuda.synchronize()
    return {"native_step": bool(tr.native), "ms_per_step": round(1e3 * (time.perf_counter() - t0) / steps, 3)}


if __name__ == "__main__":
    base = dict(dropout=0, batch_norm=True, gnn_n_layers=8, d=256, n_bars=2, resolution=8)
    out = {"workload": "BASELINE configs[1]: B=256, 2 bars, d=256, L=8",
           "training_json_native": run(base, True),
           "training_json_python_orchestration": run(base, False),
           "dropout_0.1": run(dict(base, dropout=0.1), True),
           "batch_norm_false": run(dict(base, batch_norm=False), True)}
    print(json.dumps(out))
