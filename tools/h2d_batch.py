#!/usr/bin/env python3
"""Host->device cost of one batch of the default workload (the bench keeps batches HBM-resident; this is the
PCIe-inclusive correction quoted in DESIGN.md).  Usage on the GPU box: python tools/h2d_batch.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphemus_amd.synthetic import synthetic_batch   # noqa: E402

b = synthetic_batch(256, 2, p=0.25, seed=0)
nbytes = sum(v.numel() * v.element_size() for v in b.__dict__.values() if torch.is_tensor(v))
for pin in (False, True):
    src = b
    if pin:
        src = type(b)(**{k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in b.__dict__.items()})
    for _ in range(3):
        src.to("cuda", non_blocking=pin)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        src.to("cuda", non_blocking=pin)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"batch of 256 samples: {nbytes / 1e6:.2f} MB in {len([1 for v in b.__dict__.values() if torch.is_tensor(v)])} tensors, "
          f"{'pinned' if pin else 'pageable'} host memory: {dt * 1e3:.3f} ms per batch ({nbytes / dt / 1e9:.1f} GB/s)")

# ---- the device-collate path (polyphemus_amd/data.py): token grids + structures in, batch of bar graphs built on the GPU
import numpy as np                                                    # noqa: E402
from polyphemus_amd.data import _staging, collate_on_device, relayout_sample   # noqa: E402
from polyphemus_amd.synthetic import disk_sample                      # noqa: E402

rng = np.random.default_rng(0)
samples = [relayout_sample(*disk_sample(rng, 2, 0.25), 2) for _ in range(256)]
stage = _staging(256, 2, pin=True)
for _ in range(3):
    collate_on_device(samples, 2, "cuda", stage)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    collate_on_device(samples, 2, "cuda", stage)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
nb = stage[0].numel() * 2 + stage[1].numel()
print(f"device collate of 256 samples: {nb / 1e6:.2f} MB staged, {dt * 1e3:.3f} ms per batch "
      f"(host stacking + H2D + graph kernels + token gather)")
