"""Shared helpers for the parity tests: golden loading and the 1e-4 relative metric."""
import json
import os

import numpy as np
import torch

from polyphemus_amd.graphs import BarGraphBatch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REL_TOL = 1e-4          # BASELINE.json north_star: "outputs matching reference within 1e-4 rel"


def rel_err(a, b):
    """max|a-b| / max|b|  (SURVEY §8(d)); 0 when both are all-zero."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    den = float(b.abs().max()) if b.numel() else 0.0
    num = float((a - b).abs().max()) if b.numel() else 0.0
    return num / den if den > 0 else num


def load_case(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    cfg = json.loads(str(z["cfg"]))
    return z, cfg


def batch_from_golden(z, cfg) -> BarGraphBatch:
    """The reference-built batch of a golden case as a BarGraphBatch (CPU)."""
    return BarGraphBatch(
        edge_index=torch.from_numpy(z["in/edge_index"].astype(np.int64)),
        edge_type=torch.from_numpy(z["in/etype"].astype(np.int32)),
        edge_dist=torch.from_numpy(z["in/edist"].astype(np.int32)),
        tokens=torch.from_numpy(z["in/tokens"].astype(np.int32)),
        s_tensor=torch.from_numpy(z["in/s_tensor"].astype(np.float32)),
        is_drum=torch.from_numpy(z["in/is_drum"].astype(bool)),
        bars=torch.from_numpy(z["in/bars"].astype(np.int64)),
        batch=torch.from_numpy(z["in/batch"].astype(np.int64)),
        num_nodes=int(z["in/num_nodes"]), n_bars=cfg["n_bars"])


def state_dict_from_golden(z, prefix="sd/"):
    return {k[len(prefix):]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith(prefix)}
