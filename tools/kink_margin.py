"""ReLU-kink and BatchNorm-conditioning audit of a small batch on the fp64 CPU oracle (test infrastructure: imports
oracle/).  For every ReLU of the step it prints the smallest |pre-activation| (absolute, and relative to the site's
RMS), and for every BatchNorm in training mode the smallest per-channel batch variance relative to the channel's mean
square — the two places where fp32 summation-order noise of the HIP step can be amplified into a percent-level
gradient change (a flipped ReLU mask; a near-constant channel divided by sqrt(var + eps)).

    python tools/kink_margin.py [--batch 8 --d 64 --layers 2 --seed 7 --model-seed 0]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def audit(batch=8, d=64, layers=2, seed=7, model_seed=0, n_bars=2, verbose=True):
    from oracle import vae_cpu
    from polyphemus_amd.model import VAE
    from polyphemus_amd.synthetic import synthetic_batch
    cfg = dict(dropout=0, batch_norm=True, gnn_n_layers=layers, d=d, n_bars=n_bars, resolution=8)
    torch.manual_seed(model_seed)
    vae = VAE(**cfg, device=torch.device("cpu"))
    sd = {k: v.detach().clone() for k, v in vae.state_dict().items()}
    names = [n for n, _ in vae.named_parameters()]
    eps = torch.randn(batch, d)
    cpu_batch = synthetic_batch(batch, n_bars, p=0.25, seed=seed)
    P, names = vae_cpu.split_state({k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}, names)
    b64 = cpu_batch.to("cpu")
    b64.__dict__["_c_tensor"], b64.__dict__["_edge_attrs"] = cpu_batch.c_tensor.double(), cpu_batch.edge_attrs.double()
    b64.s_tensor = cpu_batch.s_tensor.double()

    relus, bns = [], []
    real_relu, real_bn = F.relu, F.batch_norm

    def relu(x, *a, **k):
        if x.numel() and x.dim() >= 1:
            v = x.detach()
            rms = float((v ** 2).mean().sqrt())
            a_ = v.abs()
            nz = a_[a_ > 0]
            relus.append((tuple(v.shape), float(nz.min()) if nz.numel() else float("inf"), rms,
                          int((a_ < 1e-5 * max(rms, 1e-30)).sum()), int((a_ == 0).sum())))
        return real_relu(x, *a, **k)

    def batch_norm(x, rm, rv, w, b, training, mom, eps_):
        if training and x.numel():
            v = x.detach()
            dims = [i for i in range(v.dim()) if i != 1]
            var = v.var(dims, unbiased=False)
            ms = (v ** 2).mean(dims)
            bns.append((tuple(v.shape), float(var.min()), float((var / ms.clamp(min=1e-300)).min()),
                        int((var < 1e-8).sum())))
        return real_bn(x, rm, rv, w, b, training, mom, eps_)

    F.relu, F.batch_norm = relu, batch_norm
    try:
        vae_cpu.vae_forward(b64, P, cfg, True, eps.double(), msg_dropout=0.0)
    finally:
        F.relu, F.batch_norm = real_relu, real_bn
    worst_rel = min((m / max(r, 1e-300) for _, m, r, _, _ in relus if r > 0), default=float("inf"))
    if verbose:
        print(f"batch {batch} d {d} L {layers} seed {seed}: {len(relus)} ReLU sites, {len(bns)} BatchNorm sites")
        print("ReLU sites: shape, min nonzero |pre|, rms, #(|pre| < 1e-5 rms), #exact zeros")
        for i, r in enumerate(relus):
            print(f"  relu[{i:3d}] {str(r[0]):>18s} min {r[1]:.3e} rms {r[2]:.3e} rel {r[1] / max(r[2], 1e-300):.3e} near {r[3]} zero {r[4]}")
        print("BatchNorm sites: shape, min var, min var/mean-square, #(var < 1e-8)")
        for i, b in enumerate(bns):
            print(f"  bn[{i:3d}] {str(b[0]):>18s} minvar {b[1]:.3e} minrel {b[2]:.3e} tiny {b[3]}")
        print(f"smallest relative pre-ReLU margin {worst_rel:.3e}")
    return dict(relu=relus, bn=bns, worst_rel=worst_rel)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--model-seed", type=int, default=0)
    a = ap.parse_args()
    audit(a.batch, a.d, a.layers, a.seed, a.model_seed)
