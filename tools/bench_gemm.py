#!/usr/bin/env python3
"""A/B timing (and error against float64) of the GEMM tile configurations, fp32-MFMA (0..3) and split mode (4..7), on the shapes of the training step
(interleaved rounds in ONE process, HIP events; guide rule 24).  Usage on the GPU box:
    python tools/bench_gemm.py [--d 256] [--nodes 16271]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphemus_amd import ops            # noqa: E402
from polyphemus_amd._lib import lib       # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--d", type=int, default=256)
ap.add_argument("--nodes", type=int, default=16271)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--cfgs", type=str, default="0,1,2,3,4,5,6,7")
args = ap.parse_args()
d, Nn = args.d, args.nodes
dev = "cuda"
L = lib()
shapes = [  # name, transA, transB, M, N, K
    ("gcl_fwd  NN", 0, 0, Nn, d, 4 * d), ("gcl_dA   NT", 0, 1, Nn, 4 * d, d), ("gcl_dW   TN", 1, 0, 4 * d, d, Nn),
    ("gcl7_fwd NN", 0, 0, Nn, d, 7 * d),
    ("chord_e  NT", 0, 1, Nn, d, 15 * d), ("chord_edX NN", 0, 0, Nn, 15 * d, d), ("chord_edW TN", 1, 0, d, 15 * d, Nn),
    ("chord_d  NT", 0, 1, Nn, 15 * d, d), ("chord_ddX NN", 0, 0, Nn, d, 15 * d), ("chord_ddW TN", 1, 0, 15 * d, d, Nn),
    ("dur_fwd  NT", 0, 1, Nn * 15, 99, d // 2), ("dur_dH   NN", 0, 0, Nn * 15, d // 2, 99),
    ("dur_dW   TN", 1, 0, 99, d // 2, Nn * 15), ("head     NT", 0, 1, 256, d, 2 * d),
]
cfgs = [int(c) for c in args.cfgs.split(",")]
print(f"{'shape':14s} {'M':>7s} {'N':>6s} {'K':>7s} auto | " + " | ".join(f"cfg{c} us  TF/s" for c in cfgs))
for name, ta, tb, M, N, K in shapes:
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.zeros(M, N, device=dev)
    best = {c: 1e9 for c in cfgs}
    ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
    err = {}
    for c in cfgs:
        L.pm_gemm_force_config(c)
        C.zero_()
        ops.gemm(A, B, C, M, N, K, A.stride(0), B.stride(0), N, transA=bool(ta), transB=bool(tb),
                 accum=bool(ta), split_k=0 if ta else 1)
        err[c] = float((C.double() - ref).abs().max() / ref.abs().max())
    del ref
    for rnd in range(args.rounds + 1):
        for c in cfgs:
            L.pm_gemm_force_config(c)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.gemm(A, B, C, M, N, K, A.stride(0), B.stride(0), N, transA=bool(ta), transB=bool(tb),
                         accum=bool(ta), split_k=0 if ta else 1)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                best[c] = min(best[c], e0.elapsed_time(e1) / 3 * 1e3)
    L.pm_gemm_force_config(-1)
    auto = L.pm_gemm_config(ta, M, N, K)
    fl = 2.0 * M * N * K
    print(f"{name:14s} {M:7d} {N:6d} {K:7d}  c{auto}  | " + " | ".join(f"{best[c]:8.1f} {fl / best[c] / 1e6:5.1f}" for c in cfgs))
    print(f"{'':14s} max |err| / max |ref| vs float64:     " + " | ".join(f"{err[c]:14.2e}" for c in cfgs))
