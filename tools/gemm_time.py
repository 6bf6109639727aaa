#!/usr/bin/env python3
"""Development helper: time a few GEMM shapes per tile configuration (pm_gemm_force_config) in isolation.
The shapes are those of composing the chord decoder with the un-embeddings: logits = x @ Mcat^T [16 k x 1150 x 256],
its input gradient and the weight-gradient product — priced before building anything: 90-92 us for the forward product alone
against 62.7 us of the k_rows_w launch it would replace, so the composition is not worth its plumbing with these kernels."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from polyphemus_amd import ops
from polyphemus_amd._lib import lib
def run(ta, tb, M, N, K, cfg, ldc=None, reps=20):
    A = torch.randn((K, M) if ta else (M, K), device="cuda")
    B = torch.randn((N, K) if tb else (K, N), device="cuda")
    ldc = ldc or N
    C = torch.zeros(M, ldc, device="cuda")
    lib().pm_gemm_force_config(cfg)
    for _ in range(3):
        ops.gemm(A, B, C, M, N, K, A.stride(0), B.stride(0), ldc, transA=bool(ta), transB=bool(tb), accum=bool(ta), split_k=0 if ta else 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(A, B, C, M, N, K, A.stride(0), B.stride(0), ldc, transA=bool(ta), transB=bool(tb), accum=bool(ta), split_k=0 if ta else 1)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"ta{ta} tb{tb} M{M} N{N} K{K} cfg{cfg} ldc{ldc}: {us:7.1f} us  {2.0*M*N*K/us/1e6:6.1f} TFLOP/s")
for cfg in (0, 4, 7, -1):
    run(0, 1, 16271, 1150, 256, cfg)
run(0, 1, 16271, 1152, 256, 4)
run(0, 1, 16271, 1280, 256, 4)
# backward candidates: dx = dlogits @ Mcat (NN, K = 1150 / 1152)
for cfg in (0, 4, 7):
    run(0, 0, 16271, 256, 1152, cfg)
run(0, 0, 16271, 256, 1150, 0)
# Q = dlogits^T x (TN)
for cfg in (2, 5):
    run(1, 0, 1152, 256, 16271, cfg)
