#!/usr/bin/env python3
"""Run ONE GEMM shape a few times (for rocprofv3 --pmc runs).  usage: one_gemm.py ta tb M N K cfg reps"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphemus_amd import ops
from polyphemus_amd._lib import lib
ta, tb, M, N, K, cfg, reps = [int(x) for x in sys.argv[1:8]]
A = torch.randn((K, M) if ta else (M, K), device="cuda")
B = torch.randn((N, K) if tb else (K, N), device="cuda")
C = torch.zeros(M, N, device="cuda")
lib().pm_gemm_force_config(cfg)
for _ in range(reps):
    ops.gemm(A, B, C, M, N, K, A.stride(0), B.stride(0), N, transA=bool(ta), transB=bool(tb), accum=bool(ta), split_k=0 if ta else 1)
torch.cuda.synchronize()
