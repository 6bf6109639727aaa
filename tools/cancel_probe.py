#!/usr/bin/env python3
"""Probe (GPU box): accuracy of the weight-gradient contraction dW = A^T dh on data whose sum CANCELS (dh with zero
column means, A with a positive offset: the shape of a `root` gradient behind a BatchNorm), three ways: the planes kernel
(six bf16 MFMA products per fp32 product), the fp32-MFMA tile kernel, and torch fp32 — each against fp64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from polyphemus_amd import ops

def main():
    torch.manual_seed(0)
    dev = "cuda"
    for N, d, off in ((16384, 256, 0.0), (16384, 256, 1.0), (16384, 256, 4.0), (4096, 256, 4.0)):
        A = torch.randn(N, d, device=dev).abs() * 0.5 + off
        dh = torch.randn(N, d, device=dev)
        dh -= dh.mean(0, keepdim=True)
        want = A.double().T @ dh.double()
        scale = float(want.abs().max())
        # fp32 MFMA tile kernel (TN)
        C32 = torch.zeros(d, d, device=dev)
        ops.gemm(A, dh, C32, d, d, N, d, d, d, transA=True, accum=True, split_k=0)
        # planes kernel: ungrouped TN through gemm_desc with operand planes
        Ap, dhp = ops.split_planes(A), ops.split_planes(dh)
        Cp = torch.zeros(d, d, device=dev)
        ops.gemm_desc(Ap, dhp, Cp, d, d, N, d, d, d, transA=True, accum=True, split_k=0, a_plane_stride=A.numel(),
                      b_plane_stride=dh.numel(), planes=True)
        Ct = A.T @ dh
        def err(x): return float((x.double() - want).abs().max()) / scale
        terms = float((A.double().abs().T @ dh.double().abs()).max())
        print(f"N={N} offset={off}: |result| {scale:.3e}  sum|terms| {terms:.3e}  cancellation {terms/scale:.0f}x   "
              f"err/scale: planes {err(Cp):.2e}  fp32-mfma {err(C32):.2e}  torch-fp32 {err(Ct):.2e}", flush=True)

if __name__ == "__main__":
    main()
