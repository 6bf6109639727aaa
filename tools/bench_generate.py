#!/usr/bin/env python3
"""Time the generation helpers (csrc/generate.hip) at BASELINE configs[1] size against the reference's torch op
sequence run on the same GPU (utils.py:59-79, model.py:609-623).  Usage: python tools/bench_generate.py [B] [n_bars]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphemus_amd import ops  # noqa: E402


def torch_mtp(c_logits, s_tensor):                    # the reference's op sequence, on the device
    mtp = torch.zeros((*s_tensor.shape, 15, 230), device=c_logits.device).reshape(-1, 15, 230)
    silence = torch.zeros(15, 230, device=c_logits.device)
    silence[0, 129] = 1.0
    silence[1:, 130] = 1.0
    on = s_tensor.bool().reshape(-1)
    mtp[on] = c_logits
    mtp[~on] = silence
    return mtp.reshape(*s_tensor.shape, 15, 230)


def torch_binary(s_logits):
    s = torch.sigmoid(s_logits)
    s[s >= 0.5] = 1
    s[s < 0.5] = 0
    s = s.bool()
    idx = torch.nonzero(~s.any(dim=-1).any(dim=-1), as_tuple=True)
    s[idx + (0, 0)] = True
    return s


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    s = (torch.rand(B, nb, 4, 32, device="cuda") < 0.25)
    s[:, :, 0, 0] |= ~s.any(-1).any(-1)
    N = int(s.sum())
    c = torch.randn(N, 15, 230, device="cuda")
    sf = s.float()
    byts = (s.numel() + N) * 15 * 230 * 4
    t = timeit(lambda: ops.mtp_from_logits(c, sf, check=False))
    t_ref = timeit(lambda: torch_mtp(c, s))
    print(f"mtp_from_logits  B={B} nb={nb} N={N}: {t:8.1f} us  {byts / t / 1e6:7.2f} TB/s algorithmic "
          f"(write {s.numel() * 13800 / 1e6:.0f} MB + read {N * 13800 / 1e6:.0f} MB); torch op sequence {t_ref:8.1f} us")
    x = torch.randn(B, nb, 4, 32, device="cuda")
    t = timeit(lambda: ops.binary_from_logits(x))
    t_ref = timeit(lambda: torch_binary(x))
    print(f"binary_from_logits G={B * nb}: {t:8.1f} us; torch op sequence {t_ref:8.1f} us")


if __name__ == "__main__":
    main()
