"""Stand-alone timing of pm_unembed_dw at configs[1] (row lists of pm_unembed_row_lists): HIP events around 50 back-to-back calls.

    python tools/unembed_dw_bench.py       # on the GPU box; numbers in profiles/LOG.md, round 6
"""
import torch, sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from polyphemus_amd._lib import call, ptr, stream, lib
from polyphemus_amd import ops
from polyphemus_amd.synthetic import synthetic_batch
DEV = "cuda"
B, S, d = 256, 5, 256
cpu = synthetic_batch(B, 2, p=0.25, seed=1234)
b = cpu.to(DEV)
plan = ops.plan_build(b.edge_index, b.edge_type, b.edge_dist, b.bars, b.batch, b.is_drum, b.tokens, b.n_bars, b.s_tensor.shape[0], n_slots=S)
N, dh, R = cpu.num_nodes, d // 2, cpu.num_nodes * S
dl = torch.randn(N, S, 230, device=DEV); H = torch.randn(N, S, d, device=DEV)
lst = torch.empty(3, R, dtype=torch.int32, device=DEV)
cnt = torch.empty(int(lib().pm_unembed_row_counts_len(N, S)), dtype=torch.int32, device=DEV)
call("pm_unembed_row_lists", ptr(plan.tokens), ptr(plan.buf), N, plan.E, plan.G, d, S, ptr(lst), None, ptr(cnt), None, stream())
print("N", N, "R", R, "counts", cnt[:3].tolist())
g = [torch.zeros(v, dh, device=DEV) for v in (131, 131, 99)]
def f():
    call("pm_unembed_dw", ptr(dl), ptr(H), ptr(plan.buf), N, plan.E, plan.G, d, S, ptr(g[0]), ptr(g[1]), ptr(g[2]), ptr(lst), ptr(cnt), stream())
for _ in range(5): f()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): f()
e1.record(); torch.cuda.synchronize()
print(f"pm_unembed_dw stand-alone: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call")
