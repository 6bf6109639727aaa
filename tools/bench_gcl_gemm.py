"""Plain vs grouped (per track relation, row lists, stacked weights) timing of the three GCL contractions."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from polyphemus_amd import ops
Nn, d = 16271, 256
dd = d * d
dev = "cuda"
# realistic track assignment: within a bar (~64 nodes) tracks are contiguous runs
trel = torch.zeros(Nn, dtype=torch.int64)
i = 0
g = torch.Generator().manual_seed(0)
while i < Nn:
    for t in range(4):
        n = int(torch.randint(8, 24, (1,), generator=g))
        trel[i:i + n] = t
        i += n
        if i >= Nn: break
trel = trel.to(dev)
lists = torch.zeros((4, Nn), dtype=torch.int32, device=dev)
cnt = torch.zeros(8, dtype=torch.int32, device=dev)
for t in range(4):
    rows = torch.nonzero(trel == t).flatten().to(torch.int32)
    lists[t, :rows.numel()] = rows
    cnt[t] = rows.numel()
A = torch.randn(Nn, 4 * d, device=dev); W = torch.randn(7 * d, d, device=dev); bias = torch.randn(d, device=dev)
h = torch.empty(Nn, d, device=dev); dh = torch.randn(Nn, d, device=dev); dA = torch.empty(Nn, 4 * d, device=dev)
dW = torch.zeros(7 * d, d, device=dev)
grp = dict(rowmap=lists, rows_per_entry=1, dyn_entries=cnt, n_groups=4, map_group_stride=Nn, dyn_group_stride=1, partition=True)
def t(fn, reps=20):
    best = 1e9
    for r in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
print("fwd plain   ", t(lambda: ops.gemm(A, W, h, Nn, d, 4 * d, 4 * d, d, d, bias=bias)))
print("fwd grouped ", t(lambda: ops.gemm_desc(A, W, h, Nn, d, 4 * d, 4 * d, d, d, bias=bias, b_group_stride=dd, b_split_rows=d, b_shared_off=3 * dd, **grp)))
print("dA plain    ", t(lambda: ops.gemm(dh, W, dA, Nn, 4 * d, d, d, d, 4 * d, transB=True)))
print("dA grouped  ", t(lambda: ops.gemm_desc(dh, W, dA, Nn, 4 * d, d, d, d, 4 * d, transB=True, b_group_stride=dd, b_split_rows=d, b_shared_off=3 * dd, **grp)))
print("dW plain    ", t(lambda: ops.gemm(A, dh, dW, 4 * d, d, Nn, 4 * d, d, d, transA=True, accum=True, split_k=0)))
print("dW grouped  ", t(lambda: ops.gemm_desc(A, dh, dW, 4 * d, d, Nn, 4 * d, d, d, transA=True, accum=True, split_k=0, c_group_stride=dd, c_split_rows=d, c_shared_off=3 * dd, **grp)))
