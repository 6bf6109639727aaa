#!/usr/bin/env python3
"""Where and when every workgroup of one k_gcl_fwd launch ran: realtime clocks at its start and end, its XCC id and what
it worked on (blocks of K x 100 + rows), from a -DGCL_BLOCKLOG build of gcl.hip:

    python tools/build_variants.py gcl.hip log=-DGCL_BLOCKLOG
    SEED=1236 PM_LIB_PATH=polyphemus_amd/variants/libpm_log.so python tools/gcl_blocklog.py

This is how the tile schedule of csrc/tile_order.h was checked against the hardware: workgroup b runs on XCD b % 8; an
XCD hands its workgroups in order to its four shader engines in turn, and one that has to wait holds back those behind it."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from polyphemus_amd import ops
from polyphemus_amd._lib import lib
from polyphemus_amd.synthetic import synthetic_batch
seed = int(os.environ.get("SEED", 1235)); d = 256
cpu = synthetic_batch(int(os.environ.get("B", 256)), int(os.environ.get("NB", 2)), p=0.25, seed=seed); b = cpu.to("cuda")
plan = ops.plan_build(b.edge_index, b.edge_type, b.edge_dist, b.bars, b.batch, b.is_drum, b.tokens, b.n_bars, b.s_tensor.shape[0])
N = cpu.num_nodes
torch.manual_seed(0)
x = torch.randn(N, d, device="cuda")
T = ops.edge_table(torch.randn(d, 32, device="cuda") * 0.5, torch.randn(d, device="cuda") * 0.1)
W = torch.randn(7 * d, d, device="cuda") / d ** 0.5
bias = torch.randn(d, device="cuda")
Wf = ops.split_planes_frag(W, 1)
P = torch.zeros(3, N * 4 * d, dtype=torch.int16, device="cuda")
s = torch.zeros(8, 2, d, dtype=torch.float64, device="cuda")
for _ in range(5): ops.gcl_forward_fused(x, T, plan, 0.1, 5, 2, Wf, bias, col_stats=s, planes=P)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 4096)()
lib().pm_debug_read_blocklog(buf)
rows = [(i, buf[4*i], buf[4*i+1], buf[4*i+2], buf[4*i+3] % 100000) for i in range(1024) if buf[4*i] and buf[4*i+3] >= 0]
sched = [buf[4*i+3] // 100000 for i in range(1024) if buf[4*i] and buf[4*i+3] >= 0]
print("ticks (10 ns) between workgroup start and its tile being known: min / median / max", min(sched), sorted(sched)[len(sched)//2], max(sched))
t0 = min(r[1] for r in rows)
print("seed", seed, "live blocks", len(rows), "span us", (max(r[2] for r in rows) - t0) / 100.0)
for x8 in range(8):
    rs = [r for r in rows if r[0] % 8 == x8]
    xcc = sorted(set(r[3] for r in rs))
    late = [(r[0], round((r[1]-t0)/100,1), round((r[2]-t0)/100,1), r[4]) for r in rs if (r[1]-t0) > 300]
    print(" b%8 =", x8, "n", len(rs), "xcc ids", xcc, "end max", max((r[2]-t0)/100 for r in rs), "late starters", late)
import collections
byk = collections.defaultdict(list)
for r in rows: byk[r[4]].append((r[2]-r[1])/100.0)
for k in sorted(byk): print("  kind (nblk*100+rows)", k, "n", len(byk[k]), "dur min/avg/max", round(min(byk[k]),1), round(sum(byk[k])/len(byk[k]),1), round(max(byk[k]),1))
print("XCD 0 blocks (k, kind, start, end):")
rs = sorted([r for r in rows if r[0] % 8 == 0], key=lambda r: r[0])
print([(r[0] // 8, r[4], round((r[1]-t0)/100,1), round((r[2]-t0)/100,1)) for r in rs[int(os.environ.get("FROM", 20)):]])
