#!/usr/bin/env python3
"""Development micro-benchmark of the three GCL kernels of csrc/gcl.hip against the kernels they replace, on the bench
batch (B = 256, d = 256).  WHICH selects what runs (u unfused forward pair, f fused forward, p fused forward without the A'
output, n input gradient, w weight gradient); PM_LIB_PATH selects a library variant (tools/build_variants.py).  (bf16-triple
entry points: the fp16 pair format is measured through the step — tools/ab_round.sh with PM_H2=0 / 1.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from polyphemus_amd import ops
from polyphemus_amd._lib import call, ptr, stream
from polyphemus_amd.synthetic import synthetic_batch

def main():
    B = int(os.environ.get("B", 256)); d = int(os.environ.get("D", 256)); p = float(os.environ.get("P", 0.1))
    cpu = synthetic_batch(B, 2, p=0.25, seed=int(os.environ.get("SEED", 1234)))
    b = cpu.to("cuda")
    plan = ops.plan_build(b.edge_index, b.edge_type, b.edge_dist, b.bars, b.batch, b.is_drum, b.tokens, b.n_bars, b.s_tensor.shape[0])
    N, dd = cpu.num_nodes, d * d
    torch.manual_seed(0)
    x = torch.randn(N, d, device="cuda")
    T = ops.edge_table(torch.randn(d, 32, device="cuda") * 0.5, torch.randn(d, device="cuda") * 0.1)
    W = torch.randn(7 * d, d, device="cuda") / d ** 0.5
    bias = torch.randn(d, device="cuda")
    Wp, Wf = ops.split_planes(W), ops.split_planes_frag(W, 1)
    tl, tc = plan.field("trk_list"), plan.field("trk_cnt")
    P = torch.zeros(3, N * 4 * d, dtype=torch.int16, device="cuda")
    s = torch.zeros(8, 2, d, dtype=torch.float64, device="cuda")
    h = torch.zeros(N, d, device="cuda")
    def unfused():
        call("pm_segreduce_fwd_planes", ptr(x), ptr(T), ptr(plan.buf), N, plan.E, plan.G, d, p, 5, 2, 1, ptr(P), N * 4 * d, stream())
        ops.gemm_desc(P, Wp, h, N, d, 4 * d, 4 * d, d, d, bias=bias, b_group_stride=dd, b_split_rows=d, b_shared_off=3 * dd,
                      a_plane_stride=N * 4 * d, b_plane_stride=W.numel(), rowmap=tl, rows_per_entry=1, dyn_entries=tc,
                      n_groups=4, map_group_stride=N, dyn_group_stride=1, partition=True, planes=True, class_ptr=tc[8:],
                      class_block=d, b_frag=Wf, col_stats=s)
    def fused(): ops.gcl_forward_fused(x, T, plan, p, 5, 2, Wf, bias, col_stats=s, planes=P)
    def fused_np(): ops.gcl_forward_fused(x, T, plan, p, 5, 2, Wf, bias, col_stats=s)
    def timeit(fn, n=50):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return a.elapsed_time(e) / n * 1e3
    dh = torch.randn(N, d, device="cuda")
    dhp, Wft = ops.split_planes(dh), ops.split_planes_frag(W, 0)
    dA = torch.empty(N, 4 * d, device="cuda")
    def nt_unfused():
        ops.gemm_desc(dhp, Wp, dA, N, 4 * d, d, d, d, 4 * d, transB=True, b_group_stride=dd, b_split_rows=d,
                      b_shared_off=3 * dd, a_plane_stride=dh.numel(), b_plane_stride=W.numel(), rowmap=tl, rows_per_entry=1,
                      dyn_entries=tc, n_groups=4, map_group_stride=N, dyn_group_stride=1, partition=True, planes=True,
                      class_ptr=tc[8:], class_block=d, b_frag=Wft)
    def nt_fused(): ops.gcl_input_grad_fused(dhp, plan, d, Wft, out=dA)
    Aagg = torch.randn(N, 4 * d, device="cuda")
    Ap = ops.split_planes(Aagg)
    dWb = torch.zeros(7 * d, d, device="cuda")
    def tn_unfused():
        ops.gemm_desc(Ap, dhp, dWb, 4 * d, d, N, 4 * d, d, d, transA=True, accum=True, split_k=0, c_group_stride=dd,
                      c_split_rows=d, c_shared_off=3 * dd, a_plane_stride=Aagg.numel(), b_plane_stride=dh.numel(), rowmap=tl,
                      rows_per_entry=1, dyn_entries=tc, n_groups=4, map_group_stride=N, dyn_group_stride=1, partition=True,
                      planes=True, class_ptr=tc[8:], class_block=d)
    def tn_fused(): ops.gcl_weight_grad_fused(Ap, dhp, plan, d, dWb)
    tag = os.path.basename(os.environ.get("PM_LIB_PATH", "default"))
    which = os.environ.get("WHICH", "ufpnw")
    tiles = int(sum((int(c) + 63) // 64 for c in tc[:4].tolist()))
    out = [tag, f"N={N} tiles={tiles}"]
    if "u" in which: out.append(f"unfused {timeit(unfused):.1f} us")
    if "f" in which: out.append(f"fused {timeit(fused):.1f} us")
    if "p" in which: out.append(f"fused(no A' out) {timeit(fused_np):.1f} us")
    if "w" in which: out.append(f"dW grouped {timeit(tn_unfused):.1f} us  128x128 {timeit(tn_fused):.1f} us")
    if "n" in which: out.append(f"dA' grouped {timeit(nt_unfused):.1f} us  A-stationary {timeit(nt_fused):.1f} us")
    print("  ".join(out), flush=True)

if __name__ == "__main__":
    main()
