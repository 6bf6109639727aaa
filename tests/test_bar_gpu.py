"""Bar-resident aggregation of dense graphs (csrc/bar.hip, round 6; BASELINE configs[4]) against the row-gather kernels of
segreduce.hip — which tests/test_kernels_gpu.py pins to the torch restatement of GCL.message + scatter-mean
(model.py:110,123-135) — and against that restatement directly.  Forward: bit-identical planes (same edge order, same
arithmetic); backward: sums of float atomics in a different order, 2e-6."""
import ctypes as _ct

import numpy as np
import pytest
import torch

from polyphemus_amd import ops
from polyphemus_amd._lib import call, ptr, stream
from polyphemus_amd.synthetic import synthetic_batch
from test_kernels_gpu import DEV, _PmH2, _absmax_words, _pair_value, _planes_value, make_plan, segreduce_ref
from util import rel_err

pytestmark = pytest.mark.gpu


class _NormSums(_ct.Structure):          # PmNormSums (include/polyphemus_hip.h)
    _fields_ = [(k, _ct.c_void_p) for k in ("h", "mean", "var", "gamma", "beta")] + \
               [("eps", _ct.c_float), ("relu", _ct.c_int32), ("acc3", _ct.c_void_p), ("absmax_out", _ct.c_void_p)]


def _batch(kind):
    if kind == "dense":                  # every cell active: 128 nodes per bar, 127 in-edges per node
        return synthetic_batch(3, 2, seed=17, dense=True)
    if kind == "sparse":                 # the bench's bars (16 nodes on average): short, unequal segments; quarters / halves run dry
        return synthetic_batch(40, 2, p=0.3, seed=29)
    return synthetic_batch(5, 2, p=0.9, seed=3)      # nearly full bars of the reference's own edge rules


def _inputs(N, d, seed=11):
    torch.manual_seed(seed)
    x = torch.randn(N, d, device=DEV)
    T = ops.edge_table(torch.randn(d, 32, device=DEV) * 0.5, torch.randn(d, device=DEV) * 0.1)
    return x, T


@pytest.mark.parametrize("kind", ["dense", "sparse", "full"])
@pytest.mark.parametrize("d,p", [(128, 0.0), (256, 0.1), (512, 0.1)])
def test_bar_forward_planes_are_the_segment_reduces(kind, d, p):
    b, plan = make_plan(_batch(kind))
    N = plan.N
    x, T = _inputs(N, d)
    P0 = torch.zeros(3, N * 4 * d, dtype=torch.int16, device=DEV)
    P1 = torch.full((3, N * 4 * d), 77, dtype=torch.int16, device=DEV)
    call("pm_segreduce_fwd_planes", ptr(x), ptr(T), ptr(plan.buf), N, plan.E, plan.G, d, p, 5, 2, 1, ptr(P0), N * 4 * d, stream())
    call("pm_bar_aggregate_fwd", ptr(x), ptr(T), ptr(plan.buf), N, plan.E, plan.G, d, p, 5, 2, ptr(P1), N * 4 * d, None, stream())
    assert torch.equal(P0, P1)
    # ... and in the fp16 pair format: (hi + lo) / scale within 2^-21 of the exact aggregate's |max|; plane 2 untouched
    mx, mt = _absmax_words(x), _absmax_words(T)
    sA = torch.zeros(1, device=DEV)
    P2 = torch.zeros(3, N * 4 * d, dtype=torch.int16, device=DEV)
    hh = _PmH2(ptr(mx), ptr(mt), ptr(sA), 16.0, 0)
    call("pm_bar_aggregate_fwd", ptr(x), ptr(T), ptr(plan.buf), N, plan.E, plan.G, d, p, 5, 2, ptr(P2), N * 4 * d, _ct.addressof(hh), stream())
    A3 = _planes_value(P0).double()
    sa = float(sA)
    assert sa > 0 and np.log2(sa) == round(np.log2(sa)) and 2.0 ** 8 <= float(A3.abs().max()) * sa < 2.0 ** 13
    assert float((_pair_value(P2, sa) - A3).abs().max()) <= 2.0 ** -21 * float(A3.abs().max())
    assert P2[2].abs().max() == 0


def test_bar_forward_matches_the_reference_op_chain():
    cpu = _batch("dense")
    b, plan = make_plan(cpu)
    N, d, p = plan.N, 128, 0.1
    x, T = _inputs(N, d)
    P = torch.zeros(3, N * 4 * d, dtype=torch.int16, device=DEV)
    call("pm_bar_aggregate_fwd", ptr(x), ptr(T), ptr(plan.buf), N, plan.E, plan.G, d, p, 77, 3, ptr(P), N * 4 * d, None, stream())
    A = _planes_value(P).double().view(N, 4, d)
    ref = segreduce_ref(x.double(), T.double(), b, p, 77, 3).view(N, 7, d)
    trel = plan.field("node_trel").long()[:N]
    want = torch.stack([ref[torch.arange(N, device=DEV), trel], ref[:, 4], ref[:, 5], ref[:, 6]], dim=1)
    assert rel_err(A, want) < 1e-6


def test_product_from_pair_format_planes_equals_the_triple_product():
    b, plan = make_plan(_batch("dense"))
    N, d, p = plan.N, 512, 0.1
    x, T = _inputs(N, d)
    W = torch.randn(7 * d, d, device=DEV) / d ** 0.5
    bias = torch.randn(d, device=DEV)
    Wf3 = ops.split_planes_frag(W, 1)
    Wf2 = torch.zeros_like(Wf3)
    call("pm_split_planes_frag_h2", ptr(W), 7 * d, d, 1, 1, 7 * d * d, 7 * d * d * 3, 16.0, ptr(Wf2), stream())
    P3 = torch.zeros(3, N * 4 * d, dtype=torch.int16, device=DEV)
    call("pm_bar_aggregate_fwd", ptr(x), ptr(T), ptr(plan.buf), N, plan.E, plan.G, d, p, 5, 2, ptr(P3), N * 4 * d, None, stream())
    s3 = torch.zeros(8, 2, d, dtype=torch.float64, device=DEV)
    h3 = ops.gcl_forward_from_planes(P3, plan, d, Wf3, bias, col_stats=s3)
    mx, mt = _absmax_words(x), _absmax_words(T)
    sA = torch.zeros(1, device=DEV)
    P2 = torch.zeros(3, N * 4 * d, dtype=torch.int16, device=DEV)
    hh = _PmH2(ptr(mx), ptr(mt), ptr(sA), 16.0, 0)
    call("pm_bar_aggregate_fwd", ptr(x), ptr(T), ptr(plan.buf), N, plan.E, plan.G, d, p, 5, 2, ptr(P2), N * 4 * d, _ct.addressof(hh), stream())
    h2 = torch.empty(N, d, device=DEV)
    s2 = torch.zeros(8, 2, d, dtype=torch.float64, device=DEV)
    call("pm_gcl_forward_from_planes_h2", ptr(P2), N * 4 * d, ptr(plan.buf), N, plan.E, plan.G, d, ptr(Wf2), ptr(bias), 1, ptr(h2), ptr(s2),
         ptr(sA), 16.0, stream())
    assert rel_err(h2, h3) < 5e-6
    assert rel_err(s2.sum(0)[0], h2.double().sum(0)) < 1e-12 and rel_err(s2.sum(0)[1], (h2.double() ** 2).sum(0)) < 1e-12


@pytest.mark.parametrize("kind", ["dense", "sparse", "full"])
@pytest.mark.parametrize("d,p,res,norm", [(64, 0.0, True, False), (256, 0.1, False, True), (512, 0.1, True, True), (128, 0.1, False, False)])
def test_bar_backward_equals_the_segment_reduces(kind, d, p, res, norm):
    b, plan = make_plan(_batch(kind))
    N = plan.N
    x, T = _inputs(N, d)
    dA = torch.randn(N, 4 * d, device=DEV)
    dres = torch.randn(N, d, device=DEV) if res else None
    hpre = torch.randn(N, d, device=DEV) * 1.5 + 0.3
    gamma, beta = torch.rand(d, device=DEV) + 0.5, torch.randn(d, device=DEV) * 0.2
    mean, var = hpre.mean(0), hpre.var(0, unbiased=False)
    outs = []
    for fn in ("pm_segreduce_bwd", "pm_bar_aggregate_bwd"):
        dx = torch.full((N, d), float("nan"), device=DEV)
        dT = torch.randn(32, d, device=DEV, generator=torch.Generator(DEV).manual_seed(1))     # (+=)
        dT0 = dT.clone()
        acc3 = torch.zeros(8, 3, d, dtype=torch.float64, device=DEV)
        amax = torch.zeros(64, dtype=torch.int32, device=DEV)
        nn = _NormSums(ptr(hpre), ptr(mean), ptr(var), ptr(gamma), ptr(beta), 1e-5, 1, ptr(acc3), ptr(amax))
        if fn == "pm_segreduce_bwd":
            if norm:
                call("pm_segreduce_bwd_norm", ptr(x), ptr(T), ptr(dA), ptr(dres), ptr(plan.buf), N, plan.E, plan.G, d, p, 9, 4, 1, ptr(dx),
                     ptr(dT), _ct.addressof(nn), stream())
            else:
                call("pm_segreduce_bwd", ptr(x), ptr(T), ptr(dA), ptr(dres), ptr(plan.buf), N, plan.E, plan.G, d, p, 9, 4, 1, ptr(dx), ptr(dT),
                     stream())
        else:
            call("pm_bar_aggregate_bwd", ptr(x), ptr(T), ptr(dA), ptr(dres), ptr(plan.buf), N, plan.E, plan.G, d, p, 9, 4, ptr(dx), ptr(dT),
                 _ct.addressof(nn) if norm else None, stream())
        outs.append((dx, dT - dT0, acc3.sum(0), float(amax.view(torch.float32).max())))
    (dx0, dT0_, a0, m0), (dx1, dT1, a1, m1) = outs
    assert bool(torch.isfinite(dx1).all())
    assert rel_err(dx1, dx0) < 2e-6
    assert rel_err(dT1, dT0_) < 2e-5                     # (differences of fp32 sums over thousands of edges added to a random table)
    if norm:
        assert rel_err(a1, a0) < 1e-6
        assert m1 == float(dx1.abs().max())               # the |max| words: exact (d <= 256 only in the row-gather kernel)
        if d <= 256:
            assert m0 == float(dx0.abs().max())


def test_bar_backward_matches_autograd_of_the_reference_op_chain():
    cpu = _batch("dense")
    b, plan = make_plan(cpu)
    N, d, p = plan.N, 64, 0.1
    x, T = _inputs(N, d)
    xr, Tr = x.double().requires_grad_(True), T.double().requires_grad_(True)
    ref = segreduce_ref(xr, Tr, b, p, 77, 3).view(N, 7, d)
    trel = plan.field("node_trel").long()[:N]
    dA = torch.randn(N, 4, d, device=DEV)
    full = torch.zeros(N, 7, d, device=DEV, dtype=torch.float64)
    full[torch.arange(N, device=DEV), trel] = dA[:, 0].double()
    full[:, 4], full[:, 5], full[:, 6] = dA[:, 1].double(), dA[:, 2].double(), dA[:, 3].double()
    ref.backward(full)
    dx = torch.empty(N, d, device=DEV)
    dT = torch.zeros(32, d, device=DEV)
    call("pm_bar_aggregate_bwd", ptr(x), ptr(T), ptr(dA), None, ptr(plan.buf), N, plan.E, plan.G, d, p, 77, 3, ptr(dx), ptr(dT), None, stream())
    assert rel_err(dx, xr.grad) < 1e-5 and rel_err(dT, Tr.grad) < 1e-5


def test_bar_kernels_at_the_dense_shards_real_size():
    """configs[4], one GPU's shard at its REAL size (B = 64: N = 16,384 nodes, E = 2.08 M edges, d = 512, message dropout on) —
    too large for the CPU oracle's per-edge fp64 tensors, not for a kernel-level comparison: the bar-resident forward gives the
    row-gather kernel's planes bit for bit, the pair-format planes decode to them within 2^-21 of |max|, and the backward (with the
    norm sums of the layer below) agrees with `pm_segreduce_bwd_norm` to 2e-6 — the kernels tests/test_kernels_gpu.py pins to the
    torch restatement of GCL.message + scatter-mean (model.py:110,123-135)."""
    b, plan = make_plan(synthetic_batch(64, 2, seed=1234, dense=True))
    N, d, p = plan.N, 512, 0.1
    assert N == 16384 and plan.E == 2080768
    x, T = _inputs(N, d, seed=3)
    P0 = torch.zeros(3, N * 4 * d, dtype=torch.int16, device=DEV)
    P1 = torch.zeros(3, N * 4 * d, dtype=torch.int16, device=DEV)
    call("pm_segreduce_fwd_planes", ptr(x), ptr(T), ptr(plan.buf), N, plan.E, plan.G, d, p, 5, 2, 1, ptr(P0), N * 4 * d, stream())
    call("pm_bar_aggregate_fwd", ptr(x), ptr(T), ptr(plan.buf), N, plan.E, plan.G, d, p, 5, 2, ptr(P1), N * 4 * d, None, stream())
    assert torch.equal(P0, P1)
    mx, mt = _absmax_words(x), _absmax_words(T)
    sA = torch.zeros(1, device=DEV)
    hh = _PmH2(ptr(mx), ptr(mt), ptr(sA), 16.0, 0)
    P1.zero_()
    call("pm_bar_aggregate_fwd", ptr(x), ptr(T), ptr(plan.buf), N, plan.E, plan.G, d, p, 5, 2, ptr(P1), N * 4 * d, _ct.addressof(hh), stream())
    A3 = _planes_value(P0).double()
    assert float((_pair_value(P1, float(sA)) - A3).abs().max()) <= 2.0 ** -21 * float(A3.abs().max())
    del P0, P1, A3
    dA = torch.randn(N, 4 * d, device=DEV)
    hpre = torch.randn(N, d, device=DEV) * 1.5 + 0.3
    gamma, beta = torch.rand(d, device=DEV) + 0.5, torch.randn(d, device=DEV) * 0.2
    mean, var = hpre.mean(0), hpre.var(0, unbiased=False)
    outs = []
    for fn in ("pm_segreduce_bwd_norm", "pm_bar_aggregate_bwd"):
        dx = torch.empty(N, d, device=DEV)
        dT = torch.zeros(32, d, device=DEV)
        acc3 = torch.zeros(8, 3, d, dtype=torch.float64, device=DEV)
        amax = torch.zeros(64, dtype=torch.int32, device=DEV)
        nn = _NormSums(ptr(hpre), ptr(mean), ptr(var), ptr(gamma), ptr(beta), 1e-5, 1, ptr(acc3), ptr(amax))
        args = (ptr(x), ptr(T), ptr(dA), None, ptr(plan.buf), N, plan.E, plan.G, d, p, 9, 4)
        if fn == "pm_segreduce_bwd_norm":
            call(fn, *args, 1, ptr(dx), ptr(dT), _ct.addressof(nn), stream())
        else:
            call(fn, *args, ptr(dx), ptr(dT), _ct.addressof(nn), stream())
        outs.append((dx, dT, acc3.sum(0), float(amax.view(torch.float32).max())))
    (dx0, dT0, a0, _), (dx1, dT1, a1, m1) = outs
    assert rel_err(dx1, dx0) < 2e-6 and rel_err(dT1, dT0) < 2e-5 and rel_err(a1, a0) < 1e-6
    assert m1 == float(dx1.abs().max())
