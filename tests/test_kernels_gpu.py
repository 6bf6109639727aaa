"""Per-kernel parity tests (GPU): every C-ABI entry point against a plain torch / numpy
restatement of the reference op chain it replaces.  Integer work is compared bit-exactly,
floating point with the 1e-4 relative bar of BASELINE.json (most kernels sit near 1e-6)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from polyphemus_amd import constants as C
from polyphemus_amd import ops
from polyphemus_amd._lib import PROF_NCLASS, call, lib, ptr, stream
from polyphemus_amd.synthetic import synthetic_batch
from util import REL_TOL, rel_err, dropout_keep_np

pytestmark = pytest.mark.gpu
DEV = "cuda"


def make_plan(batch):
    b = batch.to(DEV)
    G = b.s_tensor.shape[0]
    plan = ops.plan_build(b.edge_index, b.edge_type, b.edge_dist, b.bars, b.batch, b.is_drum, b.tokens, b.n_bars, G)
    return b, plan


@pytest.fixture(scope="module")
def small():
    return make_plan(synthetic_batch(6, 2, p=0.25, seed=5))


# ------------------------------------------------------------------ plan (integer, bit-exact)
@pytest.mark.parametrize("dense", [False, True])
def test_plan_matches_numpy(dense):
    cpu = synthetic_batch(3 if dense else 9, 2, p=0.3, seed=21, dense=dense)
    b, plan = make_plan(cpu)
    N, E = cpu.num_nodes, cpu.edge_index.shape[1]
    src, dst = cpu.edge_index[0].numpy(), cpu.edge_index[1].numpy()
    et, ed = cpu.edge_type.numpy(), cpu.edge_dist.numpy()
    key = dst * 6 + et
    order = np.lexsort((np.arange(E), key))                  # by key, ties by edge id
    rowptr = np.zeros(N * 6 + 1, np.int64)
    np.add.at(rowptr, key + 1, 1)
    rowptr = np.cumsum(rowptr)
    f = lambda n: plan.field(n).cpu().numpy()
    np.testing.assert_array_equal(f("rowptr")[:N * 6 + 1], rowptr)
    np.testing.assert_array_equal(f("csr_eid")[:E], order)
    np.testing.assert_array_equal(f("csr_src")[:E], src[order])
    np.testing.assert_array_equal(f("csr_dist")[:E], ed[order])
    order2 = np.lexsort((np.arange(E), ed, src))             # by source, then distance, ties by edge id (runs of equal distance)
    colptr = np.zeros(N + 1, np.int64)
    np.add.at(colptr, src + 1, 1)
    np.testing.assert_array_equal(f("colptr")[:N + 1], np.cumsum(colptr))
    np.testing.assert_array_equal(f("csc_eid")[:E], order2)
    np.testing.assert_array_equal(f("csc_dst")[:E], dst[order2])
    np.testing.assert_array_equal(f("csc_reldist")[:E], et[order2] | (ed[order2] << 8))
    cnt = (rowptr[1:] - rowptr[:-1])[key[order2]]
    np.testing.assert_array_equal(f("csc_invcnt")[:E], (1.0 / np.maximum(cnt, 1)).astype(np.float32))
    nb = (cpu.bars + cpu.n_bars * cpu.batch).numpy()
    np.testing.assert_array_equal(f("node_bar")[:N], nb)
    G = cpu.s_tensor.shape[0]
    np.testing.assert_array_equal(f("bar_ptr")[:G + 1], np.concatenate([[0], np.cumsum(np.bincount(nb, minlength=G))]))
    drum = cpu.is_drum.numpy()
    nd = int(drum.sum())
    np.testing.assert_array_equal(f("group_list")[:nd], np.nonzero(drum)[0])
    np.testing.assert_array_equal(f("group_list")[N:N + (N - nd)], np.nonzero(~drum)[0])
    assert f("group_cnt")[:2].tolist() == [int(drum.sum()), int((~drum).sum())]
    tok = cpu.tokens.numpy()[:, 1:, :]
    hist = np.zeros((4, 131), np.int64)
    for g, m in enumerate((drum, ~drum)):
        hist[g] = np.bincount(tok[m][..., 0].ravel(), minlength=131)
        hist[2 + g] = np.bincount(tok[m][..., 1].ravel(), minlength=131)
    np.testing.assert_array_equal(f("tok_hist")[:4 * 131].reshape(4, 131), hist)


def test_reference_format_inputs_to_ids(small):
    b, plan = small
    et, ed = ops.edge_attrs_to_ids(b.edge_attrs.contiguous())
    assert torch.equal(et, b.edge_type) and torch.equal(ed, b.edge_dist)
    tok = ops.tokens_from_onehot(b.c_tensor.contiguous())
    assert torch.equal(tok, b.tokens)


# ------------------------------------------------------------------ message aggregation
def segreduce_ref(x, T, b, p, seed, layer):
    """reference op chain of GCL.message + scatter-mean (model.py:110,123-135) in torch."""
    N, d = x.shape
    src, dst = b.edge_index[0], b.edge_index[1]
    parts = []
    for r in range(6):
        m = b.edge_type == r
        msg = F.relu(x[src[m]] * T[b.edge_dist[m].long()])
        if p > 0:
            eids = torch.nonzero(m).flatten().cpu().numpy()
            keep = torch.from_numpy(dropout_keep_np(seed, layer, eids, d, p)).to(x.device)
            msg = msg * keep / (1.0 - p)
        h = torch.zeros(N, d, device=x.device, dtype=x.dtype).index_add_(0, dst[m], msg)
        cnt = torch.zeros(N, device=x.device, dtype=x.dtype).index_add_(0, dst[m], torch.ones(int(m.sum()), device=x.device, dtype=x.dtype))
        parts.append(h / cnt.clamp(min=1).unsqueeze(1))
    parts.append(x)
    return torch.cat(parts, dim=1)


@pytest.mark.parametrize("d,p", [(32, 0.0), (256, 0.0), (256, 0.1), (512, 0.1), (40, 0.25)])
def test_segreduce_fwd_bwd(small, d, p):
    b, plan = small
    torch.manual_seed(d)
    N = plan.N
    x = torch.randn(N, d, device=DEV)
    W = torch.randn(d, 32, device=DEV) * 0.5
    bias = torch.randn(d, device=DEV) * 0.1
    T = ops.edge_table(W, bias)
    assert rel_err(T, (W.t() + bias)) < 1e-7
    A = ops.segreduce_fwd(x, T, plan, p, 77, 3)
    xr = x.double().requires_grad_(True)
    Tr = T.double().requires_grad_(True)
    ref = segreduce_ref(xr, Tr, b, p, 77, 3)
    assert rel_err(A, ref.detach()) < 1e-6
    dA = torch.randn(N, 7 * d, device=DEV)
    dres = torch.randn(N, d, device=DEV)
    ref.backward(dA.double())
    dT = torch.zeros(32, d, device=DEV)
    dx = ops.segreduce_bwd(x, T, dA, dres, plan, p, 77, 3, dT)
    assert rel_err(dx, xr.grad + dres.double()) < 1e-5
    assert rel_err(dT, Tr.grad) < 1e-5
    dW, db = torch.zeros_like(W), torch.zeros_like(bias)
    ops.edge_table_bwd(dT, dW, db)
    assert rel_err(dW, dT.t()) < 1e-7 and rel_err(db, dT.sum(0)) < 1e-6


def test_dropout_hash_matches_numpy():
    L = lib()
    eids = np.array([0, 1, 17, 123456], np.int64)
    keep = dropout_keep_np(99, 5, eids, 8, 0.1)
    for i, e in enumerate(eids):
        for c in range(8):
            assert (L.pm_dropout_hash(99, 5, int(e), c) >= int(0.1 * 16777216.0)) == bool(keep[i, c])


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K,ta,tb", [
    (300, 256, 1792, False, False), (1000, 64, 96, False, True), (130, 131, 128, False, True),
    (1792, 256, 3000, True, False), (131, 128, 777, True, False), (256, 512, 256, False, True),
    (37, 19, 11, False, False), (4096, 256, 224, False, False), (64, 230, 40, False, True)])
def test_gemm_variants(M, N, K, ta, tb):
    torch.manual_seed(M + N + K)
    A = torch.randn((K, M) if ta else (M, K), device=DEV)
    B = torch.randn((N, K) if tb else (K, N), device=DEV)
    bias = torch.randn(N, device=DEV)
    ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
    out = torch.empty(M, N, device=DEV)
    ops.gemm(A, B, out, M, N, K, A.stride(0), B.stride(0), N, transA=ta, transB=tb, bias=bias, relu=True)
    assert rel_err(out, F.relu(ref + bias.double())) < 5e-6
    acc = torch.randn(M, N, device=DEV)
    want = acc.double() + ref
    ops.gemm(A, B, acc, M, N, K, A.stride(0), B.stride(0), N, transA=ta, transB=tb, accum=True, split_k=0)
    assert rel_err(acc, want) < 5e-6


def test_gemm_strided_views_and_rowmap():
    """the content-decoder routing: rows = (node, slot) pairs of a node list, strided A and C."""
    torch.manual_seed(3)
    Nn, d = 50, 64
    H = torch.randn(Nn, 15, d, device=DEV)
    Wp = torch.randn(131, d // 2, device=DEV)
    logits = torch.zeros(Nn, 15, 230, device=DEV)
    nodes = torch.tensor([3, 7, 8, 20, 41, 49], dtype=torch.int32, device=DEV)
    cnt = torch.tensor([5], dtype=torch.int32, device=DEV)               # only the first 5 entries are live
    ops.gemm(H, Wp, logits, nodes.numel() * 15, 131, d // 2, d, d // 2, 230, transB=True, rowmap=nodes,
             rows_per_entry=15, dyn_entries=cnt)
    ref = torch.zeros_like(logits)
    sel = nodes[:5].long()
    ref[sel, :, :131] = (H[sel, :, :d // 2].double() @ Wp.double().t()).float()
    assert rel_err(logits, ref) < 5e-6
    # duration half: A = H[..., d/2:], C = logits[..., 131:]  (misaligned C offset, scalar epilogue)
    Wd = torch.randn(99, d // 2, device=DEV)
    ops.gemm(H.view(-1)[d // 2:], Wd, logits.view(-1)[131:], Nn * 15, 99, d // 2, d, d // 2, 230, transB=True)
    assert rel_err(logits[..., 131:], H[..., d // 2:].double() @ Wd.double().t()) < 5e-6
    # weight gradient with a gathered K: dW = dlogits[rows]^T @ H[rows]
    dl = torch.randn(Nn, 15, 230, device=DEV)
    dW = torch.zeros(131, d // 2, device=DEV)
    ops.gemm(dl, H, dW, 131, d // 2, nodes.numel() * 15, 230, d, d // 2, transA=True, accum=True, split_k=0,
             rowmap=nodes, rows_per_entry=15, dyn_entries=cnt)
    want = dl[sel, :, :131].double().reshape(-1, 131).t() @ H[sel, :, :d // 2].double().reshape(-1, d // 2)
    assert rel_err(dW, want) < 5e-6


@pytest.mark.parametrize("M,N,K,ta,tb,lda,ldb", [
    (5000, 128, 131, False, False, 232, 128),      # dH = d_logits[:, :131] @ W: K tail inside 16-byte aligned rows
    (5000, 128, 99, False, False, 232, 128),
    (131, 128, 5000, True, False, 232, 256),       # dW = d_logits^T H: M tail
    (99, 128, 3001, True, False, 232, 256),
    (300, 131, 67, False, True, 68, 68),           # x @ W^T with K = 67 (both operands k-contiguous, K tail in both)
    (70, 131, 130, False, False, 132, 132),        # N tail of a [K, N] operand
])
def test_gemm_aligned_rows_with_ragged_extents(M, N, K, ta, tb, lda, ldb):
    """fp32 tile kernels: 16-byte staging whenever the ROWS are aligned (leading dimensions multiples of 4), whatever the
    extents — the float4 that straddles the end of an extent reads the row's next columns (here NaN): a K tail is zeroed
    when staged, M / N tails reach accumulator rows / columns that are never stored."""
    torch.manual_seed(M + N + K)
    A = torch.full((K, lda) if ta else (M, lda), float("nan"), device=DEV)
    B = torch.full((N, ldb) if tb else (K, ldb), float("nan"), device=DEV)
    a = torch.randn((K, M) if ta else (M, K), device=DEV)
    b = torch.randn((N, K) if tb else (K, N), device=DEV)
    A[:, :a.shape[1]] = a
    B[:, :b.shape[1]] = b
    ref = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())
    out = torch.full((M, N + 3), 7.0, device=DEV)
    ops.gemm(A, B, out, M, N, K, lda, ldb, N + 3, transA=ta, transB=tb, **({"accum": True, "split_k": 0} if ta else {}))
    want = ref + (7.0 if ta else 0.0)
    assert rel_err(out[:, :N], want) < 5e-6
    assert bool((out[:, N:] == 7.0).all())


@pytest.mark.parametrize("Kr,M,N,ldx", [(512, 256, 512, 0), (16271, 256, 1280, 0), (300, 99, 68, 0), (256, 512, 256, 1024),
                                         (77, 1, 256, 0), (4100, 1280, 256, 0)])
def test_gemm_weight_gradient_carries_the_bias_gradient(Kr, M, N, ldx):
    """PmGemmDesc.a_colsum: the weight-gradient product dW += dy^T x also leaves db += column sums of dy (what
    `lin_bwd` of the native step issues for every plain linear layer: one launch instead of two).  Both accumulate."""
    torch.manual_seed(Kr + M + N)
    dy = torch.randn(Kr, M, device=DEV)
    xs = torch.randn(Kr, ldx or N, device=DEV)
    x = xs[:, :N]
    dW0, db0 = torch.randn(M, N, device=DEV), torch.randn(M, device=DEV)
    dW, db = dW0.clone(), db0.clone()
    ops.gemm_desc(dy, xs, dW, M, N, Kr, M, xs.stride(0), N, transA=True, accum=True, split_k=0, a_colsum=db)
    assert rel_err(dW, dW0.double() + dy.double().t() @ x.double()) < 5e-6
    assert rel_err(db, db0.double() + dy.double().sum(0)) < 5e-6


@pytest.mark.parametrize("cfg", [4, 5, 6, 7])
@pytest.mark.parametrize("M,N,K,ta,tb", [
    (1000, 256, 1024, False, False), (333, 1024, 256, False, True), (1024, 256, 3001 * 4, True, False),
    (130, 132, 36, False, True), (515, 260, 72, False, False), (260, 68, 1028, True, False)])
def test_gemm_split_mode(cfg, M, N, K, ta, tb):
    """split mode (three-term bf16 operand split, six products, fp32 accumulate): same bound as the fp32 tiles."""
    torch.manual_seed(M + N + K + cfg)
    A = torch.randn((K, M) if ta else (M, K), device=DEV) * 3
    B = torch.randn((N, K) if tb else (K, N), device=DEV)
    bias = torch.randn(N, device=DEV)
    ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
    try:
        lib().pm_gemm_force_config(cfg)
        out = torch.full((M, N), float("nan"), device=DEV)
        ops.gemm(A, B, out, M, N, K, A.stride(0), B.stride(0), N, transA=ta, transB=tb, bias=bias, relu=True)
        assert rel_err(out, F.relu(ref + bias.double())) < 5e-6
        acc = torch.randn(M, N, device=DEV)
        want = acc.double() + ref
        ops.gemm(A, B, acc, M, N, K, A.stride(0), B.stride(0), N, transA=ta, transB=tb, accum=True, split_k=0)
        assert rel_err(acc, want) < 5e-6
    finally:
        lib().pm_gemm_force_config(-1)


def test_split_planes_is_exact():
    torch.manual_seed(1)
    x = torch.randn(4096, device=DEV) * torch.logspace(-20, 20, 4096, device=DEV)
    p = ops.split_planes(x)
    f = lambda t: (t.to(torch.int32) << 16).view(torch.float32).double()     # bf16 bits -> value
    assert torch.equal((f(p[0]) + f(p[1]) + f(p[2])).float(), x)


@pytest.mark.parametrize("planes", [False, True, "bfrag"])
@pytest.mark.parametrize("partition", [False, True])
@pytest.mark.parametrize("Nn,d", [(700, 64), (1000, 32), (333, 128), (2100, 256)])
def test_gemm_grouped_stacked_compact_gcl(Nn, d, partition, planes):
    """The three contractions of the compact GCL (model.py:112,116 on [track block | onset | next | x]):
    rows partitioned into four relation groups (row lists + device counts), B / C stacked as
    [weight[t] (group rows) ; weight[4]; weight[5]; root (shared rows)]."""
    torch.manual_seed(Nn + d)
    trel = torch.randint(0, 4, (Nn,), device=DEV)
    trel[: Nn // 3] = 2                                           # uneven groups
    lists = torch.full((4, Nn), -7, dtype=torch.int32, device=DEV)   # entries past the live count are never read
    cnt = torch.zeros(8, dtype=torch.int32, device=DEV)
    for t in range(4):
        rows = torch.nonzero(trel == t).flatten().to(torch.int32)
        lists[t, : rows.numel()] = rows[torch.randperm(rows.numel(), device=DEV)]
        cnt[t] = rows.numel()
    A = torch.randn(Nn, 4 * d, device=DEV)
    W = torch.randn(7 * d, d, device=DEV)                          # [W_0..W_3 | W_4 | W_5 | root]
    bias = torch.randn(d, device=DEV)
    dd = d * d
    Wn = torch.stack([torch.cat([W[t * d:(t + 1) * d], W[4 * d:]]) for t in range(4)]).double()  # [4, 4d, d]
    grp = dict(rowmap=lists, rows_per_entry=1, dyn_entries=cnt, n_groups=4, map_group_stride=Nn, dyn_group_stride=1,
               partition=partition)
    if planes:                                                    # operands pre-split into three bf16 planes
        Ap, Wp = ops.split_planes(A), ops.split_planes(W)
        pa, pw = dict(planes=True, a_plane_stride=A.numel(), b_plane_stride=W.numel()), None
    # "bfrag": the weights additionally as fragment-major planes (B taken straight into registers where the shape allows)
    fn = dict(b_frag=ops.split_planes_frag(W, 1)) if planes == "bfrag" else {}
    ft = dict(b_frag=ops.split_planes_frag(W, 0)) if planes == "bfrag" else {}
    # forward
    h = torch.full((Nn, d), float("nan"), device=DEV)
    if planes:
        ops.gemm_desc(Ap, Wp, h, Nn, d, 4 * d, 4 * d, d, d, bias=bias, b_group_stride=dd, b_split_rows=d,
                      b_shared_off=3 * dd, **pa, **grp, **fn)
    else:
        ops.gemm_desc(A, W, h, Nn, d, 4 * d, 4 * d, d, d, bias=bias, b_group_stride=dd, b_split_rows=d,
                      b_shared_off=3 * dd, **grp)
    want = torch.einsum("nk,nkj->nj", A.double(), Wn[trel]) + bias.double()
    assert rel_err(h, want) < 5e-6
    # input gradient
    dh = torch.randn(Nn, d, device=DEV)
    dA = torch.full((Nn, 4 * d), float("nan"), device=DEV)
    if planes:
        dhp = ops.split_planes(dh)
        ops.gemm_desc(dhp, Wp, dA, Nn, 4 * d, d, d, d, 4 * d, transB=True, b_group_stride=dd, b_split_rows=d,
                      b_shared_off=3 * dd, planes=True, a_plane_stride=dh.numel(), b_plane_stride=W.numel(), **grp, **ft)
    else:
        ops.gemm_desc(dh, W, dA, Nn, 4 * d, d, d, d, 4 * d, transB=True, b_group_stride=dd, b_split_rows=d,
                      b_shared_off=3 * dd, **grp)
    assert rel_err(dA, torch.einsum("nj,nkj->nk", dh.double(), Wn[trel])) < 5e-6
    # weight gradient (group rows plain / shared rows collected from all groups), accumulating
    dW = torch.randn(7 * d, d, device=DEV)
    want = dW.double().clone()
    for t in range(4):
        m = trel == t
        want[t * d:(t + 1) * d] += A[m, :d].double().t() @ dh[m].double()
    want[4 * d:] += A[:, d:].double().t() @ dh.double()
    if planes:
        ops.gemm_desc(Ap, dhp, dW, 4 * d, d, Nn, 4 * d, d, d, transA=True, accum=True, split_k=0, c_group_stride=dd,
                      c_split_rows=d, c_shared_off=3 * dd, planes=True, a_plane_stride=A.numel(),
                      b_plane_stride=dh.numel(), **grp)
    else:
        ops.gemm_desc(A, dh, dW, 4 * d, d, Nn, 4 * d, d, d, transA=True, accum=True, split_k=0, c_group_stride=dd,
                      c_split_rows=d, c_shared_off=3 * dd, **grp)
    assert rel_err(dW, want) < 5e-6


# ------------------------------------------------------------------ batch norm
@pytest.mark.parametrize("shape,I", [((1000, 256), 1), ((37, 24), 1), ((500, 1), 1), ((64, 8, 4, 32), 128),
                                     ((20, 16, 4, 8), 32)])
def test_bn_train_fwd_bwd(shape, I):
    torch.manual_seed(len(shape) + shape[0])
    x = torch.randn(*shape, device=DEV) * 2 + 0.5
    Cn = shape[1]
    O = shape[0]
    g = torch.rand(Cn, device=DEV) + 0.5
    be = torch.randn(Cn, device=DEV) * 0.3
    rm, rv = torch.zeros(Cn, device=DEV), torch.ones(Cn, device=DEV)
    res = torch.randn_like(x)
    mean, var = ops.bn_stats(x, O, Cn, I, rm, rv)
    y = ops.bn_apply(x, O, Cn, I, mean, var, g, be, residual=res, relu=True)
    xr = x.double().requires_grad_(True)
    gr, ber = g.double().requires_grad_(True), be.double().requires_grad_(True)
    rm2, rv2 = torch.zeros(Cn, device=DEV, dtype=torch.float64), torch.ones(Cn, device=DEV, dtype=torch.float64)
    yr = F.relu(F.batch_norm(xr, rm2, rv2, gr, ber, True, 0.1, 1e-5)) + res.double()
    assert rel_err(y, yr.detach()) < 1e-5
    assert rel_err(rm, rm2) < 1e-5 and rel_err(rv, rv2) < 1e-5
    dy = torch.randn_like(x)
    yr.backward(dy.double())
    dg, db = torch.zeros(Cn, device=DEV), torch.zeros(Cn, device=DEV)
    dx = ops.bn_bwd(x, dy, O, Cn, I, mean, var, g, be, dg, db, relu=True)
    assert rel_err(dx, xr.grad) < 1e-5
    assert rel_err(dg, gr.grad) < 1e-5 and rel_err(db, ber.grad) < 1e-5


def test_split_planes_frag_layout():
    """`pm_split_planes_frag`: per (32-wide n tile, 16-wide k-step, plane) a 1 KiB block in MFMA operand order; the three
    planes still sum to the fp32 value exactly."""
    torch.manual_seed(2)
    W = torch.randn(2, 96, 64, device=DEV) * torch.logspace(-3, 3, 64, device=DEV)
    f = lambda t: (t.to(torch.int32) << 16).view(torch.float32)
    for kind in (0, 1):
        out = ops.split_planes_frag(W, kind)                          # [2, 96*64*3]
        for m in range(2):
            Wm = W[m] if kind == 0 else W[m].t()                     # [n, k]
            n_tot, k_tot = Wm.shape
            blocks = out[m].view(-1, 3, 64, 8)                       # [block, plane, lane, 8]
            val = (f(blocks[:, 0]) + f(blocks[:, 1]) + f(blocks[:, 2]))   # [block, lane, 8]
            if kind == 0:
                val = val.view(n_tot // 32, k_tot // 16, 2, 32, 8)   # [ntile, kstep, half, n, e]
                got = val.permute(0, 3, 1, 2, 4).reshape(n_tot, k_tot)
            else:
                val = val.view(k_tot // 16, n_tot // 32, 2, 32, 8)   # [kstep, ntile, half, n, e]
                got = val.permute(1, 3, 0, 2, 4).reshape(n_tot, k_tot)
            assert torch.equal(got, Wm)


@pytest.mark.parametrize("M,N,K,tb", [(1000, 256, 1024, False), (777, 1024, 256, True), (64, 128, 32, True),
                                       (130, 384, 96, False)])
def test_gemm_planes_b_direct(M, N, K, tb):
    """planes-mode GEMM with the weight operand as fragment-major planes (config 9) against fp64, with bias / ReLU /
    column statistics in the epilogue like the GCL forward."""
    torch.manual_seed(M + N)
    A = torch.randn(M, K, device=DEV)
    W = torch.randn((N, K) if tb else (K, N), device=DEV)
    bias = torch.randn(N, device=DEV)
    ref = A.double() @ (W.double().t() if tb else W.double()) + bias.double()
    out = torch.full((M, N), float("nan"), device=DEV)
    sums = torch.zeros(8, 2, N, dtype=torch.float64, device=DEV)
    ops.gemm_desc(ops.split_planes(A), ops.split_planes(W), out, M, N, K, K, K if tb else N, N, transB=tb, bias=bias,
                  planes=True, a_plane_stride=A.numel(), b_plane_stride=W.numel(), col_stats=sums,
                  b_frag=ops.split_planes_frag(W, 0 if tb else 1))
    assert rel_err(out, ref) < 5e-6
    assert rel_err(sums.sum(0)[0], out.double().sum(0)) < 1e-9


def _planes_value(p):
    """int16 [3, n] bf16 planes -> float64 value p1 + p2 + p3"""
    f = lambda t: (t.to(torch.int32) << 16).view(torch.float32).double()
    return f(p[0]) + f(p[1]) + f(p[2])


@pytest.mark.parametrize("O,C,relu,res", [(256, 256, True, False), (256, 512, True, True), (37, 24, False, True), (2048, 40, True, False),
                                          (1, 8, False, False)])
def test_bn_small_batch_one_launch_per_direction(O, C, relu, res):
    """`pm_bn_small_fwd / _bwd` (the norms of the heads: statistics + running statistics + apply, and backward sums + dx, in
    one launch each) against torch's training-mode batch_norm in fp64."""
    torch.manual_seed(O + C)
    x = torch.randn(O, C, device=DEV) * 2 + 0.5
    g, be = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.3
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    r = torch.randn_like(x) if res else None
    y, mean, var = torch.full_like(x, float("nan")), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    call("pm_bn_small_fwd", ptr(x), O, C, 1e-5, ptr(g), ptr(be), ptr(r), int(relu), ptr(y), ptr(mean), ptr(var), ptr(rm), ptr(rv),
         0.1, stream())
    xr = x.double().requires_grad_(True)
    gr, ber = g.double().requires_grad_(True), be.double().requires_grad_(True)
    rm2, rv2 = torch.zeros(C, device=DEV, dtype=torch.float64), torch.ones(C, device=DEV, dtype=torch.float64)
    if O > 1:
        yr = F.batch_norm(xr, rm2, rv2, gr, ber, True, 0.1, 1e-5)
    else:                                  # (torch refuses one row in training mode; the kernel gives var = 0)
        yr = (xr - xr.mean(0)) * torch.rsqrt(torch.zeros(C, device=DEV, dtype=torch.float64) + 1e-5) * gr + ber
    if relu:
        yr = F.relu(yr)
    if res:
        yr = yr + r.double()
    assert rel_err(y, yr.detach()) < 1e-5
    assert rel_err(mean, x.double().mean(0)) < 1e-6
    if O > 1:
        assert rel_err(var, x.double().var(0, unbiased=False)) < 1e-5 and rel_err(rm, rm2) < 1e-5 and rel_err(rv, rv2) < 1e-5
    dy = torch.randn(O, C, device=DEV)
    yr.backward(dy.double())
    dg, db, dbp = torch.ones(C, device=DEV), torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    dx = torch.full_like(x, float("nan"))
    call("pm_bn_small_bwd", ptr(x), ptr(dy), O, C, ptr(mean), ptr(var), 1e-5, ptr(g), ptr(be), int(relu), ptr(dg), ptr(db),
         ptr(dbp), ptr(dx), stream())
    assert rel_err(dx, xr.grad) < 2e-5
    assert rel_err(dg - 1, gr.grad) < 1e-5 and rel_err(db - 1, ber.grad) < 1e-5          # (+=)
    assert float(dbp.abs().max()) < 1e-3 * max(1e-30, float(dy.abs().sum(0).max()))


@pytest.mark.parametrize("M,K,C", [(1000, 64, 256), (333, 40, 32), (5000, 128, 128)])
def test_gcl_norm_fused_with_gemm_epilogue(M, K, C):
    """The GCL norm of the native step: column statistics accumulated by the GEMM epilogue (col_stats, replicated
    fp64 accumulators), pm_bn_apply_fused (mean / var / running stats as a side effect, + ReLU + residual) and
    pm_bn_bwd_fused (atomics instead of a finalize launch; dx as fp32 and as three bf16 planes)."""
    torch.manual_seed(M + C)
    R = 8                                                              # PM_BN_REPL
    A = torch.randn(M, K, device=DEV)
    W = torch.randn(K, C, device=DEV) * 0.3
    bias = torch.randn(C, device=DEV)
    h = torch.empty(M, C, device=DEV)
    sums = torch.zeros(R, 2, C, dtype=torch.float64, device=DEV)
    ops.gemm_desc(A, W, h, M, C, K, K, C, C, bias=bias, col_stats=sums)
    href = A.double() @ W.double() + bias.double()
    assert rel_err(h, href) < 5e-6
    assert rel_err(sums.sum(0)[0], h.double().sum(0)) < 1e-12 and rel_err(sums.sum(0)[1], (h.double() ** 2).sum(0)) < 1e-12
    g, be = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.3
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    res = torch.randn(M, C, device=DEV)
    y, mean, var = torch.empty_like(h), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    call("pm_bn_apply_fused", ptr(h), M, C, ptr(sums), 1e-5, ptr(g), ptr(be), ptr(res), 1, ptr(y), ptr(mean), ptr(var),
         ptr(rm), ptr(rv), 0.1, stream())
    xr = h.double().requires_grad_(True)
    gr, ber = g.double().requires_grad_(True), be.double().requires_grad_(True)
    rm2, rv2 = torch.zeros(C, device=DEV, dtype=torch.float64), torch.ones(C, device=DEV, dtype=torch.float64)
    yr = F.relu(F.batch_norm(xr, rm2, rv2, gr, ber, True, 0.1, 1e-5)) + res.double()
    assert rel_err(y, yr.detach()) < 1e-5
    assert rel_err(mean, h.double().mean(0)) < 1e-6 and rel_err(var, h.double().var(0, unbiased=False)) < 1e-5
    assert rel_err(rm, rm2) < 1e-5 and rel_err(rv, rv2) < 1e-5
    dy = torch.randn(M, C, device=DEV)
    yr.backward(dy.double())
    for planes in (False, True):
        dg, db, dbp = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        acc3 = torch.zeros(R, 3, C, dtype=torch.float64, device=DEV)
        dx = torch.full((M, C), float("nan"), device=DEV)
        dxp = torch.zeros(3, M * C, dtype=torch.int16, device=DEV)
        call("pm_bn_bwd_fused", ptr(h), ptr(dy), M, C, ptr(mean), ptr(var), 1e-5, ptr(g), ptr(be), 1, ptr(dg), ptr(db),
             ptr(dbp), None if planes else ptr(dx), ptr(acc3), ptr(dxp) if planes else None, M * C, 0, stream())
        got = _planes_value(dxp).view(M, C) if planes else dx
        assert rel_err(got, xr.grad) < 1e-5
        assert rel_err(dg, gr.grad) < 1e-5 and rel_err(db, ber.grad) < 1e-5
        assert float(dbp.abs().max()) < 1e-3 * float(dy.abs().sum(0).max())   # bias in front of a batch-stat norm: zero


def test_segreduce_planes_output_equals_fp32_output(small):
    b, plan = small
    torch.manual_seed(11)
    N, d = plan.N, 64
    x = torch.randn(N, d, device=DEV)
    T = ops.edge_table(torch.randn(d, 32, device=DEV) * 0.5, torch.randn(d, device=DEV) * 0.1)
    for compact in (0, 1):
        nb = 4 if compact else 7
        A = torch.empty(N, nb * d, device=DEV)
        P = torch.zeros(3, N * nb * d, dtype=torch.int16, device=DEV)
        call("pm_segreduce_fwd", ptr(x), ptr(T), ptr(plan.buf), N, plan.E, plan.G, d, 0.1, 5, 2, compact, ptr(A), stream())
        call("pm_segreduce_fwd_planes", ptr(x), ptr(T), ptr(plan.buf), N, plan.E, plan.G, d, 0.1, 5, 2, compact, ptr(P),
             N * nb * d, stream())
        assert torch.equal(_planes_value(P).float().view(N, nb * d), A)        # the split is exact


def test_bn_eval_uses_running_stats():
    x = torch.randn(300, 64, device=DEV)
    rm, rv = torch.randn(64, device=DEV), torch.rand(64, device=DEV) + 0.5
    g, be = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV)
    y = ops.bn_apply(x, 300, 64, 1, rm, rv, g, be)
    assert rel_err(y, F.batch_norm(x, rm, rv, g, be, False)) < 1e-6


def test_elementwise_helpers():
    a, b = torch.randn(1000, 33, device=DEV), torch.randn(1000, 33, device=DEV)
    assert torch.equal(ops.add(a, b), a + b)
    assert torch.equal(ops.relu_bwd(a, b), a * (b > 0))
    out = torch.ones(33, device=DEV)
    ops.colsum_acc(a, 1000, 33, 33, out)
    assert rel_err(out, 1 + a.double().sum(0)) < 1e-5
    rows = torch.tensor([5, 9, 60, 61, 3], dtype=torch.int32, device=DEV)
    cnt = torch.tensor([4], dtype=torch.int32, device=DEV)
    out = torch.zeros(20, device=DEV)
    ops.colsum_rows_acc(a.view(-1)[7:], 20, 33, rows, 10, cnt, 5, out)            # 100 entries of 10 rows, offset view
    want = a.view(100, 10, 33)[rows[:4].long(), :, 7:27].double().sum((0, 1))
    assert rel_err(out, want) < 1e-5
    mu, lv, eps = torch.randn(64, 32, device=DEV), torch.randn(64, 32, device=DEV), torch.randn(64, 32, device=DEV)
    z = ops.reparam_fwd(mu, lv, eps)
    assert rel_err(z, torch.exp(0.5 * lv) * eps + mu) < 1e-6
    dz = torch.randn_like(z)
    dmu, dlv = torch.zeros_like(mu), torch.zeros_like(lv)
    ops.reparam_bwd(dz, lv, eps, dmu, dlv)
    assert rel_err(dmu, dz) == 0 and rel_err(dlv, dz * eps * 0.5 * torch.exp(0.5 * lv)) < 1e-6


# ------------------------------------------------------------------ embedding front
def embed_ref(P, b, d, training, rstats):
    """ContentEncoder embeddings (model.py:352-377) on one-hot inputs, in fp64 torch."""
    c = b.c_tensor[:, 1:, :].double()
    out = torch.zeros(c.shape[0], 15, d, dtype=torch.float64, device=c.device)
    for grp, (mask, pk, bk) in enumerate(((b.is_drum, "pd", "d"), (~b.is_drum, "pn", "n"))):
        t = c[mask]
        if t.shape[0] == 0:
            continue
        pe = F.linear(t[..., :131], P["w_" + pk], P["b_" + pk]).view(-1, d // 2)
        pe = F.batch_norm(pe, rstats["rm_" + bk], rstats["rv_" + bk], P["g_" + bk], P["be_" + bk], training, 0.1, 1e-5)
        de = F.linear(t[..., 131:], P["w_du"], P["b_du"]).view(-1, d // 2)
        de = F.batch_norm(de, rstats["rm_u"], rstats["rv_u"], P["g_u"], P["be_u"], training, 0.1, 1e-5)
        out[mask] = torch.cat((pe.view(-1, 15, d // 2), de.view(-1, 15, d // 2)), -1)
    return out


@pytest.mark.parametrize("d,training", [(32, True), (256, True), (64, False)])
def test_embed_fwd_bwd(small, d, training):
    b, plan = small
    torch.manual_seed(d)
    dh = d // 2
    mk = lambda *s: torch.randn(*s, device=DEV)
    P32 = dict(w_pd=mk(dh, 131) * .3, b_pd=mk(dh) * .1, w_pn=mk(dh, 131) * .3, b_pn=mk(dh) * .1, w_du=mk(dh, 99) * .3,
               b_du=mk(dh) * .1, g_d=torch.rand(dh, device=DEV) + .5, be_d=mk(dh) * .2, g_n=torch.rand(dh, device=DEV) + .5,
               be_n=mk(dh) * .2, g_u=torch.rand(dh, device=DEV) + .5, be_u=mk(dh) * .2)
    R32 = {k: (torch.rand(dh, device=DEV) + .5 if k.startswith("rv") else mk(dh) * .1)
           for k in ("rm_d", "rv_d", "rm_n", "rv_n", "rm_u", "rv_u")}
    P64 = {k: v.double().requires_grad_(True) for k, v in P32.items()}
    R64 = {k: v.double().clone() for k, v in R32.items()}
    tables = torch.empty(4, 131, dh, device=DEV)
    stats = torch.empty(4, 2, dh, device=DEV)
    call("pm_embed_tables", *[ptr(P32[k]) for k in ("w_pd", "b_pd", "w_pn", "b_pn", "w_du", "b_du", "g_d", "be_d", "g_n",
                                                    "be_n", "g_u", "be_u")],
         *[ptr(R32[k]) for k in ("rm_d", "rv_d", "rm_n", "rv_n", "rm_u", "rv_u")], ptr(plan.tok_hist), d, int(training),
         1e-5, 0.1, ptr(tables), ptr(stats), stream())
    X = torch.empty(plan.N, 15, d, device=DEV)
    call("pm_embed_gather", ptr(tables), ptr(plan.tokens), ptr(plan.is_drum), plan.N, d, 15, ptr(X), stream())
    ref = embed_ref(P64, b, d, training, R64)
    assert rel_err(X, ref.detach()) < 1e-5
    for k in R32:
        assert rel_err(R32[k], R64[k]) < 1e-5, k
    if not training:
        return
    dX = torch.randn_like(X)
    ref.backward(dX.double())
    S = torch.empty(4, 131, dh, device=DEV)
    call("pm_embed_bwd_scatter", ptr(dX), ptr(plan.tokens), ptr(plan.buf), plan.N, plan.E, plan.G, d, 15, ptr(S), stream())
    G32 = {k: torch.zeros_like(v) for k, v in P32.items()}
    call("pm_embed_tables_bwd", ptr(S), *[ptr(P32[k]) for k in ("w_pd", "b_pd", "w_pn", "b_pn", "w_du", "b_du", "g_d",
                                                               "g_n", "g_u")], ptr(stats), ptr(plan.tok_hist), d, 1e-5,
         *[ptr(G32[k]) for k in ("w_pd", "b_pd", "w_pn", "b_pn", "w_du", "b_du", "g_d", "be_d", "g_n", "be_n", "g_u",
                                 "be_u")], stream())
    for k in P32:
        if k.startswith("b_"):      # a bias in front of batch-stat BN has an analytically zero gradient: compare absolutely
            assert float((G32[k].double() - P64[k].grad).abs().max()) < 1e-4 * float(P64["w_" + k[2:]].grad.abs().max()), k
        else:
            assert rel_err(G32[k], P64[k].grad) < 2e-5, k


@pytest.mark.parametrize("drums", ["none", "all"])
def test_chord_table_algebra_with_an_empty_node_group(drums):
    """chord.hip when one of the two node groups (drums / non-drums) is empty: its chunks of the one-hot sums exit, its rows of
    the token sums stay zero, the other group's results are those of the gather-and-product formulation."""
    cpu = synthetic_batch(5, 2, p=0.25, seed=11)
    cpu.is_drum[:] = drums == "all"
    b, plan = make_plan(cpu)
    d, S, N, dh = 64, 4, plan.N, 32
    torch.manual_seed(3)
    tables = torch.randn(4, 131, dh, device=DEV)
    Wc = torch.randn(d, 15 * d, device=DEV) / (15 * d) ** 0.5
    bc = torch.randn(d, device=DEV) * 0.1
    tok = plan.tokens.view(N, 16, 2).clone()
    tok[:, S + 1:, 0] = 130; tok[:, S + 1:, 1] = 98
    tok = tok.contiguous()
    g = 0 if drums == "all" else 1
    pit, dur = tok[:, 1:, 0].long(), tok[:, 1:, 1].long()
    X = torch.cat((tables.double()[g][pit], tables.double()[2 + g][dur]), -1).reshape(N, 15 * d)
    ref = torch.relu(X @ Wc.double().t() + bc.double())
    cvec, PT = torch.empty(2, d, device=DEV), torch.empty(2, S, 2, 131, d, device=DEV)
    call("pm_chord_tables_fwd", ptr(tables), ptr(Wc), d, S, ptr(PT), ptr(bc), ptr(cvec), stream())
    x0 = torch.empty(N, d, device=DEV)
    call("pm_chord_sum_fwd", ptr(PT), ptr(cvec), ptr(tok), ptr(plan.is_drum), N, d, S, ptr(x0), stream())
    assert rel_err(x0, ref) < 2e-6
    dy = torch.randn(N, d, device=DEV)
    Gt = torch.zeros(2, S, 2, 131, d, device=DEV)
    call("pm_chord_sum_bwd", ptr(dy), ptr(tok), ptr(plan.buf), N, plan.E, plan.G, d, S, ptr(Gt), stream())
    assert float(Gt[1 - g].abs().max()) == 0.0
    want = torch.zeros(131, d, dtype=torch.float64, device=DEV).index_add_(0, pit[:, 1], dy.double())
    assert rel_err(Gt[g, 1, 0], want) < 1e-6
    assert rel_err(Gt[g, :, 0].sum(1), dy.double().sum(0).expand(S, d)) < 1e-5     # every (slot, kind) adds up to the column sums


@pytest.mark.parametrize("d,S", [(32, 15), (256, 5), (128, 3), (256, 1), (512, 2)])
def test_chord_encoder_as_table_algebra(small, d, S):
    """chord.hip: x0 = relu(b + sum_s X[:, s] @ Wc_s^T) through projected tables and row lookups, and its backward (token
    sums of the output gradient on the matrix cores, weight / bias / table gradients from them), against the formulation
    they replace — gather X, product, fp64 autograd — on the same tables and tokens (slots >= S: the PAD tail, closed form)."""
    b, plan = small
    torch.manual_seed(100 * d + S)
    N, dh = plan.N, d // 2
    tables = torch.randn(4, 131, dh, device=DEV) * 0.7
    Wc = torch.randn(d, 15 * d, device=DEV) / (15 * d) ** 0.5 * 3
    bc = torch.randn(d, device=DEV) * 0.1
    tok = plan.tokens.view(N, 16, 2).clone()
    tok[:, S + 1:, 0] = 130; tok[:, S + 1:, 1] = 98                      # slots beyond S: PAD in every node
    tok = tok.contiguous()
    # ---- the replaced formulation, fp64
    grp = (~plan.is_drum.bool()).long()                                 # 0 drums, 1 non-drums
    T64 = tables.double().requires_grad_(True)
    W64, b64 = Wc.double().requires_grad_(True), bc.double().requires_grad_(True)
    pit, dur = tok[:, 1:, 0].long(), tok[:, 1:, 1].long()
    X = torch.cat((T64[grp[:, None], pit], T64[2 + grp[:, None], dur]), -1)       # [N, 15, d]
    pre = X.reshape(N, 15 * d) @ W64.t() + b64
    ref = torch.relu(pre)
    # ---- table form
    cvec = torch.empty(2, d, device=DEV)
    call("pm_chord_pad_vec", ptr(tables), ptr(Wc), ptr(bc), d, S, ptr(cvec), stream())
    PT = torch.full((2, S, 2, 131, d), float("nan"), device=DEV)
    cvec2 = torch.empty(2, d, device=DEV)
    call("pm_chord_tables_fwd", ptr(tables), ptr(Wc), d, S, ptr(PT), ptr(bc), ptr(cvec2), stream())
    assert rel_err(cvec2, cvec) < 1e-6                                  # (the same vector from the tables launch)
    x0 = torch.empty(N, d, device=DEV)
    call("pm_chord_sum_fwd", ptr(PT), ptr(cvec), ptr(tok), ptr(plan.is_drum), N, d, S, ptr(x0), stream())
    assert rel_err(x0, ref.detach()) < 2e-6
    # ---- backward
    dy = torch.randn(N, d, device=DEV) * (ref.detach() > 0).float()      # gradient of the pre-activation, ReLU mask applied
    pre.backward(dy.double())
    Gt = torch.zeros(2, S, 2, 131, d, device=DEV)                       # (the caller clears it)
    if d % 32 == 0:
        call("pm_chord_sum_bwd", ptr(dy), ptr(tok), ptr(plan.buf), N, plan.E, plan.G, d, S, ptr(Gt), stream())
        want = torch.zeros(2, S, 2, 131, d, dtype=torch.float64, device=DEV)
        for s_ in range(S):
            for kind, ids in ((0, pit), (1, dur)):
                w_ = torch.zeros(2 * 131, d, dtype=torch.float64, device=DEV)
                w_.index_add_(0, grp * 131 + ids[:, s_], dy.double())
                want[:, s_, kind] = w_.view(2, 131, d)
        assert rel_err(Gt, want) < 1e-6
    else:
        return
    dW = torch.full((d, 15 * d), 0.5, device=DEV)
    db = torch.full((d,), 0.25, device=DEV)
    Stab = torch.zeros(4, 131, dh, device=DEV)
    call("pm_chord_tables_bwd_w", ptr(Gt), ptr(tables), d, S, ptr(dW), ptr(db), stream())
    call("pm_chord_tables_bwd_x", ptr(Gt), ptr(Wc), d, S, ptr(Stab), stream())
    gsum = torch.empty(2, d, device=DEV)
    call("pm_chord_pad_bwd", ptr(dy), ptr(plan.is_drum), N, d, S, ptr(tables), ptr(Wc), ptr(gsum), ptr(dW), ptr(Stab), stream())
    assert rel_err(dW - 0.5, W64.grad) < 2e-5
    assert rel_err(db - 0.25, b64.grad) < 1e-5
    tg = T64.grad.clone()
    tg[2:, 99:] = 0                                                     # (rows of the duration tables beyond their vocabulary)
    Stab[2:, 99:] = 0
    assert rel_err(Stab, tg) < 2e-5


# ------------------------------------------------------------------ pooling / broadcast
@pytest.mark.parametrize("d", [32, 256])
def test_attention_pool_fwd_bwd(small, d):
    b, plan = small
    torch.manual_seed(d)
    N, G = plan.N, plan.G
    x = torch.randn(N, d, device=DEV)
    w, bb = torch.randn(d, device=DEV) * 0.2, torch.randn(1, device=DEV)
    bg, bbe = torch.rand(1, device=DEV) + 0.5, torch.randn(1, device=DEV)
    rm, rv = torch.zeros(1, device=DEV), torch.ones(1, device=DEV)
    g = ops.gate_fwd(x, w, bb)
    gm, gv = ops.bn_stats(g, N, 1, 1, rm, rv)
    alpha, out = ops.attnpool_fwd(x, g, gm, gv, bg, bbe, plan)
    seg = (b.bars + b.n_bars * b.batch)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), bb.double().requires_grad_(True)
    bgr, bber = bg.double().requires_grad_(True), bbe.double().requires_grad_(True)
    gr = F.batch_norm((xr @ wr + br).view(-1, 1), None, None, bgr, bber, True, 0.1, 1e-5)
    gmax = torch.full((G, 1), float("-inf"), dtype=torch.float64, device=DEV).scatter_reduce(0, seg.view(-1, 1), gr, "amax")
    e = (gr - gmax[seg]).exp()
    al = e / (torch.zeros(G, 1, dtype=torch.float64, device=DEV).index_add_(0, seg, e)[seg] + 1e-16)
    ref = torch.zeros(G, d, dtype=torch.float64, device=DEV).index_add_(0, seg, al * xr)
    assert rel_err(out, ref.detach()) < 1e-5 and rel_err(alpha, al.detach().flatten()) < 1e-5
    dout = torch.randn(G, d, device=DEV)
    ref.backward(dout.double())
    dw, db, dbg, dbb = (torch.zeros_like(t) for t in (w, bb, bg, bbe))
    dx = ops.attnpool_bwd(x, g, gm, gv, bg, alpha, dout, w, plan, dw, db, dbg, dbb)
    assert rel_err(dx, xr.grad) < 1e-5
    assert rel_err(dw, wr.grad) < 1e-4 and rel_err(dbg, bgr.grad) < 1e-4
    assert float((db.double() - br.grad).abs().max()) < 1e-4 and float((dbb.double() - bber.grad).abs().max()) < 1e-4


def test_bar_broadcast(small):
    b, plan = small
    bars = torch.randn(plan.G, 64, device=DEV)
    seg = (b.bars + b.n_bars * b.batch)
    x = ops.bar_broadcast_fwd(bars, plan)
    assert torch.equal(x, bars[seg])
    dx = torch.randn(plan.N, 64, device=DEV)
    assert rel_err(ops.bar_broadcast_bwd(dx, plan), torch.zeros(plan.G, 64, device=DEV, dtype=torch.float64).index_add_(0, seg, dx.double())) < 1e-6


# ------------------------------------------------------------------ structure CNN
@pytest.mark.parametrize("G", [13, 512])
@pytest.mark.parametrize("Ci,Co,H,W,up4", [(1, 8, 4, 32, False), (8, 16, 4, 8, False), (16, 8, 4, 32, True), (8, 1, 4, 32, False),
                                           (3, 5, 3, 12, False), (2, 3, 2, 16, True)])
def test_conv3x3(Ci, Co, H, W, up4, G):
    """The structure CNN's convolutions (model.py:219-230,279-285) against torch in fp64: the model's four shapes (round 6:
    position-per-thread kernels), two other shapes (the generic kernels), one workgroup and many."""
    torch.manual_seed(Ci * Co)
    x = torch.randn(G, Ci, H, W // 4 if up4 else W, device=DEV)
    w, bias = torch.randn(Co, Ci, 3, 3, device=DEV) * 0.3, torch.randn(Co, device=DEV)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), bias.double().requires_grad_(True)
    xin = F.interpolate(xr, scale_factor=(1, 4), mode="nearest") if up4 else xr
    ref = F.conv2d(xin, wr, br, padding=1)
    y = ops.conv3x3_fwd(x, w, bias, G, Ci, Co, H, W, up4)
    assert rel_err(y, ref.detach()) < 1e-6
    dy = torch.randn_like(y)
    ref.backward(dy.double())
    dx = ops.conv3x3_bwd_data(dy, w, G, Ci, Co, H, W, up4)
    assert rel_err(dx, xr.grad) < 1e-6
    dw, db = torch.zeros_like(w), torch.zeros_like(bias)
    ops.conv3x3_bwd_weight(x, dy, G, Ci, Co, H, W, dw, db, up4)
    assert rel_err(dw, wr.grad) < 1e-6 and rel_err(db, br.grad) < 1e-6


def test_maxpool4():
    x = torch.randn(7, 8, 4, 32, device=DEV)
    x[0, 0, 0, :8] = 0.0                                          # ties: first index wins
    xr = x.clone().requires_grad_(True)
    ref = F.max_pool2d(xr, (1, 4), stride=(1, 4))
    y = ops.maxpool4_fwd(x)
    assert torch.equal(y, ref.detach())
    dy = torch.randn_like(y)
    ref.backward(dy)
    assert torch.equal(ops.maxpool4_bwd(x, dy), xr.grad)


# ------------------------------------------------------------------ loss / optimiser
def test_losses_match_reference_formulas(small):
    b, plan = small
    torch.manual_seed(0)
    N = plan.N
    logits = torch.randn(N, 15, 230, device=DEV) * 2
    out, dl = ops.content_ce(logits, plan, grad_scale=1.0)
    lr = logits.double().requires_grad_(True)
    tgt = b.tokens[:, 1:, :].long().reshape(-1, 2)
    lp = F.cross_entropy(lr.view(-1, 230)[:, :131], tgt[:, 0], ignore_index=130)
    ld = F.cross_entropy(lr.view(-1, 230)[:, 131:], tgt[:, 1], ignore_index=98)
    (lp + ld).backward()
    o = out.cpu()
    assert abs(o[0] - lp.item()) < 1e-6 and abs(o[1] - ld.item()) < 1e-6
    assert rel_err(dl, lr.grad) < 1e-5
    mu, lv = torch.randn(16, 32, device=DEV), torch.randn(16, 32, device=DEV) * 0.3
    dmu, dlv = torch.zeros_like(mu), torch.zeros_like(lv)
    ops.kld(mu, lv, out, beta=0.5, dmu=dmu, dlog_var=dlv)
    mr, lvr = mu.double().requires_grad_(True), lv.double().requires_grad_(True)
    k = (-0.5 * torch.sum(1 + lvr - mr.pow(2) - lvr.exp(), dim=1)).mean()
    (0.5 * k).backward()
    assert abs(out.cpu()[3] - k.item()) < 1e-6 * max(1, abs(k.item()))
    assert rel_err(dmu, mr.grad) < 1e-5 and rel_err(dlv, lvr.grad) < 1e-5
    s_log, tgt_s = torch.randn(12, 4, 32, device=DEV), (torch.rand(12, 4, 32, device=DEV) < 0.25).float()
    out, ds = ops.bce_logits(s_log, tgt_s, out, want_grad=True)
    sr = s_log.double().requires_grad_(True)
    bl = F.binary_cross_entropy_with_logits(sr, tgt_s.double())
    bl.backward()
    assert abs(out.cpu()[2] - bl.item()) < 1e-6 and rel_err(ds, sr.grad) < 1e-5


@pytest.mark.parametrize("n", [1000, 100003])
def test_adam_matches_torch(n):
    torch.manual_seed(n)
    p = torch.randn(n, device=DEV)
    pref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pref], lr=5e-6, betas=(0.9, 0.98), eps=1e-9)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in (1, 2, 3):
        g = torch.randn(n, device=DEV)
        g[::7] = 0.0
        lr = 5e-6 if step == 1 else 1e-4
        for pg in opt.param_groups:
            pg["lr"] = lr
        pref.grad = g.clone()
        opt.step()
        ops.adam_step(p, g, m, v, lr, 0.9, 0.98, 1e-9, step)
        assert rel_err(p, pref.detach()) < 1e-6
    assert float((p - pref.detach()).abs().max()) < 2.5e-7


# ------------------------------------------------------------------ evaluation metrics
def test_accuracy_and_structure_metric_kernels():
    """integer counts of `_accuracies` (training.py:349-497) against a torch restatement, bit-exact."""
    torch.manual_seed(5)
    N = 700
    tok = torch.stack([torch.randint(0, 131, (N, 16)), torch.randint(0, 99, (N, 16))], -1).to(torch.int32)
    tok[:, 6:, 0], tok[:, 6:, 1] = 130, 98                                   # PAD tail
    tok[::7, 3:, 0], tok[::7, 3:, 1] = 130, 98
    logits = torch.randn(N, 15, 230)
    hit_p, hit_d = torch.rand(N, 15) < 0.6, torch.rand(N, 15) < 0.5        # plant correct predictions
    tp, td = tok[:, 1:, 0].long(), tok[:, 1:, 1].long()
    logits.scatter_add_(2, tp.unsqueeze(-1), 9.0 * hit_p.unsqueeze(-1).float())
    logits.scatter_add_(2, (131 + td).unsqueeze(-1), 9.0 * hit_d.unsqueeze(-1).float())
    logits[5, 2, 17] = logits[5, 2, 40] = 50.0                               # a tie: first index wins (torch.argmax)
    drum = torch.rand(N) < 0.3
    got = ops.content_accuracy(logits.to(DEV), tok.to(DEV), drum.to(DEV)).cpu()
    pr, dr = logits[..., :131].argmax(-1), logits[..., 131:].argmax(-1)
    np_, nd_ = tp != 130, td != 98
    cp, cd = (pr == tp) & np_, (dr == td) & nd_
    want = [cp.sum(), np_.sum(), cp[drum].sum(), np_[drum].sum(), cd.sum(), nd_.sum(), (cp & cd).sum(), 0]
    assert got.tolist() == [int(v) for v in want]
    s_log = torch.randn(64, 4, 32)
    s_log.view(-1)[:5] = torch.tensor([0.0, -0.0, 1e-3, -1e-3, 30.0])
    s_tgt = (torch.rand(64, 4, 32) < 0.2).float()
    m = ops.structure_metrics(s_log.to(DEV).contiguous(), s_tgt.to(DEV).contiguous()).cpu().tolist()
    pred = torch.sigmoid(s_log) >= 0.5
    t = s_tgt.bool()
    assert m == [int((pred == t).sum()), int((pred & t).sum()), int(pred.sum()), int(t.sum())]


def test_plan_track_lists_are_class_sorted_and_gemm_skips_zero_blocks():
    """PM_PLAN_TRK_LIST / TRK_CNT: per track relation the nodes sorted by (receives onset edges, receives next edges) in
    the order (0,0) (1,0) (1,1) (0,1) with boundaries b0..b4, ascending inside a class; and the three GCL contractions
    with `class_ptr` (all-zero onset / next blocks skipped tile by tile) equal the unmasked ones wherever defined."""
    cpu = synthetic_batch(24, 2, p=0.2, seed=13)
    b, plan = make_plan(cpu)
    N = cpu.num_nodes
    dst, et = cpu.edge_index[1].numpy(), cpu.edge_type.numpy()
    on, nx = np.zeros(N, bool), np.zeros(N, bool)
    on[dst[et == 4]] = True
    nx[dst[et == 5]] = True
    trel = np.zeros(N, np.int64)
    for r in range(4):
        trel[dst[et == r]] = r
    cnt = plan.field("trk_cnt").cpu().numpy()
    lists = plan.field("trk_list").cpu().numpy()[:4 * N].reshape(4, N)
    gray = {(False, False): 0, (True, False): 1, (True, True): 2, (False, True): 3}
    cls = np.array([gray[(bool(a), bool(c))] for a, c in zip(on, nx)])
    for t in range(4):
        bnd = cnt[8 + 5 * t: 8 + 5 * t + 5]
        assert bnd[0] == 0 and bnd[4] == cnt[t] == int((trel == t).sum())
        for k in range(4):
            want = np.nonzero((trel == t) & (cls == k))[0]
            np.testing.assert_array_equal(lists[t, bnd[k]:bnd[k + 1]], want)
    # behind the counters: the tile schedule of the GCL products, as csrc/tile_order.h derives it from them (the host copy)
    from polyphemus_amd._lib import gcl_tile_order
    for cpu2 in (cpu, synthetic_batch(256, 2, p=0.25, seed=1235)):        # (the second one: 258 tiles, two of them split)
        _, plan2 = make_plan(cpu2)
        tc2 = plan2.field("trk_cnt").cpu().numpy()
        want = gcl_tile_order(tc2[:32], True, cpu2.num_nodes)
        got = tc2[32:32 + 4 * len(want)].reshape(-1, 4)
        np.testing.assert_array_equal(got[:, :3], np.array(want))
        assert len(tc2) == 32 + 4 * len(want)
    # masked contractions == unmasked ones (d = 64: one 64-wide tile per block column; d = 128 also with the weights
    # as fragment-major planes: the B-direct tiles are 128 wide)
    for d, bfrag in ((64, False), (128, False), (128, True)):
        _masked_contractions(plan, N, on, nx, d, bfrag)


def _masked_contractions(plan, N, on, nx, d, bfrag):
    dd = d * d
    torch.manual_seed(0)
    A = torch.randn(N, 4 * d, device=DEV)
    nodes = torch.arange(N)
    A[torch.from_numpy(~on), d:2 * d] = 0                           # the aggregate's zero blocks
    A[torch.from_numpy(~nx), 2 * d:3 * d] = 0
    W = torch.randn(7 * d, d, device=DEV)
    dh = torch.randn(N, d, device=DEV)
    Ap, Wp, dhp = ops.split_planes(A), ops.split_planes(W), ops.split_planes(dh)
    fn = dict(b_frag=ops.split_planes_frag(W, 1)) if bfrag else {}
    ft = dict(b_frag=ops.split_planes_frag(W, 0)) if bfrag else {}
    tl = plan.field("trk_list")
    tc = plan.field("trk_cnt")
    grp = dict(rowmap=tl, rows_per_entry=1, dyn_entries=tc, n_groups=4, map_group_stride=N, dyn_group_stride=1,
               partition=True, planes=True)
    res = []
    for masked in (False, True):
        ck = dict(class_ptr=tc[8:], class_block=d) if masked else {}
        h = torch.zeros(N, d, device=DEV)
        ops.gemm_desc(Ap, Wp, h, N, d, 4 * d, 4 * d, d, d, b_group_stride=dd, b_split_rows=d, b_shared_off=3 * dd,
                      a_plane_stride=A.numel(), b_plane_stride=W.numel(), **grp, **ck, **fn)
        dA = torch.zeros(N, 4 * d, device=DEV)
        ops.gemm_desc(dhp, Wp, dA, N, 4 * d, d, d, d, 4 * d, transB=True, b_group_stride=dd, b_split_rows=d,
                      b_shared_off=3 * dd, a_plane_stride=dh.numel(), b_plane_stride=W.numel(), **grp, **ck, **ft)
        dW = torch.zeros(7 * d, d, device=DEV)
        ops.gemm_desc(Ap, dhp, dW, 4 * d, d, N, 4 * d, d, d, transA=True, accum=True, split_k=0, c_group_stride=dd,
                      c_split_rows=d, c_shared_off=3 * dd, a_plane_stride=A.numel(), b_plane_stride=dh.numel(), **grp, **ck)
        res.append((h, dA, dW))
    (h0, dA0, dW0), (h1, dA1, dW1) = res
    assert rel_err(h1, h0) < 1e-6 and rel_err(dW1, dW0) < 1e-6
    trel_t = plan.field("node_trel").long()[:N]
    Wn = torch.stack([torch.cat([W[t * d:(t + 1) * d], W[4 * d:]]) for t in range(4)]).double()
    assert rel_err(h1, torch.einsum("nk,nkj->nj", A.double(), Wn[trel_t])) < 5e-6
    keep = torch.ones(N, 4 * d, dtype=torch.bool, device=DEV)      # dA: only the blocks a node's edges can read are defined
    keep[torch.from_numpy(~on), d:2 * d] = False
    keep[torch.from_numpy(~nx), 2 * d:3 * d] = False
    assert rel_err(dA1[keep], dA0[keep]) < 1e-6


@pytest.mark.parametrize("d,p,dense", [(128, 0.0, False), (256, 0.0, False), (256, 0.15, False), (128, 0.1, True),
                                       (256, 0.0, True), (128, 0.1, "tiny"), (256, 0.0, "tiny"),
                                       (512, 0.0, False), (512, 0.1, False), (512, 0.1, True), (512, 0.0, "tiny"),
                                       (256, 0.1, "split"), (512, 0.1, "split")])
def test_gcl_forward_fused_equals_segreduce_plus_product(d, p, dense):
    """`pm_gcl_forward_fused` (aggregate built in LDS, contracted in the same kernel) against the unfused pair it
    replaces — `pm_segreduce_fwd_planes` then the grouped planes product with row classes: same edge order and message
    arithmetic, so the A' planes it leaves for the backward are BIT-identical to the pair's wherever the backward reads
    them; h agrees to fp32 accumulation order (the fused kernel contracts the self block first) and against an fp64
    product of the exact planes; the BatchNorm column sums agree with its own h.  dense: up to 127 edges per node (the edge
    lists overflow the LDS cache and the in-flight gather: the serial tail paths)."""
    if dense == "tiny":                # one sparse sample: a few dozen nodes, tiles far from full, class ranges partly empty
        cpu = synthetic_batch(1, 2, p=0.12, seed=41)
    elif dense == "split":             # 258 row tiles for 256 CUs: two tiles run as 32-row halves (csrc/tile_order.h)
        cpu, dense = synthetic_batch(256, 2, p=0.25, seed=1235), False
    else:
        cpu = synthetic_batch(3 if dense else 40, 2, p=0.3, seed=17, dense=dense)
    assert cpu.track_unique
    b, plan = make_plan(cpu)
    N, dd = cpu.num_nodes, d * d
    torch.manual_seed(3)
    x = torch.randn(N, d, device=DEV)
    T = ops.edge_table(torch.randn(d, 32, device=DEV) * 0.5, torch.randn(d, device=DEV) * 0.1)
    W = torch.randn(7 * d, d, device=DEV) / d ** 0.5
    bias = torch.randn(d, device=DEV)
    Wp, Wf = ops.split_planes(W), ops.split_planes_frag(W, 1)
    tl, tc = plan.field("trk_list"), plan.field("trk_cnt")
    # unfused pair
    P0 = torch.zeros(3, N * 4 * d, dtype=torch.int16, device=DEV)
    call("pm_segreduce_fwd_planes", ptr(x), ptr(T), ptr(plan.buf), N, plan.E, plan.G, d, p, 5, 2, 1, ptr(P0), N * 4 * d,
         stream())
    h0 = torch.zeros(N, d, device=DEV)
    s0 = torch.zeros(8, 2, d, dtype=torch.float64, device=DEV)
    ops.gemm_desc(P0, Wp, h0, N, d, 4 * d, 4 * d, d, d, bias=bias, b_group_stride=dd, b_split_rows=d, b_shared_off=3 * dd,
                  a_plane_stride=N * 4 * d, b_plane_stride=W.numel(), rowmap=tl, rows_per_entry=1, dyn_entries=tc,
                  n_groups=4, map_group_stride=N, dyn_group_stride=1, partition=True, planes=True, class_ptr=tc[8:],
                  class_block=d, b_frag=Wf, col_stats=s0)
    # fused
    P1 = torch.full((3, N * 4 * d), 0x7fc0, dtype=torch.int16, device=DEV)       # NaN pattern: unwritten blocks show
    s1 = torch.zeros(8, 2, d, dtype=torch.float64, device=DEV)
    h1 = ops.gcl_forward_fused(x, T, plan, p, 5, 2, Wf, bias, col_stats=s1, planes=P1)
    assert rel_err(h1, h0) < 5e-6
    A64 = _planes_value(P0).double().view(N, 4 * d)
    trel_t = plan.field("node_trel").long()[:N]
    Wn = torch.stack([torch.cat([W[t * d:(t + 1) * d], W[4 * d:]]) for t in range(4)]).double()
    want = torch.einsum("nk,nkj->nj", A64, Wn[trel_t]) + bias.double()
    assert rel_err(h1, want) < 5e-6 and rel_err(h0, want) < 5e-6
    assert rel_err(s1.sum(0)[0], h1.double().sum(0)) < 1e-12 and rel_err(s1.sum(0)[1], (h1.double() ** 2).sum(0)) < 1e-12
    # A' planes: track and self blocks everywhere; onset / next blocks for the rows that receive such edges
    dst, et = cpu.edge_index[1], cpu.edge_type
    on, nx = torch.zeros(N, dtype=torch.bool), torch.zeros(N, dtype=torch.bool)
    on[dst[et == 4]] = True
    nx[dst[et == 5]] = True
    keep = torch.ones(N, 4, d, dtype=torch.bool)
    keep[~on, 1] = False
    keep[~nx, 2] = False
    keep = keep.view(-1).to(DEV)
    for k in range(3):
        assert torch.equal(P1[k][keep], P0[k][keep])
    # without the planes output and without row classes: same h
    h2 = ops.gcl_forward_fused(x, T, plan, p, 5, 2, Wf, bias, use_classes=False)
    assert torch.equal(h2, h1)
    if d == 512:          # the product with the aggregate read from the pair's planes (the dense-graph path): same chain
        s3 = torch.zeros(8, 2, d, dtype=torch.float64, device=DEV)
        h3 = ops.gcl_forward_from_planes(P0, plan, d, Wf, bias, col_stats=s3)
        assert torch.equal(h3, h1)
        assert rel_err(s3.sum(0)[0], h1.double().sum(0)) < 1e-12


@pytest.mark.parametrize("d,B", [(128, 40), (256, 40), (256, 1), (512, 40), (512, 1), (256, "split"), (512, "split")])
def test_gcl_input_grad_fused_equals_grouped_product(d, B):
    """`pm_gcl_input_grad_fused` (dh rows resident in LDS, all 4d output columns per workgroup) against the grouped planes
    product with transB it replaces: same six products in the same k order -> BIT-identical wherever the segment-reduce
    backward reads dA' (track and self blocks of every row, onset / next blocks of the rows that receive such edges)."""
    if B == "split":                   # 258 row tiles for 256 CUs: two tiles run as 32-row halves (csrc/tile_order.h)
        cpu = synthetic_batch(256, 2, p=0.25, seed=1235)
    else:
        cpu = synthetic_batch(B, 2, p=0.3 if B > 1 else 0.12, seed=19)
    b, plan = make_plan(cpu)
    N, dd = cpu.num_nodes, d * d
    torch.manual_seed(4)
    dh = torch.randn(N, d, device=DEV)
    W = torch.randn(7 * d, d, device=DEV) / d ** 0.5
    dhp, Wp, Wft = ops.split_planes(dh), ops.split_planes(W), ops.split_planes_frag(W, 0)
    tl, tc = plan.field("trk_list"), plan.field("trk_cnt")
    dA0 = torch.zeros(N, 4 * d, device=DEV)
    ops.gemm_desc(dhp, Wp, dA0, N, 4 * d, d, d, d, 4 * d, transB=True, b_group_stride=dd, b_split_rows=d,
                  b_shared_off=3 * dd, a_plane_stride=dh.numel(), b_plane_stride=W.numel(), rowmap=tl, rows_per_entry=1,
                  dyn_entries=tc, n_groups=4, map_group_stride=N, dyn_group_stride=1, partition=True, planes=True,
                  class_ptr=tc[8:], class_block=d, b_frag=Wft)
    dA1 = ops.gcl_input_grad_fused(dhp, plan, d, Wft, out=torch.full((N, 4 * d), float("nan"), device=DEV))
    dst, et = cpu.edge_index[1], cpu.edge_type
    on, nx = torch.zeros(N, dtype=torch.bool), torch.zeros(N, dtype=torch.bool)
    on[dst[et == 4]] = True
    nx[dst[et == 5]] = True
    keep = torch.ones(N, 4, d, dtype=torch.bool)
    keep[~on, 1] = False
    keep[~nx, 2] = False
    keep = keep.view(N, 4 * d).to(DEV)
    assert torch.equal(dA1[keep], dA0[keep])
    trel_t = plan.field("node_trel").long()[:N]
    Wn = torch.stack([torch.cat([W[t * d:(t + 1) * d], W[4 * d:]]) for t in range(4)]).double()       # [4, 4d, d]
    want = torch.einsum("nk,njk->nj", dh.double(), Wn[trel_t])
    assert rel_err(dA1[keep], want[keep]) < 2e-6
    dA2 = ops.gcl_input_grad_fused(dhp, plan, d, Wft, use_classes=False)
    assert torch.equal(dA2[keep], dA1[keep]) and bool(torch.isfinite(dA2).all())


@pytest.mark.parametrize("d,B", [(128, 40), (256, 40), (256, 1), (256, "split")])
def test_gcl_input_grad_with_the_norm_backward_inside_equals_the_two_calls(d, B):
    """`pm_gcl_input_grad_bn` (the BatchNorm backward in the prologue of the input-gradient kernel) against
    `pm_bn_bwd_fused` (planes out) followed by `pm_gcl_input_grad_fused`: the same dh planes, dA', dgamma, dbeta and bias
    gradient, bit for bit, with the column sums of `pm_bn_bwd_sums`; and against an fp64 statement of the norm backward."""
    from polyphemus_amd._lib import call, ptr, stream
    if B == "split":
        cpu = synthetic_batch(256, 2, p=0.25, seed=1235)
    else:
        cpu = synthetic_batch(B, 2, p=0.3 if B > 1 else 0.12, seed=19)
    b, plan = make_plan(cpu)
    N = cpu.num_nodes
    torch.manual_seed(6)
    h = torch.randn(N, d, device=DEV) * 1.5 + 0.3
    du = torch.randn(N, d, device=DEV)
    gamma, beta = torch.rand(d, device=DEV) + 0.5, torch.randn(d, device=DEV) * 0.2
    mean, var = h.mean(0), h.var(0, unbiased=False)
    W = torch.randn(7 * d, d, device=DEV) / d ** 0.5
    Wft = ops.split_planes_frag(W, 0)
    acc3 = ops.bn_bwd_sums(h, du, mean, var, gamma, beta)
    # the two calls
    g0 = [torch.full((d,), 0.25, device=DEV) for _ in range(3)]
    planes0 = torch.zeros(3, N * d, dtype=torch.int16, device=DEV)
    call("pm_bn_bwd_fused", ptr(h), ptr(du), N, d, ptr(mean), ptr(var), 1e-5, ptr(gamma), ptr(beta), 1, ptr(g0[0]), ptr(g0[1]),
         ptr(g0[2]), None, ptr(acc3), ptr(planes0), N * d, 1, stream())
    dA0 = ops.gcl_input_grad_fused(planes0, plan, d, Wft)
    # the one call
    g1 = [torch.full((d,), 0.25, device=DEV) for _ in range(3)]
    dA1, planes1 = ops.gcl_input_grad_bn(h, du, mean, var, gamma, beta, acc3, plan, Wft, dgamma=g1[0], dbeta=g1[1], dbias_pre=g1[2])
    assert torch.equal(planes1, planes0)
    dst, et = cpu.edge_index[1], cpu.edge_type
    on, nx = torch.zeros(N, dtype=torch.bool), torch.zeros(N, dtype=torch.bool)
    on[dst[et == 4]] = True
    nx[dst[et == 5]] = True
    keep = torch.ones(N, 4, d, dtype=torch.bool)
    keep[~on, 1] = False
    keep[~nx, 2] = False
    keep = keep.view(N, 4 * d).to(DEV)
    assert torch.equal(dA1[keep], dA0[keep])
    for a, b_ in zip(g1, g0):
        assert torch.equal(a, b_)
    # ... with the residual gradient riding in the self block: dA'[:, 3d:] + du, everything else unchanged
    dA2, planes2 = ops.gcl_input_grad_bn(h, du, mean, var, gamma, beta, acc3, plan, Wft, add_residual=True)
    assert torch.equal(planes2, planes0)
    assert torch.equal(dA2[:, 3 * d:], dA0[:, 3 * d:] + du)
    k3 = keep[:, :3 * d]
    assert torch.equal(dA2[:, :3 * d][k3], dA0[:, :3 * d][k3])
    # fp64 statement of the norm backward (autograd of relu(BN(h)) w.r.t. h)
    hd = h.double().requires_grad_(True)
    xh = (hd - hd.mean(0)) / torch.sqrt(hd.var(0, unbiased=False) + 1e-5)
    torch.relu(xh * gamma.double() + beta.double()).backward(du.double())
    got = sum(planes1[k].view(torch.bfloat16).double() for k in range(3)).view(N, d)
    assert rel_err(got, hd.grad) < 5e-6
    assert rel_err(g1[1] - 0.25, (du.double() * (xh * gamma.double() + beta.double() > 0)).sum(0)) < 1e-5


@pytest.mark.parametrize("d,B", [(128, 40), (256, 40), (256, 3), (128, 1), (512, 40), (512, 1)])
def test_gcl_weight_grad_fused_equals_grouped_product(d, B):
    """`pm_gcl_weight_grad_fused` (128x128 tiles, loader waves + LDS ring, K slices by atomics) against the grouped planes
    product with transA it replaces and against an fp64 contraction of the exact planes; accumulates into dW (+=)."""
    cpu = synthetic_batch(B, 2, p=0.3 if B > 1 else 0.12, seed=23)
    b, plan = make_plan(cpu)
    N, dd = cpu.num_nodes, d * d
    torch.manual_seed(5)
    A = torch.randn(N, 4 * d, device=DEV)
    dst, et = cpu.edge_index[1], cpu.edge_type
    on, nx = torch.zeros(N, dtype=torch.bool), torch.zeros(N, dtype=torch.bool)
    on[dst[et == 4]] = True
    nx[dst[et == 5]] = True
    A[(~on).to(DEV), d:2 * d] = 0                                     # the aggregate's zero blocks
    A[(~nx).to(DEV), 2 * d:3 * d] = 0
    dh = torch.randn(N, d, device=DEV)
    Ap, dhp = ops.split_planes(A), ops.split_planes(dh)
    tl, tc = plan.field("trk_list"), plan.field("trk_cnt")
    base = torch.randn(7 * d, d, device=DEV)
    dW0 = base.clone()
    ops.gemm_desc(Ap, dhp, dW0, 4 * d, d, N, 4 * d, d, d, transA=True, accum=True, split_k=0, c_group_stride=dd,
                  c_split_rows=d, c_shared_off=3 * dd, a_plane_stride=A.numel(), b_plane_stride=dh.numel(), rowmap=tl,
                  rows_per_entry=1, dyn_entries=tc, n_groups=4, map_group_stride=N, dyn_group_stride=1, partition=True,
                  planes=True, class_ptr=tc[8:], class_block=d)
    dW1 = ops.gcl_weight_grad_fused(Ap, dhp, plan, d, base.clone())
    trel = plan.field("node_trel").long()[:N]
    want = base.double().clone()
    for t in range(4):
        rows = trel == t
        want[t * d:(t + 1) * d] += A[rows, :d].double().T @ dh[rows].double()
    want[4 * d:] += A[:, d:].double().T @ dh.double()
    scale = float((want - base.double()).abs().max())
    assert float((dW1.double() - want).abs().max()) < 2e-6 * scale
    assert float((dW0.double() - want).abs().max()) < 2e-6 * scale
    dW2 = ops.gcl_weight_grad_fused(Ap, dhp, plan, d, base.clone(), use_classes=False)
    assert float((dW2.double() - want).abs().max()) < 2e-6 * scale


@pytest.mark.parametrize("K,Nout,N,kind", [(256, 1280, 16271, 0), (256, 1280, 5000, 1), (128, 384, 77, 0), (128, 256, 64, 1),
                                           (512, 2560, 16271, 0), (512, 1536, 777, 1), (512, 512, 64, 0),
                                           # 257 / 259 tiles for 256 CUs: one / three tiles run as 32-row halves (tile_order.h)
                                           (256, 1280, 16417, 1), (256, 512, 16550, 0), (512, 1024, 16417, 0)])
def test_rows_times_weight_matches_fp64(K, Nout, N, kind):
    """`pm_rows_times_weight` (A-stationary: the fp32 rows of a 64-row tile split into bf16 planes once, all output columns
    from that LDS image) for both weight orientations: y = x W^T + b (kind 0, W [Nout, K]) and y = x W[:, :Nout] (kind 1,
    W [K, ldw]); fp32-exact products against an fp64 product; strided x and y; rows past N untouched."""
    torch.manual_seed(9)
    ldx, ldc = K + 8, Nout + 4
    X = torch.randn(N, ldx, device=DEV)
    bias = torch.randn(Nout, device=DEV) if kind == 0 else None
    if kind == 0:
        W = torch.randn(Nout, K, device=DEV) / K ** 0.5
        want = X[:, :K].double() @ W.double().T + bias.double()
        tiles = 0
    else:
        ldw = Nout + 3 * K                                             # the product uses the first Nout columns of W [K, ldw]
        W = torch.randn(K, ldw, device=DEV) / K ** 0.5
        want = X[:, :K].double() @ W[:, :Nout].double()
        tiles = ldw // 32
    Wf = ops.split_planes_frag(W, kind)
    C = torch.full((N, ldc), 7.0, device=DEV)
    call("pm_rows_times_weight", ptr(X), ldx, N, K, ptr(Wf), kind, tiles, Nout, ptr(bias), ptr(C), ldc, stream())
    assert rel_err(C[:, :Nout], want) < 2e-6
    assert bool((C[:, Nout:] == 7.0).all())


@pytest.mark.parametrize("K,Nout,N,kind", [(1280, 256, 16271, 0), (1280, 256, 5000, 1), (384, 128, 77, 0), (128, 128, 64, 1),
                                           (2560, 512, 16271, 0), (1536, 512, 777, 1), (128, 512, 64, 1),
                                           (1280, 256, 16417, 1), (512, 256, 16550, 0), (1024, 512, 16417, 0)])
def test_rows_times_weight_longk_matches_fp64(K, Nout, N, kind):
    """`pm_rows_times_weight_longk` (producer waves split 64 x 128 fp32 chunks into bf16 planes in an LDS ring, MFMA waves
    contract them): y = x W[:, :K]^T (kind 0, W [Nout, ldw]) and y = x W (kind 1, W [K, Nout]) against an fp64 product."""
    torch.manual_seed(10)
    ldx, ldc = K + 4, Nout + 4
    X = torch.randn(N, ldx, device=DEV)
    if kind == 0:
        ldw = K + 256                                                  # the product uses the first K columns of W [Nout, ldw]
        W = torch.randn(Nout, ldw, device=DEV) / K ** 0.5
        want = X[:, :K].double() @ W[:, :K].double().T
        pitch = ldw // 16
    else:
        W = torch.randn(K, Nout, device=DEV) / K ** 0.5
        want = X[:, :K].double() @ W.double()
        pitch = 0
    Wf = ops.split_planes_frag(W, kind)
    C = torch.full((N, ldc), 7.0, device=DEV)
    call("pm_rows_times_weight_longk", ptr(X), ldx, N, K, ptr(Wf), kind, pitch, Nout, ptr(C), ldc, stream())
    assert rel_err(C[:, :Nout], want) < 3e-6
    assert bool((C[:, Nout:] == 7.0).all())


@pytest.mark.parametrize("K,M,Nn,lda,ldb,ldc", [(16271, 256, 1280, 256, 1280, 3840), (16271, 1280, 256, 1280, 256, 256),
                                                (300, 128, 128, 132, 136, 128), (5, 128, 256, 128, 256, 260),
                                                (16417, 512, 2560, 512, 2560, 7680)])
def test_rows_tn_weight_grad_matches_fp64(K, M, Nn, lda, ldb, ldc):
    """`pm_rows_tn_weight_grad` (chord encoder / decoder weight gradients: 128x128 tiles, fp32 rows split into bf16 planes
    by the loader waves, K slices by atomics): C += A^T B and colsum_a += column sums of A, against fp64; both accumulate;
    columns of C beyond Nn untouched."""
    torch.manual_seed(K + M)
    A = torch.randn(K, lda, device=DEV)
    B = torch.randn(K, ldb, device=DEV)
    C0, cs0 = torch.randn(M, ldc, device=DEV), torch.randn(M, device=DEV)
    C, cs = C0.clone(), cs0.clone()
    call("pm_rows_tn_weight_grad", ptr(A), lda, M, ptr(B), ldb, Nn, K, ptr(C), ldc, ptr(cs), stream())
    want = C0[:, :Nn].double() + A[:, :M].double().t() @ B[:, :Nn].double()
    assert rel_err(C[:, :Nn], want) < 2e-6
    assert torch.equal(C[:, Nn:], C0[:, Nn:])
    assert rel_err(cs, cs0.double() + A[:, :M].double().sum(0)) < 5e-6
    C2 = C0.clone()
    call("pm_rows_tn_weight_grad", ptr(A), lda, M, ptr(B), ldb, Nn, K, ptr(C2), ldc, None, stream())
    assert rel_err(C2[:, :Nn], want) < 2e-6


@pytest.mark.parametrize("d,B,S,drums", [(256, 40, 5, "mixed"), (128, 3, 15, "mixed"), (512, 24, 4, "mixed"), (256, 1, 7, "mixed"),
                                          (256, 6, 5, "none"), (256, 6, 5, "all"), (128, 300, 2, "mixed")])
def test_unembed_input_gradient_in_one_launch(d, B, S, drums):
    """`pm_unembed_dh`: dH = d_logits @ W of the three un-embeddings (pitch per drum / non-drum row list, duration on all
    rows) in one launch on the bf16 pipe, against the fp64 products; every (node, slot) row written exactly once.  Also an
    empty drum / non-drum row list (a job without tiles) and more tiles than workgroups (B = 300)."""
    import ctypes
    cpu = synthetic_batch(B, 2, p=0.3 if B > 1 else 0.12, seed=29)
    if drums != "mixed":
        cpu.is_drum = torch.full_like(cpu.is_drum, drums == "all")
    b = cpu.to(DEV)
    plan = ops.plan_build(b.edge_index, b.edge_type, b.edge_dist, b.bars, b.batch, b.is_drum, b.tokens, b.n_bars,
                          b.s_tensor.shape[0], n_slots=S)
    N, dh = cpu.num_nodes, d // 2
    torch.manual_seed(d + B)
    dl = torch.randn(N, S, 230, device=DEV)
    Wd, Wn, Wu = (torch.randn(131, dh, device=DEV) / 11, torch.randn(131, dh, device=DEV) / 11, torch.randn(99, dh, device=DEV) / 10)
    scratch = torch.empty(int(lib().pm_unembed_dh_scratch_bytes(d)), dtype=torch.uint8, device=DEV)
    dH = torch.full((N, S, d), float("nan"), device=DEV)
    for prep in (1, 0):
        call("pm_unembed_dh", ptr(dl), ptr(Wd), ptr(Wn), ptr(Wu), ptr(plan.buf), N, plan.E, plan.G, d, S, ptr(dH), ptr(scratch),
             prep, stream())
    drum = b.is_drum.bool()
    want = torch.empty(N, S, d, dtype=torch.float64, device=DEV)
    want[drum, :, :dh] = dl[drum][:, :, :131].double() @ Wd.double()
    want[~drum, :, :dh] = dl[~drum][:, :, :131].double() @ Wn.double()
    want[:, :, dh:] = dl[:, :, 131:].double() @ Wu.double()
    assert bool(torch.isfinite(dH).all())
    assert rel_err(dH, want) < 2e-6


@pytest.mark.parametrize("d,B,S,drums", [(256, 40, 6, "mixed"), (128, 3, 15, "mixed"), (512, 12, 4, "mixed"), (256, 6, 5, "none"),
                                          (256, 6, 5, "all"), (256, 300, 3, "mixed"), (256, 1, 7, "mixed")])
def test_decoder_head_over_the_rows_without_pad_targets(d, B, S, drums):
    """Round 6: `pm_unembed_row_lists` compacts the head's three row lists (pitch of the drum rows, pitch of the others,
    duration of all) to the rows that have a target — CrossEntropyLoss(ignore_index) gives a PAD target neither loss nor gradient
    (training.py:101-102,316-323); a row stays when its pitch OR its duration target is not PAD —, lists the rest separately and
    clears their halves of dH; `pm_unembed_ce_rows` / `pm_unembed_dh_rows` then run over the lists.  The lists against numpy
    (ascending, exact); losses, bias gradients, logits and d_logits of the listed rows and dH against `pm_unembed_ce` /
    `pm_unembed_dh` over every row; a second pass over the PAD lists completes the logits."""
    import numpy as np
    cpu = synthetic_batch(B, 2, p=0.3 if B > 1 else 0.12, seed=31)
    if drums != "mixed":
        cpu.is_drum = torch.full_like(cpu.is_drum, drums == "all")
    cpu.tokens[::7, 2, 0] = 130                                   # PAD in one vocabulary only: the row stays
    cpu.tokens[3::11, 1, 1] = 98
    b = cpu.to(DEV)
    plan = ops.plan_build(b.edge_index, b.edge_type, b.edge_dist, b.bars, b.batch, b.is_drum, b.tokens, b.n_bars,
                          b.s_tensor.shape[0], n_slots=S)
    N, dh, R = cpu.num_nodes, d // 2, cpu.num_nodes * S
    lists = torch.full((6, R), -7, dtype=torch.int32, device=DEV)
    counts = torch.full((int(lib().pm_unembed_row_counts_len(N, S)),), -1, dtype=torch.int32, device=DEV)
    dH = torch.full((N, S, d), float("nan"), device=DEV)
    call("pm_unembed_row_lists", ptr(plan.tokens), ptr(plan.buf), N, plan.E, plan.G, d, S, ptr(lists), ptr(lists[3:]), ptr(counts),
         ptr(dH), stream())
    tok = cpu.tokens.numpy()[:, 1:S + 1]                          # target of row (n, s) = tokens[n, s + 1]
    drum = cpu.is_drum.numpy().astype(bool)
    live = (tok[..., 0] != 130) | (tok[..., 1] != 98)
    assert 0 < live.sum() < live.size
    rows = np.arange(R).reshape(N, S)
    jobs = [np.broadcast_to(drum[:, None], (N, S)), np.broadcast_to(~drum[:, None], (N, S)), np.ones((N, S), bool)]
    got_n = counts.tolist()
    for j in range(3):
        for base, sel in ((0, live), (3, ~live)):
            want = rows[jobs[j] & sel]
            n = got_n[base + j + (1 if base else 0)]              # counts: [0..2] listed, [4..6] left out
            assert n == want.size, (j, base, got_n[:8], want.size)
            assert np.array_equal(lists[base + j, :n].cpu().numpy(), want), (j, base)
            assert bool((lists[base + j, n:] == -7).all())
    z = dH.cpu().numpy()
    assert np.all(z[~live] == 0) and np.isnan(z[live]).all()

    torch.manual_seed(d + B)
    H = torch.randn(N, S, d, device=DEV)
    W = [torch.randn(v, dh, device=DEV) * 0.2 for v in (131, 131, 99)]
    bias = [torch.randn(v, device=DEV) for v in (131, 131, 99)]
    wpl = torch.empty(int(lib().pm_unembed_scratch_bytes(d)), dtype=torch.uint8, device=DEV)
    head = (ptr(H), ptr(W[0]), ptr(bias[0]), ptr(W[1]), ptr(bias[1]), ptr(W[2]), ptr(bias[2]), ptr(plan.tokens), ptr(plan.buf),
            N, plan.E, plan.G, d, S, 1.0, None)
    res = []
    for skip in (False, True):
        db = [torch.zeros(v, device=DEV) for v in (131, 131, 99)]
        out = torch.zeros(4, dtype=torch.float64, device=DEV)
        dl = torch.full((N, S, 230), float("nan"), device=DEV)
        lg = torch.full((N, S, 230), float("nan"), device=DEV)
        tail = (ptr(lg), ptr(dl), ptr(db[0]), ptr(db[1]), ptr(db[2]), ptr(out), ptr(wpl))
        if skip:
            call("pm_unembed_ce_rows", *head, *tail, ptr(lists), ptr(counts), stream())
        else:
            call("pm_unembed_ce", *head, *tail, stream())
        res.append((out, dl, db, lg))
    (o0, dl0, db0, lg0), (o1, dl1, db1, lg1) = res
    assert torch.allclose(o0, o1, rtol=1e-9, atol=0) and float(o0[0]) > 0 and float(o0[1]) > 0
    for j in range(3):
        assert rel_err(db1[j], db0[j]) < 1e-5, j
    m = torch.from_numpy(live).to(DEV)
    assert torch.equal(dl1[m], dl0[m]) and torch.equal(lg1[m], lg0[m])
    assert bool((dl0[~m] == 0).all())                             # what the lists leave out
    assert bool(torch.isnan(dl1[~m]).all()) and bool(torch.isnan(lg1[~m]).all())

    scratch = torch.empty(int(lib().pm_unembed_dh_scratch_bytes(d)), dtype=torch.uint8, device=DEV)
    dH0 = torch.full((N, S, d), float("nan"), device=DEV)
    for prep in (1, 0):
        call("pm_unembed_dh", ptr(dl0), ptr(W[0]), ptr(W[1]), ptr(W[2]), ptr(plan.buf), N, plan.E, plan.G, d, S, ptr(dH0), ptr(scratch),
             prep, stream())
    call("pm_unembed_dh_rows", ptr(dl1), ptr(W[0]), ptr(W[1]), ptr(W[2]), ptr(plan.buf), N, plan.E, plan.G, d, S, ptr(dH), ptr(scratch),
         ptr(lists), ptr(counts), stream())
    assert torch.equal(dH, dH0)

    # the second pass of a caller that wants every logit: the rows left out (no loss, no gradient)
    out2 = torch.zeros(4, dtype=torch.float64, device=DEV)
    call("pm_unembed_ce_rows", *head, ptr(lg1), ptr(dl1), None, None, None, ptr(out2), ptr(wpl), ptr(lists[3:]), ptr(counts[4:]), stream())
    assert torch.equal(lg1, lg0) and torch.equal(dl1, dl0) and float(out2[:2].abs().sum()) == 0


def test_head_row_lists_beyond_a_thousand_chunks():
    """`pm_unembed_row_lists` at 1.1 M candidate rows (72 k nodes x 15 slots: 1070 chunks of 1024 — a chunk sums the counts of MORE chunks
    than it has threads before it writes): lists and counts exact against numpy."""
    import numpy as np
    cpu = synthetic_batch(1150, 2, p=0.25, seed=41)
    b = cpu.to(DEV)
    S = 15
    plan = ops.plan_build(b.edge_index, b.edge_type, b.edge_dist, b.bars, b.batch, b.is_drum, b.tokens, b.n_bars,
                          b.s_tensor.shape[0], n_slots=S)
    N, R = cpu.num_nodes, cpu.num_nodes * S
    assert R > 1024 * 1024
    lists = torch.full((6, R), -7, dtype=torch.int32, device=DEV)
    counts = torch.full((int(lib().pm_unembed_row_counts_len(N, S)),), -1, dtype=torch.int32, device=DEV)
    call("pm_unembed_row_lists", ptr(plan.tokens), ptr(plan.buf), N, plan.E, plan.G, 256, S, ptr(lists), ptr(lists[3:]), ptr(counts),
         None, stream())
    tok = cpu.tokens.numpy()[:, 1:S + 1]
    drum = cpu.is_drum.numpy().astype(bool)
    live = (tok[..., 0] != 130) | (tok[..., 1] != 98)
    rows = np.arange(R).reshape(N, S)
    jobs = [np.broadcast_to(drum[:, None], (N, S)), np.broadcast_to(~drum[:, None], (N, S)), np.ones((N, S), bool)]
    got = counts.tolist()
    for j in range(3):
        for base, sel in ((0, live), (3, ~live)):
            want = rows[jobs[j] & sel]
            n = got[base + j + (1 if base else 0)]
            assert n == want.size, (j, base)
            assert np.array_equal(lists[base + j, :n].cpu().numpy(), want), (j, base)


@pytest.mark.parametrize("d,B,S,drums,lists", [(256, 40, 6, "mixed", True), (256, 40, 6, "mixed", False), (512, 12, 4, "mixed", True),
                                                (256, 6, 5, "none", True), (256, 6, 5, "all", False), (256, 300, 3, "mixed", True),
                                                (256, 1, 7, "mixed", True)])
def test_unembed_weight_gradients_in_one_launch(d, B, S, drums, lists):
    """`pm_unembed_dw` (round 6): dW_j += d_logits[rows_j, block_j]^T H[rows_j, half_j] of the three un-embeddings in one launch —
    every row read once, the duration block from column 130 on with the weight row shifted by one — against the fp64 products over
    the plan's drum / non-drum rows and all rows, or over the lists of `pm_unembed_row_lists` (rows left out hold NaN: they must
    not be read); the gradients ACCUMULATE; an empty job and more k-tiles than workgroups (B = 300)."""
    import numpy as np
    cpu = synthetic_batch(B, 2, p=0.3 if B > 1 else 0.12, seed=37)
    if drums != "mixed":
        cpu.is_drum = torch.full_like(cpu.is_drum, drums == "all")
    b = cpu.to(DEV)
    plan = ops.plan_build(b.edge_index, b.edge_type, b.edge_dist, b.bars, b.batch, b.is_drum, b.tokens, b.n_bars,
                          b.s_tensor.shape[0], n_slots=S)
    N, dh, R = cpu.num_nodes, d // 2, cpu.num_nodes * S
    torch.manual_seed(d + B + S)
    dl = torch.randn(N, S, 230, device=DEV)
    H = torch.randn(N, S, d, device=DEV)
    tok = cpu.tokens.numpy()[:, 1:S + 1]
    live = torch.from_numpy((tok[..., 0] != 130) | (tok[..., 1] != 98)).to(DEV) if lists else torch.ones(N, S, dtype=torch.bool, device=DEV)
    lp = cp = None
    if lists:
        lst = torch.empty(3, R, dtype=torch.int32, device=DEV)
        cnt = torch.empty(int(lib().pm_unembed_row_counts_len(N, S)), dtype=torch.int32, device=DEV)
        call("pm_unembed_row_lists", ptr(plan.tokens), ptr(plan.buf), N, plan.E, plan.G, d, S, ptr(lst), None, ptr(cnt), None, stream())
        dl[~live] = float("nan")
        H[~live] = float("nan")
        lp, cp = ptr(lst), ptr(cnt)
    g0 = [torch.randn(v, dh, device=DEV) for v in (131, 131, 99)]
    g = [t.clone() for t in g0]
    call("pm_unembed_dw", ptr(dl), ptr(H), ptr(plan.buf), N, plan.E, plan.G, d, S, ptr(g[0]), ptr(g[1]), ptr(g[2]), lp, cp, stream())
    drum = b.is_drum.bool()[:, None].expand(N, S)
    dl64, H64 = torch.nan_to_num(dl).double(), torch.nan_to_num(H).double()
    for j, (rows, cols, half) in enumerate(((drum & live, slice(0, 131), slice(0, dh)), (~drum & live, slice(0, 131), slice(0, dh)),
                                            (live, slice(131, 230), slice(dh, d)))):
        want = g0[j].double() + dl64[rows][:, cols].t() @ H64[rows][:, half]
        assert bool(torch.isfinite(g[j]).all()), j
        assert rel_err(g[j], want) < 3e-6, (j, rel_err(g[j], want))


def test_unembed_input_gradient_refuses_offsets_beyond_32_bits():
    """`pm_unembed_dh` addresses d_logits and dH with 32-bit byte offsets: sizes beyond that are refused before anything is
    launched (the step then takes the GEMM path, `vae_step.hip`)."""
    d, S, N = 256, 5, 500_000                                      # 2.5 M rows x 1 KB of dH each
    x = torch.zeros(16, device=DEV)
    rc = lib().pm_unembed_dh(ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), N, 1, 1, d, S, ptr(x), ptr(x), 0, stream())
    assert rc == -3                                                # PM_E_UNSUPPORTED


def test_launch_profiler_class_mask_and_stride():
    """`pm_prof_configure` (bench.py's roofline timing): only the selected classes are bracketed, every stride-th
    launch of each; durations and algorithmic work come back per class."""
    import ctypes
    L = lib()
    A = torch.randn(256, 128, device=DEV)
    B = torch.randn(128, 64, device=DEV)
    out = torch.empty(256, 64, device=DEV)

    def run(mask, stride, launches):
        L.pm_prof_configure(mask, stride)
        L.pm_prof_begin(64)
        for _ in range(launches):
            ops.gemm(A, B, out, 256, 64, 128, 128, 64, 64)            # NN, 64x64x16 tiles: class 0
        torch.cuda.synchronize()
        ms, work, cnt = (ctypes.c_double * PROF_NCLASS)(), (ctypes.c_double * PROF_NCLASS)(), (ctypes.c_int64 * PROF_NCLASS)()
        assert L.pm_prof_end(ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(work, ctypes.c_void_p),
                             ctypes.cast(cnt, ctypes.c_void_p)) == 0
        return list(ms), list(work), list(cnt)
    try:
        ms, work, cnt = run(-1, 1, 6)
        assert cnt[0] == 6 and sum(cnt) == 6 and ms[0] > 0 and work[0] == 6 * 2.0 * 256 * 64 * 128
        ms, work, cnt = run(-1, 4, 9)                                 # launches 0, 4, 8
        assert cnt[0] == 3
        ms, work, cnt = run(1 << 33, 1, 5)                            # segment-reduce forward only: no GEMM is timed
        assert sum(cnt) == 0
        assert L.pm_prof_configure(-1, 0) != 0
    finally:
        L.pm_prof_configure(-1, 1)


def test_eval_batchnorm_folded_into_weights():
    """pm_bn_fold_weights + the RELU_ADD epilogue: x + relu(BN_eval(A @ W + b)) in one product."""
    from polyphemus_amd import ops
    torch.manual_seed(5)
    N, K, d = 777, 7 * 64, 64
    A, W, b, x = (torch.randn(N, K, device=DEV), torch.randn(K, d, device=DEV) * 0.1, torch.randn(d, device=DEV),
                  torch.randn(N, d, device=DEV))
    gamma, beta, rm, rv = (torch.rand(d, device=DEV) + 0.5, torch.randn(d, device=DEV), torch.randn(d, device=DEV),
                           torch.rand(d, device=DEV) + 0.1)
    Wf, bf = ops.bn_fold_weights(W, b, gamma, beta, rm, rv, 1e-5)
    out = x.clone()
    ops.gemm(A, Wf, out, N, d, K, K, d, d, bias=bf, relu_add=True)
    h = A.double() @ W.double() + b.double()
    ref = x.double() + torch.relu((h - rm.double()) / torch.sqrt(rv.double() + 1e-5) * gamma.double() + beta.double())
    assert rel_err(out, ref) < 1e-5


@pytest.mark.parametrize("d,B,p", [(256, 24, 0.25), (32, 5, 0.2), (16, 4, 0.1), (512, 6, 0.2)])
def test_fused_unembed_ce_matches_products_plus_loss(d, B, p):
    """pm_unembed_ce against the unfused path — the three Linear(d/2 -> 131 | 131 | 99) products of
    model.py:561-567 in float64 + CrossEntropyLoss(ignore_index = PAD) of training.py:316-323: logits, d_logits,
    both losses and the three bias gradients."""
    from polyphemus_amd import ops
    from polyphemus_amd.synthetic import synthetic_batch
    torch.manual_seed(d)
    g = synthetic_batch(B, 2, p=p, seed=d + B).to(DEV)
    plan = ops.plan_build(g.edge_index, g.edge_type, g.edge_dist, g.bars, g.batch, g.is_drum, g.tokens, 2, g.s_tensor.shape[0])
    N, dh = g.num_nodes, d // 2
    H = torch.randn(N, 15, d, device=DEV)
    W = [torch.randn(v, dh, device=DEV) * 0.2 for v in (131, 131, 99)]
    bias = [torch.randn(v, device=DEV) for v in (131, 131, 99)]
    db = [torch.zeros(v, device=DEV) for v in (131, 131, 99)]
    out, dl, lg = ops.unembed_ce(H, W[0], bias[0], W[1], bias[1], W[2], bias[2], plan, want_logits=True, dbias=db)
    Hd = H.double()
    drum = g.is_drum.bool()
    ref = torch.empty(N, 15, 230, dtype=torch.float64, device=DEV)
    ref[drum, :, :131] = Hd[drum][..., :dh] @ W[0].double().T + bias[0].double()
    ref[~drum, :, :131] = Hd[~drum][..., :dh] @ W[1].double().T + bias[1].double()
    ref[..., 131:] = Hd[..., dh:] @ W[2].double().T + bias[2].double()
    assert rel_err(lg, ref) < 2e-6
    ref.requires_grad_(True)
    tp, td = g.tokens[:, 1:, 0].long().reshape(-1), g.tokens[:, 1:, 1].long().reshape(-1)
    lp = F.cross_entropy(ref[..., :131].reshape(-1, 131), tp, ignore_index=130)
    ld = F.cross_entropy(ref[..., 131:].reshape(-1, 99), td, ignore_index=98)
    (lp + ld).backward()
    o = out.tolist()
    assert abs(o[0] - float(lp)) < 1e-6 * max(1.0, float(lp)) and abs(o[1] - float(ld)) < 1e-6 * max(1.0, float(ld))
    assert rel_err(dl, ref.grad) < 1e-5
    gb = ref.grad
    assert rel_err(db[0], gb[drum][..., :131].sum((0, 1))) < 1e-4
    assert rel_err(db[1], gb[~drum][..., :131].sum((0, 1))) < 1e-4
    assert rel_err(db[2], gb[..., 131:].sum((0, 1))) < 1e-4


@pytest.mark.parametrize("cells", [1, 2, 3, 4, 5, 7, 9])
def test_plan_build_on_tiny_graphs_stays_inside_the_plan_buffer(cells):
    """N = 1 .. 9 nodes (one sample of one bar): pm_plan_build carves tile sums, node classes and class histograms out of the
    plan's scratch field; round 3's layout reserved less than that for N <= 4 and the kernels wrote past the caller's
    buffer.  The plan is built into a buffer with guard words behind it, and must match the numpy plan."""
    import numpy as np
    from polyphemus_amd import constants as C
    from polyphemus_amd._lib import call, plan_layout, ptr, stream
    from polyphemus_amd.graphs import collate_samples, graph_from_structure
    rng = np.random.default_rng(cells)
    s = np.zeros((1, 4, 32), bool)
    s.reshape(-1)[rng.choice(128, size=cells, replace=False)] = True
    g = graph_from_structure(s)
    g["tokens"] = np.full((int(s.sum()), 16, 2), 0, np.int32)
    g["tokens"][:, 0] = (C.PITCH_SOS, C.DUR_SOS)
    g["tokens"][:, 1] = (60, 8)
    g["tokens"][:, 2] = (C.PITCH_EOS, C.DUR_EOS)
    g["tokens"][:, 3:] = (C.PITCH_PAD, C.DUR_PAD)
    g["s_tensor"] = s.astype(np.float32)
    cpu = collate_samples([g], 1)
    b = cpu.to(DEV)
    N, E, G = cpu.num_nodes, cpu.edge_index.shape[1], 1
    assert N == cells
    off = plan_layout(N, E, G)
    GUARD = 256
    buf = torch.full((off[-1] + GUARD,), 0x5A5A5A5A, dtype=torch.int32, device=DEV)
    drum = b.is_drum.view(torch.uint8) if b.is_drum.dtype == torch.bool else b.is_drum
    call("pm_plan_build", ptr(b.edge_index), ptr(b.edge_type.to(torch.int32)), ptr(b.edge_dist.to(torch.int32)), ptr(b.bars),
         ptr(b.batch), ptr(drum), ptr(b.tokens.to(torch.int32).contiguous()), 1, 15, N, E, G, ptr(buf), stream())
    torch.cuda.synchronize()
    assert bool((buf[off[-1]:] == 0x5A5A5A5A).all()), "pm_plan_build wrote past the end of the plan buffer"
    src, dst = cpu.edge_index[0].numpy(), cpu.edge_index[1].numpy()
    et = cpu.edge_type.numpy()
    rowptr = np.zeros(N * 6 + 1, np.int64)
    np.add.at(rowptr, dst * 6 + et + 1, 1)
    np.testing.assert_array_equal(buf[off[0]:off[0] + N * 6 + 1].cpu().numpy(), np.cumsum(rowptr))
    colptr = np.zeros(N + 1, np.int64)
    np.add.at(colptr, src + 1, 1)
    j = 4                                                     # PM_PLAN_COLPTR
    np.testing.assert_array_equal(buf[off[j]:off[j] + N + 1].cpu().numpy(), np.cumsum(colptr))


def test_relu_bwd_planes_matches_torch_and_the_plane_split():
    """pm_relu_bwd_planes (the GCL layer tail's backward of a model built with batch_norm = False, model.py:202-206):
    dh = dy * [h > 0], bit for bit against torch, as fp32 and as the exact three-plane bf16 split of that fp32 value."""
    from polyphemus_amd._lib import call, ptr, stream
    torch.manual_seed(4)
    n = 4096 * 96
    dy = torch.randn(n, device=DEV) * torch.logspace(-12, 12, n, device=DEV)
    h = torch.randn(n, device=DEV)
    h[::7] = 0.0                                                  # relu'(0) = 0
    want = torch.where(h > 0, dy, torch.zeros_like(dy))
    dh = torch.full_like(dy, float("nan"))
    planes = torch.zeros(3, n, dtype=torch.int16, device=DEV)
    call("pm_relu_bwd_planes", ptr(dy), ptr(h), n, ptr(dh), ptr(planes), n, stream())
    assert torch.equal(dh, want)
    f = lambda t: (t.to(torch.int32) << 16).view(torch.float32).double()
    assert torch.equal((f(planes[0]) + f(planes[1]) + f(planes[2])).float(), want)
    assert torch.equal(planes, ops.split_planes(want))
    only = torch.zeros(3, n, dtype=torch.int16, device=DEV)      # planes only / fp32 only
    call("pm_relu_bwd_planes", ptr(dy), ptr(h), n, None, ptr(only), n, stream())
    assert torch.equal(only, planes)
    with pytest.raises(Exception):
        call("pm_relu_bwd_planes", ptr(dy), ptr(h), n - 1, ptr(dh), None, 0, stream())      # n % 4 != 0


# ---- the fp16 pair operand format of the GCL products (PmH2, include/polyphemus_hip.h) --------------------------------------
import ctypes as _ct
from polyphemus_amd import _lib as _lib_mod


class _PmH2(_ct.Structure):
    _fields_ = [("absmax_in", _ct.c_void_p), ("absmax_aux", _ct.c_void_p), ("scale_out", _ct.c_void_p), ("w_scale", _ct.c_float),
                ("reserved", _ct.c_int32)]


def _absmax_words(t):
    w = torch.zeros(64, dtype=torch.int32, device=t.device)
    call("pm_absmax", ptr(t), t.numel(), ptr(w), stream())
    return w


def _pair_value(planes, scale):
    """[2 or 3, n] int16 planes in the fp16 pair format -> fp64 values"""
    return (planes[0].view(torch.float16).double() + planes[1].view(torch.float16).double()) / float(scale)


@pytest.mark.parametrize("d,p,xs,gs", [(128, 0.0, 1.0, 1.0), (256, 0.1, 37.0, 1e-6), (256, 0.0, 1e-3, 1e3), (512, 0.1, 5.0, 1e-5),
                                       (256, 0.1, "split", 1e-4)])
def test_gcl_products_in_the_fp16_pair_format(d, p, xs, gs):
    """The three GCL products on fp16 pair operands (v * 2^k = hi + lo, three MFMA products per fp32 product; PmH2) against the
    same kernels on the exact bf16 triple and against fp64, over operand magnitudes from 1e-6 to 1e3 (`xs`: scale of the layer
    input, `gs`: of the incoming gradient — the power-of-two scales come from the tensors' |max| words):
      * `pm_absmax`: the 64 words hold max |x| exactly;
      * forward: h to 5e-6 of both; the A' planes decode (hi + lo) / scale to the triple's exact aggregate within 2^-21 of its |max|;
      * input gradient (d <= 256: norm backward inside, `pm_gcl_input_grad_bn_h2`; d = 512: `pm_bn_bwd_fused_h2` +
        `pm_gcl_input_grad_fused_h2`): dA' to 5e-6, the dh planes decode to the triple's dh within 2^-21 of |max|;
      * weight gradient from the two pair-format planes: 3e-6 of an fp64 contraction of the decoded operands."""
    if xs == "split":                  # 258 row tiles for 256 CUs: half tiles (csrc/tile_order.h)
        cpu, xs = synthetic_batch(256, 2, p=0.25, seed=1235), 3.0
    else:
        cpu = synthetic_batch(40, 2, p=0.3, seed=29)
    b, plan = make_plan(cpu)
    N, dd = cpu.num_nodes, d * d
    clamps0 = _lib_mod.h2_clamp_events()
    torch.manual_seed(11)
    x = torch.randn(N, d, device=DEV) * xs
    T = ops.edge_table(torch.randn(d, 32, device=DEV) * 0.5, torch.randn(d, device=DEV) * 0.1)
    W = torch.randn(7 * d, d, device=DEV) / d ** 0.5
    bias = torch.randn(d, device=DEV) * xs
    WS = 16.0
    Wf3, Wft3 = ops.split_planes_frag(W, 1), ops.split_planes_frag(W, 0)
    Wf2, Wft2 = torch.zeros_like(Wf3), torch.zeros_like(Wft3)
    for kind, dst in ((1, Wf2), (0, Wft2)):
        call("pm_split_planes_frag_h2", ptr(W), 7 * d, d, kind, 1, 7 * dd, 7 * dd * 3, WS, ptr(dst), stream())
    mx, mt = _absmax_words(x), _absmax_words(T)
    assert float(mx.view(torch.float32).max()) == float(x.abs().max()) and float(mt.view(torch.float32).max()) == float(T.abs().max())
    # ---- forward
    P3 = torch.zeros(3, N * 4 * d, dtype=torch.int16, device=DEV)
    h3 = ops.gcl_forward_fused(x, T, plan, p, 5, 2, Wf3, bias, planes=P3)
    P2 = torch.zeros(3, N * 4 * d, dtype=torch.int16, device=DEV)
    sA = torch.zeros(1, device=DEV)
    s2 = torch.zeros(8, 2, d, dtype=torch.float64, device=DEV)
    h2 = torch.empty(N, d, device=DEV)
    hh = _PmH2(ptr(mx), ptr(mt), ptr(sA), WS, 0)
    call("pm_gcl_forward_fused_h2", ptr(x), ptr(T), ptr(plan.buf), N, plan.E, plan.G, d, p, 5, 2, ptr(Wf2), ptr(bias), 1, ptr(h2),
         ptr(s2), ptr(P2), N * 4 * d, _ct.addressof(hh), stream())
    A3 = _planes_value(P3).double()
    A2 = _pair_value(P2, float(sA))
    sa = float(sA)
    assert sa > 0 and np.log2(sa) == round(np.log2(sa)) and 2.0 ** 9 <= float(A3.abs().max()) * sa < 2.0 ** 13
    assert float((A2 - A3).abs().max()) <= 2.0 ** -21 * float(A3.abs().max())
    assert rel_err(h2, h3) < 5e-6
    trel_t = plan.field("node_trel").long()[:N]
    Wn = torch.stack([torch.cat([W[t * d:(t + 1) * d], W[4 * d:]]) for t in range(4)]).double()
    assert rel_err(h2, torch.einsum("nk,nkj->nj", A3.view(N, 4 * d), Wn[trel_t]) + bias.double()) < 5e-6
    assert rel_err(s2.sum(0)[0], h2.double().sum(0)) < 1e-12
    assert P2[2].abs().max() == 0                                      # plane 2 is not part of the format
    # ---- input gradient with the norm backward
    hpre = torch.randn(N, d, device=DEV) * 1.5 + 0.3
    du = torch.randn(N, d, device=DEV) * gs
    gamma, beta = torch.rand(d, device=DEV) + 0.5, torch.randn(d, device=DEV) * 0.2
    mean, var = hpre.mean(0), hpre.var(0, unbiased=False)
    acc3 = ops.bn_bwd_sums(hpre, du, mean, var, gamma, beta)
    mdu = _absmax_words(du)
    sdh = torch.zeros(1, device=DEV)
    dA2 = torch.full((N, 4 * d), float("nan"), device=DEV)
    D2 = torch.zeros(3, N * d, dtype=torch.int16, device=DEV)
    hb = _PmH2(ptr(mdu), None, ptr(sdh), WS, 0)
    if d <= 256:
        dA3, D3 = ops.gcl_input_grad_bn(hpre, du, mean, var, gamma, beta, acc3, plan, Wft3)
        nb = ops._BnBwd(ptr(hpre), ptr(du), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(acc3), None, None, None, 1e-5, 1, 0, 0)
        call("pm_gcl_input_grad_bn_h2", _ct.addressof(nb), ptr(D2), N * d, ptr(plan.buf), N, plan.E, plan.G, d, ptr(Wft2), 1, ptr(dA2),
             _ct.addressof(hb), stream())
    else:
        D3 = torch.zeros(3, N * d, dtype=torch.int16, device=DEV)
        call("pm_bn_bwd_fused", ptr(hpre), ptr(du), N, d, ptr(mean), ptr(var), 1e-5, ptr(gamma), ptr(beta), 1, None, None, None, None,
             ptr(acc3), ptr(D3), N * d, 1, stream())
        dA3 = ops.gcl_input_grad_fused(D3, plan, d, Wft3)
        call("pm_bn_bwd_fused_h2", ptr(hpre), ptr(du), N, d, ptr(mean), ptr(var), 1e-5, ptr(gamma), ptr(beta), 1, None, None, None,
             ptr(acc3), ptr(D2), N * d, 1, _ct.addressof(hb), stream())
        call("pm_gcl_input_grad_fused_h2", ptr(D2), N * d, ptr(plan.buf), N, plan.E, plan.G, d, ptr(Wft2), 1, ptr(dA2), ptr(sdh), WS,
             stream())
    dh3 = _planes_value(D3).double()
    dh2 = _pair_value(D2, float(sdh))
    assert 2.0 ** 4 <= float(dh3.abs().max()) * float(sdh) < 2.0 ** 15     # inside fp16's window with room on both sides
    assert float((dh2 - dh3).abs().max()) <= 2.0 ** -21 * float(dh3.abs().max())
    dst, et = cpu.edge_index[1], cpu.edge_type
    on, nx = torch.zeros(N, dtype=torch.bool), torch.zeros(N, dtype=torch.bool)
    on[dst[et == 4]] = True
    nx[dst[et == 5]] = True
    keep = torch.ones(N, 4, d, dtype=torch.bool)
    keep[~on, 1] = False
    keep[~nx, 2] = False
    keep = keep.view(N, 4 * d).to(DEV)
    assert rel_err(dA2[keep], dA3[keep]) < 5e-6
    # ---- weight gradient from the two pair-format plane sets
    base = torch.randn(7 * d, d, device=DEV) * xs * gs
    dW2 = base.clone()
    call("pm_gcl_weight_grad_fused_h2", ptr(P2), N * 4 * d, ptr(D2), N * d, ptr(plan.buf), N, plan.E, plan.G, d, 1, ptr(dW2), ptr(sA),
         ptr(sdh), stream())
    A2v, trel = A2.view(N, 4 * d), trel_t
    A2v = torch.where(keep, A2v, torch.zeros_like(A2v))               # (blocks the forward does not write for a row are zero aggregates)
    want = base.double().clone()
    dh2v = dh2.view(N, d)
    for t in range(4):
        rows = trel == t
        want[t * d:(t + 1) * d] += A2v[rows, :d].T @ dh2v[rows]
    want[4 * d:] += A2v[:, d:].T @ dh2v
    scale = float((want - base.double()).abs().max())
    assert float((dW2.double() - want).abs().max()) < 3e-6 * scale
    assert _lib_mod.h2_clamp_events() == clamps0              # nothing saturated: every operand fitted its power-of-two scale


def test_pair_format_saturation_is_counted():
    """the fp16 pair split clamps at +-65504 where a tensor's |max| BOUND was too small (dh: the bound
    16 gamma rstd |du|max holds for |xhat| <= 14, a BatchNorm column with one outlier reaches sqrt(N)) — silently, before round 6.
    `pm_h2_clamp_events` counts the threads that cut a value.  Here the bound is made wrong on purpose: (1) |du|max words that
    understate the gradient 4096-fold in `pm_gcl_input_grad_bn_h2` and `pm_bn_bwd_fused_h2`; (2) weight planes with a scale that
    takes glorot-range weights out of fp16's range.  The properly scaled calls before and after leave the counter alone."""
    cpu = synthetic_batch(40, 2, p=0.3, seed=29)
    b, plan = make_plan(cpu)
    N = cpu.num_nodes
    torch.manual_seed(5)
    _lib_mod.h2_clamp_events(reset=True)
    for d in (256, 512):
        W = torch.randn(7 * d, d, device=DEV) / d ** 0.5
        Wft2 = torch.zeros(3, 7 * d * d, dtype=torch.int16, device=DEV)
        call("pm_split_planes_frag_h2", ptr(W), 7 * d, d, 0, 1, 7 * d * d, 7 * d * d * 3, 16.0, ptr(Wft2), stream())
        assert _lib_mod.h2_clamp_events() == 0
        hpre = torch.randn(N, d, device=DEV) * 1.5 + 0.3
        du = torch.randn(N, d, device=DEV)
        gamma, beta = torch.rand(d, device=DEV) + 0.5, torch.randn(d, device=DEV) * 0.2
        mean, var = hpre.mean(0), hpre.var(0, unbiased=False)
        acc3 = ops.bn_bwd_sums(hpre, du, mean, var, gamma, beta)
        for lie in (1.0, 1.0 / 4096):
            mdu = _absmax_words(du * lie)
            sdh = torch.zeros(1, device=DEV)
            dA2 = torch.empty(N, 4 * d, device=DEV)
            D2 = torch.zeros(3, N * d, dtype=torch.int16, device=DEV)
            hb = _PmH2(ptr(mdu), None, ptr(sdh), 16.0, 0)
            if d <= 256:
                nb = ops._BnBwd(ptr(hpre), ptr(du), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(acc3), None, None, None, 1e-5, 1, 0, 0)
                call("pm_gcl_input_grad_bn_h2", _ct.addressof(nb), ptr(D2), N * d, ptr(plan.buf), N, plan.E, plan.G, d, ptr(Wft2), 1,
                     ptr(dA2), _ct.addressof(hb), stream())
            else:
                call("pm_bn_bwd_fused_h2", ptr(hpre), ptr(du), N, d, ptr(mean), ptr(var), 1e-5, ptr(gamma), ptr(beta), 1, None, None, None,
                     ptr(acc3), ptr(D2), N * d, 1, _ct.addressof(hb), stream())
            n = _lib_mod.h2_clamp_events(reset=True)
            assert (n == 0) if lie == 1.0 else (n > 0), (d, lie, n)
    call("pm_split_planes_frag_h2", ptr(W), 7 * d, d, 0, 1, 7 * d * d, 7 * d * d * 3, 1.0e7, ptr(Wft2), stream())
    assert _lib_mod.h2_clamp_events(reset=True) > 0
    assert _lib_mod.h2_clamp_events() == 0
