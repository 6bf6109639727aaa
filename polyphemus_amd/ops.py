"""Thin torch-tensor wrappers over the C ABI (one function per entry point).

Every wrapper checks device / dtype / contiguity, allocates outputs with torch (the
library never allocates) and enqueues on torch's current stream.  No arithmetic
happens here: if the HIP extension is missing these raise `HipExtensionError`.
"""
from __future__ import annotations

from typing import Optional

import ctypes

import torch

from . import constants as C
from ._lib import PLAN_FIELDS, HipExtensionError, call, lib, plan_layout, ptr, stream

F32, I32, I64, U8, F64 = torch.float32, torch.int32, torch.int64, torch.uint8, torch.float64

# Optional per-launch timing with HIP events on the launch stream (used by bench.py for the roofline
# figures): PROF = {"kernel class": [(start_event, end_event, algorithmic_work), ...]} or None.
PROF = None


def _prof_begin(name: str):
    if PROF is None:
        return None
    PROF.setdefault(name, [])
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return e0


def _prof_end(name: str, e0, work: float):
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        PROF[name].append((e0, e1, work))


_GEMM_TILES = ("64x64x16", "128x128x16", "64x64x32", "128x128x32", "x6:128x128x16", "x6:128x64x16", "x6:64x64x32",
               "x6:128x128x32", "planes:64x64x32", "planesB:64x128x32", "planes:128x128x32")


def gemm_class(transA: bool, transB: bool, M: int, N: int, K: int) -> str:
    """Kernel instance a GEMM call maps to (asks the library which tile configuration it picks)."""
    from ._lib import lib
    return f"gemm_{'T' if transA else 'N'}{'T' if transB else 'N'}_{_GEMM_TILES[lib().pm_gemm_config(int(transA), M, N, K)]}"


def _chk(t: Optional[torch.Tensor], dtype, name: str, allow_none=False):
    if t is None:
        if allow_none:
            return
        raise ValueError(f"{name} is None")
    if not t.is_cuda:
        raise HipExtensionError(f"{name} lives on {t.device}: the HIP path needs device tensors (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")


class Plan:
    """Device-side graph plan of one batch (see include/polyphemus_hip.h, pm_plan_build)."""

    def __init__(self, buf: torch.Tensor, N: int, E: int, G: int, n_bars: int, tokens: torch.Tensor,
                 is_drum: torch.Tensor):
        self.buf, self.N, self.E, self.G, self.n_bars = buf, N, E, G, n_bars
        self.B = G // n_bars
        self.tokens, self.is_drum = tokens, is_drum
        self._off = plan_layout(N, E, G)

    def field(self, name: str) -> torch.Tensor:
        i = PLAN_FIELDS.index(name)
        v = self.buf[self._off[i]:self._off[i + 1]]
        return v.view(torch.float32) if name == "csc_invcnt" else v

    @property
    def tok_hist(self) -> torch.Tensor:
        return self.field("tok_hist")

    @property
    def drum_list(self) -> torch.Tensor:
        return self.field("group_list")[:self.N]

    @property
    def nondrum_list(self) -> torch.Tensor:
        return self.field("group_list")[self.N:2 * self.N]

    @property
    def drum_rows(self) -> torch.Tensor:
        """(node, slot) rows node*15+slot of the drum nodes (first 15*n_drum entries are live)."""
        return self.field("row_list")[:self.N * C.N_SLOTS]

    @property
    def nondrum_rows(self) -> torch.Tensor:
        return self.field("row_list")[self.N * C.N_SLOTS:2 * self.N * C.N_SLOTS]

    @property
    def group_cnt(self) -> torch.Tensor:
        """[n_drum, n_non_drum, 15*n_drum, 15*n_non_drum] (device)."""
        return self.field("group_cnt")


def edge_attrs_to_ids(edge_attrs: torch.Tensor):
    _chk(edge_attrs, F32, "edge_attrs")
    E = edge_attrs.shape[0]
    et = torch.empty(E, dtype=I32, device=edge_attrs.device)
    ed = torch.empty(E, dtype=I32, device=edge_attrs.device)
    call("pm_edge_attrs_to_ids", ptr(edge_attrs), E, ptr(et), ptr(ed), stream())
    return et, ed


def tokens_from_onehot(c_tensor: torch.Tensor) -> torch.Tensor:
    _chk(c_tensor, F32, "c_tensor")
    N = c_tensor.shape[0]
    tok = torch.empty(N, 16, 2, dtype=I32, device=c_tensor.device)
    call("pm_tokens_from_onehot", ptr(c_tensor), N, ptr(tok), stream())
    return tok


def plan_build(edge_index, edge_type, edge_dist, bars, batch, is_drum, tokens, n_bars: int, G: int,
               n_slots: int = C.N_SLOTS) -> Plan:
    _chk(edge_index, I64, "edge_index"); _chk(edge_type, I32, "edge_type"); _chk(edge_dist, I32, "edge_dist")
    _chk(bars, I64, "bars"); _chk(batch, I64, "batch"); _chk(tokens, I32, "tokens")
    if is_drum.dtype == torch.bool:
        is_drum = is_drum.view(U8)
    _chk(is_drum, U8, "is_drum")
    N, E = bars.shape[0], edge_index.shape[1]
    off = plan_layout(N, E, G)
    buf = torch.empty(off[-1], dtype=I32, device=bars.device)
    call("pm_plan_build", ptr(edge_index), ptr(edge_type), ptr(edge_dist), ptr(bars), ptr(batch), ptr(is_drum),
         ptr(tokens), n_bars, n_slots, N, E, G, ptr(buf), stream())
    return Plan(buf, N, E, G, n_bars, tokens, is_drum)


def edge_table(nn_weight, nn_bias):
    _chk(nn_weight, F32, "nn.weight"); _chk(nn_bias, F32, "nn.bias")
    d = nn_weight.shape[0]
    T = torch.empty(C.N_DISTS, d, dtype=F32, device=nn_weight.device)
    call("pm_edge_table", ptr(nn_weight), ptr(nn_bias), d, ptr(T), stream())
    return T


def edge_table_bwd(dT, d_nn_weight, d_nn_bias):
    call("pm_edge_table_bwd", ptr(dT), dT.shape[1], ptr(d_nn_weight), ptr(d_nn_bias), stream())


def graph_build(s_tensor: torch.Tensor, n_bars: int):
    """Bar graphs of a batch on the device (`pm_graph_count` + `pm_graph_emit`): `s_tensor` float32 [G,4,32] 0/1
    (empty bars get cell [0,0] switched on in place, data.py:152-153).  Returns a dict of device tensors with the
    reference's node numbering and edge order; ONE host read (N, E) sizes the outputs."""
    _chk(s_tensor, F32, "s_tensor")
    G = s_tensor.shape[0]
    dev = s_tensor.device
    bn, be = torch.empty(G, dtype=I32, device=dev), torch.empty(G, dtype=I32, device=dev)
    npt, ept = torch.empty(G + 1, dtype=I32, device=dev), torch.empty(G + 1, dtype=I32, device=dev)
    tot = torch.empty(2, dtype=I32, device=dev)
    call("pm_graph_count", ptr(s_tensor), G, ptr(bn), ptr(be), ptr(npt), ptr(ept), ptr(tot), stream())
    N, E = (int(v) for v in tot.tolist())
    out = dict(edge_index=torch.empty(2, E, dtype=I64, device=dev), edge_type=torch.empty(E, dtype=I32, device=dev),
               edge_dist=torch.empty(E, dtype=I32, device=dev), bars=torch.empty(N, dtype=I64, device=dev),
               batch=torch.empty(N, dtype=I64, device=dev), is_drum=torch.empty(N, dtype=U8, device=dev),
               node_cell=torch.empty(N, dtype=I32, device=dev), node_ptr=npt, edge_ptr=ept, num_nodes=N)
    call("pm_graph_emit", ptr(s_tensor), G, n_bars, ptr(npt), ptr(ept), N, E, ptr(out["edge_index"]), ptr(out["edge_type"]),
         ptr(out["edge_dist"]), ptr(out["bars"]), ptr(out["batch"]), ptr(out["is_drum"]), ptr(out["node_cell"]), stream())
    out["is_drum"] = out["is_drum"].view(torch.bool)
    return out


def binary_from_logits(s_logits: torch.Tensor, thresh: float = 0.5) -> torch.Tensor:
    """`Decoder._binary_from_logits` (model.py:609-623) without the `nonzero` sync: bool tensor of s_logits' shape."""
    _chk(s_logits, F32, "s_logits")
    if s_logits.shape[-2:] != (4, 32):
        raise ValueError("s_logits must be [..., 4, 32]")
    out = torch.empty(s_logits.shape, dtype=U8, device=s_logits.device)
    call("pm_binary_from_logits", ptr(s_logits), s_logits.numel() // 128, float(thresh), None, ptr(out), stream())
    return out.view(torch.bool)


def mtp_from_logits(c_logits: torch.Tensor, s_tensor: torch.Tensor, check: bool = True) -> torch.Tensor:
    """`mtp_from_logits` (utils.py:59-79): [B,nb,4,32,15,230] with the nodes' logits on the active cells of `s_tensor`
    ([B,nb,4,32], any dtype) and hard silences elsewhere.  `check` reads the active-cell count back (one host sync) and
    raises like the reference's masked assignment when it differs from c_logits' node count."""
    _chk(c_logits, F32, "c_logits")
    if c_logits.dim() != 3 or c_logits.shape[1:] != (C.MAX_SIMU_TOKENS - 1, C.D_TOKEN_PAIR):
        raise ValueError("c_logits must be [N,15,230]")
    if s_tensor.dim() != 4 or s_tensor.shape[-2:] != (4, 32):
        raise ValueError("s_tensor must be [B,n_bars,4,32]")
    s = s_tensor.to(F32).contiguous()
    G, N, dev = s.numel() // 128, c_logits.shape[0], c_logits.device
    bn, npt = torch.empty(G, dtype=I32, device=dev), torch.empty(G + 1, dtype=I32, device=dev)
    mtp = torch.empty(*s_tensor.shape, C.MAX_SIMU_TOKENS - 1, C.D_TOKEN_PAIR, dtype=F32, device=dev)
    call("pm_mtp_from_logits", ptr(c_logits), ptr(s), G, N, ptr(bn), ptr(npt), ptr(mtp), stream())
    if check:
        active = int(npt[G])
        if active != N:
            raise ValueError(f"shape mismatch: s_tensor has {active} active cells, c_logits has {N} nodes")
    return mtp


def gcl_forward_fused(x, T, plan: Plan, dropout_p: float, seed: int, layer_uid: int, w_frag, bias, col_stats=None,
                      planes=None, use_classes: bool = True):
    """`pm_gcl_forward_fused`: h = A'(x) @ [W_t; W_4; W_5; root] + bias of one GCL layer in one kernel (compact graphs,
    d in {128, 256, 512}); `w_frag` = `split_planes_frag(W, 1)`; `planes` (int16 [3, N*4d], optional) receives the A' planes."""
    _chk(x, F32, "x"); _chk(T, F32, "T")
    N, d = x.shape
    h = torch.empty(N, d, dtype=F32, device=x.device)
    call("pm_gcl_forward_fused", ptr(x), ptr(T), ptr(plan.buf), N, plan.E, plan.G, d, float(dropout_p),
         seed & 0xFFFFFFFF, layer_uid, ptr(w_frag), ptr(bias), 1 if use_classes else 0, ptr(h), ptr(col_stats),
         ptr(planes), 0 if planes is None else planes.shape[1], stream())
    return h


def gcl_forward_from_planes(a_planes, plan: Plan, d: int, w_frag, bias, col_stats=None, use_classes: bool = True):
    """`pm_gcl_forward_from_planes`: h = A' @ [W_t; W_4; W_5; root] + bias with the aggregate read from the planes
    `pm_segreduce_fwd_planes` wrote (int16 [3, N*4d]); the dense-graph path at d = 512."""
    N = plan.N
    h = torch.empty(N, d, dtype=F32, device=a_planes.device)
    call("pm_gcl_forward_from_planes", ptr(a_planes), a_planes.shape[1], ptr(plan.buf), N, plan.E, plan.G, d, ptr(w_frag),
         ptr(bias), 1 if use_classes else 0, ptr(h), ptr(col_stats), stream())
    return h


def gcl_input_grad_fused(dh_planes, plan: Plan, d: int, w_frag_t, use_classes: bool = True, out=None):
    """`pm_gcl_input_grad_fused`: dA' [N, 4d] = dh @ [W_t; W_4; W_5; root]^T per track group, A-stationary;
    `dh_planes` int16 [3, N*d] (`split_planes(dh)`), `w_frag_t` = `split_planes_frag(W, 0)`."""
    N = plan.N
    dA = out if out is not None else torch.empty(N, 4 * d, dtype=F32, device=dh_planes.device)
    call("pm_gcl_input_grad_fused", ptr(dh_planes), dh_planes.shape[1], ptr(plan.buf), N, plan.E, plan.G, d, ptr(w_frag_t),
         1 if use_classes else 0, ptr(dA), stream())
    return dA


class _BnBwd(ctypes.Structure):          # PmBnBwd (include/polyphemus_hip.h)
    _fields_ = [(k, ctypes.c_void_p) for k in ("h", "du", "mean", "var", "gamma", "beta", "acc3", "dgamma", "dbeta", "dbias_pre")] + \
               [("eps", ctypes.c_float), ("relu", ctypes.c_int32), ("add_residual", ctypes.c_int32), ("reserved", ctypes.c_int32)]


def bn_bwd_sums(h, du, mean, var, gamma, beta, eps: float = 1e-5, relu: bool = True):
    """`pm_bn_bwd_sums`: the three column sums of the norm backward, fp64 [PM_BN_REPL, 3, C]."""
    O, Cc = h.shape
    acc3 = torch.zeros(8, 3, Cc, dtype=torch.float64, device=h.device)
    call("pm_bn_bwd_sums", ptr(h), ptr(du), O, Cc, ptr(mean), ptr(var), float(eps), ptr(gamma), ptr(beta), 1 if relu else 0,
         ptr(acc3), stream())
    return acc3


def gcl_input_grad_bn(h, du, mean, var, gamma, beta, acc3, plan: Plan, w_frag_t, dgamma=None, dbeta=None, dbias_pre=None,
                      eps: float = 1e-5, relu: bool = True, use_classes: bool = True, add_residual: bool = False):
    """`pm_gcl_input_grad_bn`: the norm backward inside the input gradient; returns (dA' [N, 4d], dh planes int16 [3, N*d])."""
    N, d = h.shape
    dA = torch.empty(N, 4 * d, dtype=F32, device=h.device)
    planes = torch.empty(3, N * d, dtype=torch.int16, device=h.device)
    nb = _BnBwd(ptr(h), ptr(du), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(acc3), ptr(dgamma), ptr(dbeta), ptr(dbias_pre),
                float(eps), 1 if relu else 0, 1 if add_residual else 0, 0)
    call("pm_gcl_input_grad_bn", ctypes.addressof(nb), ptr(planes), planes.shape[1], ptr(plan.buf), N, plan.E, plan.G, d,
         ptr(w_frag_t), 1 if use_classes else 0, ptr(dA), stream())
    return dA, planes


def gcl_weight_grad_fused(a_planes, dh_planes, plan: Plan, d: int, dW, use_classes: bool = True):
    """`pm_gcl_weight_grad_fused`: dW [7d, d] += A'^T dh per track group (stacked [W_t; W_4; W_5; root] rows);
    `a_planes` int16 [3, N*4d], `dh_planes` int16 [3, N*d]."""
    _chk(dW, F32, "dW")
    call("pm_gcl_weight_grad_fused", ptr(a_planes), a_planes.shape[1], ptr(dh_planes), dh_planes.shape[1], ptr(plan.buf),
         plan.N, plan.E, plan.G, d, 1 if use_classes else 0, ptr(dW), stream())
    return dW


def segreduce_fwd(x, T, plan: Plan, dropout_p: float, seed: int, layer_uid: int, out=None):
    _chk(x, F32, "x"); _chk(T, F32, "T")
    N, d = x.shape
    A = out if out is not None else torch.empty(N, 7 * d, dtype=F32, device=x.device)
    e0 = _prof_begin("segreduce_fwd")
    call("pm_segreduce_fwd", ptr(x), ptr(T), ptr(plan.buf), N, plan.E, plan.G, d, float(dropout_p),
         seed & 0xFFFFFFFF, layer_uid, 0, ptr(A), stream())
    _prof_end("segreduce_fwd", e0, 4.0 * d * N * (1 + C.N_EDGE_TYPES) + 12.0 * plan.E)   # algorithmic HBM bytes
    return A


def segreduce_bwd(x, T, dA, dres, plan: Plan, dropout_p: float, seed: int, layer_uid: int, dT, out=None):
    _chk(x, F32, "x"); _chk(dA, F32, "dA"); _chk(dres, F32, "dres", allow_none=True); _chk(dT, F32, "dT")
    N, d = x.shape
    dx = out if out is not None else torch.empty(N, d, dtype=F32, device=x.device)
    call("pm_segreduce_bwd", ptr(x), ptr(T), ptr(dA), ptr(dres), ptr(plan.buf), N, plan.E, plan.G, d,
         float(dropout_p), seed & 0xFFFFFFFF, layer_uid, 0, ptr(dx), ptr(dT), stream())
    return dx


GEMM_RELU, GEMM_ACCUM, GEMM_PARTITION, GEMM_RELU_ADD = 1, 2, 4, 16


def bn_fold_weights(W, bias, gamma, beta, running_mean, running_var, eps=1e-5):
    """(W * s, bias * s + t) with s = gamma / sqrt(running_var + eps), t = beta - running_mean * s (columns of W)."""
    W2 = W.reshape(-1, W.shape[-1])
    Wo, bo = torch.empty_like(W2), torch.empty_like(bias)
    call("pm_bn_fold_weights", ptr(W2), W2.shape[0], W2.shape[1], ptr(bias), ptr(gamma), ptr(beta), ptr(running_mean),
         ptr(running_var), eps, ptr(Wo), ptr(bo), stream())
    return Wo, bo


def gemm(A, B, out, M, N, K, lda, ldb, ldc, transA=False, transB=False, bias=None, relu=False, accum=False,
         split_k=1, rowmap=None, rows_per_entry=0, dyn_entries=None, relu_add=False):
    """out[M,N] (=|+=) op(A) op(B) (+bias)(relu); A/B/out may be views with an element offset
    (pass the sliced tensor: its data_ptr() carries the offset) and explicit leading dimensions.
    relu_add: out = out + relu(op(A) op(B) + bias)."""
    flags = (GEMM_RELU if relu else 0) | (GEMM_ACCUM if accum else 0) | (GEMM_RELU_ADD if relu_add else 0)
    cls = gemm_class(transA, transB, M, N, K) if PROF is not None else ""
    e0 = _prof_begin(cls)
    call("pm_gemm_f32", int(transA), int(transB), M, N, K, ptr(A), lda, ptr(B), ldb, ptr(out), ldc, ptr(bias),
         flags, split_k, ptr(rowmap), rows_per_entry, ptr(dyn_entries), stream())
    _prof_end(cls, e0, 2.0 * M * N * K)                                                  # algorithmic flops
    return out


class _GemmDesc(ctypes.Structure):
    """ctypes mirror of PmGemmDesc (include/polyphemus_hip.h)."""
    _fields_ = [("transA", ctypes.c_int32), ("transB", ctypes.c_int32), ("M", ctypes.c_int32), ("N", ctypes.c_int32),
                ("K", ctypes.c_int32), ("A", ctypes.c_void_p), ("lda", ctypes.c_int32), ("B", ctypes.c_void_p),
                ("ldb", ctypes.c_int32), ("C", ctypes.c_void_p), ("ldc", ctypes.c_int32), ("bias", ctypes.c_void_p),
                ("flags", ctypes.c_int32), ("split_k", ctypes.c_int32), ("rowmap", ctypes.c_void_p),
                ("rows_per_entry", ctypes.c_int32), ("dyn_entries", ctypes.c_void_p), ("n_groups", ctypes.c_int32),
                ("a_group_stride", ctypes.c_int64), ("b_group_stride", ctypes.c_int64),
                ("c_group_stride", ctypes.c_int64), ("bias_group_stride", ctypes.c_int64),
                ("map_group_stride", ctypes.c_int32), ("dyn_group_stride", ctypes.c_int32),
                ("b_split_rows", ctypes.c_int32), ("b_shared_off", ctypes.c_int64),
                ("c_split_rows", ctypes.c_int32), ("c_shared_off", ctypes.c_int64), ("col_stats", ctypes.c_void_p),
                ("operand_planes", ctypes.c_int32), ("a_plane_stride", ctypes.c_int64),
                ("b_plane_stride", ctypes.c_int64), ("class_ptr", ctypes.c_void_p), ("class_block", ctypes.c_int32),
                ("b_frag", ctypes.c_void_p), ("a_colsum", ctypes.c_void_p)]


def gemm_desc(A, B, out, M, N, K, lda, ldb, ldc, transA=False, transB=False, bias=None, relu=False, accum=False,
              split_k=1, rowmap=None, rows_per_entry=0, dyn_entries=None, n_groups=1, a_group_stride=0,
              b_group_stride=0, c_group_stride=0, bias_group_stride=0, map_group_stride=0, dyn_group_stride=0,
              b_split_rows=0, b_shared_off=0, c_split_rows=0, c_shared_off=0, partition=False, col_stats=None, planes=False, a_plane_stride=0,
              b_plane_stride=0, class_ptr=None, class_block=0, b_frag=None, a_colsum=None):
    """Grouped / stacked-operand GEMM (`pm_gemm_f32_desc`): see PmGemmDesc in the header."""
    q = _GemmDesc(int(transA), int(transB), M, N, K, ptr(A), lda, ptr(B), ldb, ptr(out), ldc, ptr(bias),
                  (GEMM_RELU if relu else 0) | (GEMM_ACCUM if accum else 0) | (GEMM_PARTITION if partition else 0),
                  split_k, ptr(rowmap), rows_per_entry,
                  ptr(dyn_entries), n_groups, a_group_stride, b_group_stride, c_group_stride, bias_group_stride,
                  map_group_stride, dyn_group_stride, b_split_rows, b_shared_off, c_split_rows, c_shared_off, ptr(col_stats), int(planes), a_plane_stride, b_plane_stride, ptr(class_ptr), class_block,
                  ptr(b_frag), ptr(a_colsum))
    call("pm_gemm_f32_desc", ctypes.addressof(q), stream())
    return out


def split_planes(x: torch.Tensor) -> torch.Tensor:
    """fp32 tensor -> int16 [3, numel] holding the three bf16 planes with x = p1 + p2 + p3 exactly
    (`pm_split_planes`; the operand format of `gemm_desc(..., planes=True)`)."""
    _chk(x, F32, "x")
    n = x.numel()
    out = torch.empty(3, n, dtype=torch.int16, device=x.device)
    call("pm_split_planes", ptr(x), n, ptr(out), n, stream())
    return out


def split_planes_frag(W: torch.Tensor, kind: int) -> torch.Tensor:
    """Fragment-major bf16 planes of weight matrices W [n_mats, rows, cols] or [rows, cols] (`pm_split_planes_frag`;
    the `b_frag` operand of `gemm_desc`): kind 0 for a product that uses W transposed (transB), 1 otherwise."""
    _chk(W, F32, "W")
    mats = W.shape[0] if W.dim() == 3 else 1
    rows, cols = W.shape[-2:]
    out = torch.empty(mats, rows * cols * 3, dtype=torch.int16, device=W.device)
    call("pm_split_planes_frag", ptr(W), rows, cols, kind, mats, rows * cols, rows * cols * 3, ptr(out), stream())
    return out


def linear(x, weight, bias=None, relu=False, out=None):
    """y = x @ weight.T + bias for contiguous x [M,K], weight [N,K] (nn.Linear layout)."""
    M, K = x.shape
    N = weight.shape[0]
    y = out if out is not None else torch.empty(M, N, dtype=F32, device=x.device)
    return gemm(x, weight, y, M, N, K, x.stride(0), weight.stride(0), y.stride(0), transB=True, bias=bias, relu=relu)


def bn_scratch(C_: int, device) -> torch.Tensor:
    return torch.empty(256 * 3 * C_ + 4 * C_, dtype=F64, device=device)          # PM_BN_SCRATCH(C)


def bn_stats(x, O, C_, I, running_mean=None, running_var=None, momentum=0.1, scratch=None):
    mean = torch.empty(C_, dtype=F32, device=x.device)
    var = torch.empty(C_, dtype=F32, device=x.device)
    scratch = scratch if scratch is not None else bn_scratch(C_, x.device)
    call("pm_bn_stats", ptr(x), O, C_, I, ptr(mean), ptr(var), ptr(running_mean), ptr(running_var), momentum,
         ptr(scratch), stream())
    return mean, var


def bn_apply(x, O, C_, I, mean, var, gamma, beta, eps=1e-5, residual=None, relu=False, out=None):
    y = out if out is not None else torch.empty_like(x)
    call("pm_bn_apply", ptr(x), O, C_, I, ptr(mean), ptr(var), eps, ptr(gamma), ptr(beta), ptr(residual), int(relu),
         ptr(y), stream())
    return y


def bn_bwd(x, dy, O, C_, I, mean, var, gamma, beta, dgamma, dbeta, eps=1e-5, relu=False, out=None, scratch=None,
           dbias_pre=None):
    dx = out if out is not None else torch.empty_like(x)
    scratch = scratch if scratch is not None else bn_scratch(C_, x.device)
    call("pm_bn_bwd", ptr(x), ptr(dy), O, C_, I, ptr(mean), ptr(var), eps, ptr(gamma), ptr(beta), int(relu),
         ptr(dgamma), ptr(dbeta), ptr(dbias_pre), ptr(dx), ptr(scratch), stream())
    return dx


# ---- split forms for synchronised BatchNorm (data parallel, statistics over the global batch) -----------------------------
def bn_partial_sums(x, O, C_, I, dy=None, mean=None, var=None, gamma=None, beta=None, eps=1e-5, relu=False):
    """[3][C] fp64 column sums of this rank: {sum x, sum x^2, 0}, or with dy {sum du, sum du*xhat, sum xhat}."""
    sums = torch.empty(3, C_, dtype=F64, device=x.device)
    call("pm_bn_partial_sums", ptr(x), ptr(dy), O, C_, I, ptr(mean), ptr(var), eps, ptr(gamma), ptr(beta), int(relu),
         ptr(sums), ptr(bn_scratch(C_, x.device)), stream())
    return sums


def bn_stats_from_sums(sums, count, C_, running_mean=None, running_var=None, momentum=0.1):
    mean = torch.empty(C_, dtype=F32, device=sums.device)
    var = torch.empty(C_, dtype=F32, device=sums.device)
    call("pm_bn_stats_from_sums", ptr(sums), float(count), C_, ptr(mean), ptr(var), ptr(running_mean), ptr(running_var),
         momentum, stream())
    return mean, var


def bn_bwd_from_sums(x, dy, O, C_, I, mean, var, gamma, beta, sums_local, sums_global, count_global, dgamma, dbeta, eps=1e-5,
                     relu=False, dbias_pre=None):
    dx = torch.empty_like(x)
    scratch = torch.empty(2 * C_, dtype=F64, device=x.device)
    call("pm_bn_bwd_from_sums", ptr(x), ptr(dy), O, C_, I, ptr(mean), ptr(var), eps, ptr(gamma), ptr(beta), int(relu),
         ptr(sums_local), ptr(sums_global), float(count_global), ptr(dgamma), ptr(dbeta), ptr(dbias_pre), ptr(dx),
         ptr(scratch), stream())
    return dx


def relu_bwd(dy, y, out=None):
    dx = out if out is not None else torch.empty_like(dy)
    call("pm_relu_bwd", ptr(dy), ptr(y), dy.numel(), ptr(dx), stream())
    return dx


def relu_residual(x, res=None):
    y = torch.empty_like(x)
    call("pm_relu_residual_fwd", ptr(x), ptr(res), x.numel(), ptr(y), stream())
    return y


def dropout_rows(x, cols: int, p: float, seed: int, site: int, out=None):
    """x viewed as [numel / cols, cols] times the counter-hash keep mask of (seed, site) / (1 - p); contiguous input."""
    if not x.is_contiguous():
        raise ValueError("dropout_rows needs a contiguous tensor")
    y = out if out is not None else torch.empty_like(x)
    call("pm_dropout_rows", ptr(x), x.numel() // cols, cols, float(p), seed & 0xFFFFFFFF, site, ptr(y), stream())
    return y


def add(a, b, out=None):
    o = out if out is not None else torch.empty_like(a)
    call("pm_add", ptr(a), ptr(b), a.numel(), ptr(o), stream())
    return o


def grad_accumulate(grads, accum, scale: float, first: bool):
    """accum = (0 if first else accum) + scale * grads (training.py:149,158)."""
    _chk(grads, F32, "grads"); _chk(accum, F32, "accum")
    if grads.numel() != accum.numel():
        raise ValueError("grads / accum size mismatch")
    call("pm_grad_accumulate", ptr(grads), ptr(accum), grads.numel(), float(scale), int(bool(first)), stream())
    return accum


def colsum_acc(x, M, C_, ld, out):
    call("pm_colsum_acc", ptr(x), M, C_, ld, ptr(out), stream())


def colsum_rows_acc(x, C_, ld, rowmap, rows_per_entry, dyn_entries, max_entries, out):
    call("pm_colsum_rows_acc", ptr(x), C_, ld, ptr(rowmap), rows_per_entry, ptr(dyn_entries), max_entries, ptr(out),
         stream())


def reparam_fwd(mu, log_var, eps):
    z = torch.empty_like(mu)
    call("pm_reparam_fwd", ptr(mu), ptr(log_var), ptr(eps), mu.numel(), ptr(z), stream())
    return z


def reparam_bwd(dz, log_var, eps, dmu, dlog_var):
    call("pm_reparam_bwd", ptr(dz), ptr(log_var), ptr(eps), dz.numel(), ptr(dmu), ptr(dlog_var), stream())


def gate_fwd(x, w, b):
    N, d = x.shape
    g = torch.empty(N, dtype=F32, device=x.device)
    call("pm_gate_fwd", ptr(x), ptr(w), ptr(b), N, d, ptr(g), stream())
    return g


def attnpool_fwd(x, g, g_mean, g_var, bn_g, bn_b, plan: Plan, eps=1e-5):
    N, d = x.shape
    alpha = torch.empty(N, dtype=F32, device=x.device)
    out = torch.empty(plan.G, d, dtype=F32, device=x.device)
    call("pm_attnpool_fwd", ptr(x), ptr(g), ptr(g_mean), ptr(g_var), eps, ptr(bn_g), ptr(bn_b), ptr(plan.buf), N,
         plan.E, plan.G, d, ptr(alpha), ptr(out), stream())
    return alpha, out


def attnpool_bwd(x, g, g_mean, g_var, bn_g, alpha, dout, gate_w, plan: Plan, d_gate_w, d_gate_b, d_bn_g, d_bn_b,
                 eps=1e-5, x_gate=None):
    """Returns dx, or (dx, dx_gate) when the gate MLP saw `x_gate` instead of x (dropout in front of its Linear)."""
    N, d = x.shape
    dx = torch.empty_like(x)
    dxg = torch.empty_like(x) if x_gate is not None else None
    scratch = torch.empty(3 * N + 8, dtype=F32, device=x.device)
    call("pm_attnpool_bwd", ptr(x), ptr(g), ptr(g_mean), ptr(g_var), eps, ptr(bn_g), ptr(alpha), ptr(dout),
         ptr(gate_w), ptr(plan.buf), N, plan.E, plan.G, d, ptr(dx), ptr(d_gate_w), ptr(d_gate_b), ptr(d_bn_g),
         ptr(d_bn_b), ptr(scratch), ptr(x_gate), ptr(dxg), stream())
    return dx if x_gate is None else (dx, dxg)


def attnpool_bwd_sync(x, g, g_mean, g_var, bn_g, alpha, dout, gate_w, plan: Plan, d_gate_w, d_gate_b, d_bn_g, d_bn_b, reduce_,
                      count_global, eps=1e-5, x_gate=None):
    """`attnpool_bwd` with the BatchNorm1d(1) of the gate synchronised over the ranks: `reduce_(t)` sums a tensor in place
    over the ranks, `count_global` is the global node count."""
    N, d = x.shape
    dx = torch.empty_like(x)
    dxg = torch.empty_like(x) if x_gate is not None else None
    scratch = torch.empty(3 * N + 8, dtype=F32, device=x.device)
    call("pm_attnpool_bwd_sums", ptr(x), ptr(g), ptr(g_mean), ptr(g_var), eps, ptr(alpha), ptr(dout), ptr(plan.buf), N,
         plan.E, plan.G, d, ptr(scratch), stream())
    gs = scratch[:4].view(F64).clone()                     # the two local sums (fp64 at the head of the scratch)
    reduce_(gs)
    call("pm_attnpool_bwd_from_sums", ptr(x), ptr(g), ptr(g_mean), ptr(g_var), eps, ptr(bn_g), ptr(alpha), ptr(dout),
         ptr(gate_w), ptr(plan.buf), N, plan.E, plan.G, d, ptr(dx), ptr(d_gate_w), ptr(d_gate_b), ptr(d_bn_g), ptr(d_bn_b),
         ptr(scratch), ptr(x_gate), ptr(dxg), ptr(gs), float(count_global), stream())
    return dx if x_gate is None else (dx, dxg)


def bar_broadcast_fwd(bars, plan: Plan):
    d = bars.shape[1]
    x = torch.empty(plan.N, d, dtype=F32, device=bars.device)
    call("pm_bar_broadcast_fwd", ptr(bars), ptr(plan.buf), plan.N, plan.E, plan.G, d, ptr(x), stream())
    return x


def bar_broadcast_bwd(dx, plan: Plan):
    d = dx.shape[1]
    db = torch.empty(plan.G, d, dtype=F32, device=dx.device)
    call("pm_bar_broadcast_bwd", ptr(dx), ptr(plan.buf), plan.N, plan.E, plan.G, d, ptr(db), stream())
    return db


def conv3x3_fwd(x, w, b, G, Ci, Co, H, W, up4=False):
    y = torch.empty(G, Co, H, W, dtype=F32, device=x.device)
    call("pm_conv3x3_fwd", ptr(x), ptr(w), ptr(b), G, Ci, Co, H, W, int(up4), ptr(y), stream())
    return y


def conv3x3_bwd_data(dy, w, G, Ci, Co, H, W, up4=False):
    dx = torch.empty(G, Ci, H, W // 4 if up4 else W, dtype=F32, device=dy.device)
    call("pm_conv3x3_bwd_data", ptr(dy), ptr(w), G, Ci, Co, H, W, int(up4), ptr(dx), stream())
    return dx


def conv3x3_bwd_weight(x, dy, G, Ci, Co, H, W, dw, db, up4=False):
    call("pm_conv3x3_bwd_weight", ptr(x), ptr(dy), G, Ci, Co, H, W, int(up4), ptr(dw), ptr(db), stream())


def maxpool4_fwd(x):
    y = torch.empty(*x.shape[:-1], x.shape[-1] // 4, dtype=F32, device=x.device)
    call("pm_maxpool4_fwd", ptr(x), y.numel(), ptr(y), stream())
    return y


def maxpool4_bwd(x, dy):
    dx = torch.empty_like(x)
    call("pm_maxpool4_bwd", ptr(x), ptr(dy), dy.numel(), ptr(dx), stream())
    return dx


def content_ce(c_logits, plan: Plan, grad_scale=1.0, want_grad=True, out=None, dbias=None):
    """dbias = (d_bias_pitch_drum [131], d_bias_pitch_non_drum [131], d_bias_dur [99]) accumulates the
    un-embedding bias gradients inside the same pass (needs want_grad)."""
    N = c_logits.shape[0]
    out = out if out is not None else torch.empty(4, dtype=F64, device=c_logits.device)
    dl = torch.empty_like(c_logits) if want_grad else None
    b = dbias if dbias is not None else (None, None, None)
    call("pm_content_ce", ptr(c_logits), ptr(plan.tokens), ptr(plan.tok_hist), ptr(plan.is_drum), N, C.N_SLOTS, grad_scale,
         ptr(dl), ptr(b[0]), ptr(b[1]), ptr(b[2]), ptr(out), stream())
    return out, dl


def unembed_ce(H, w_pd, b_pd, w_pnd, b_pnd, w_dur, b_dur, plan: Plan, grad_scale=1.0, want_logits=False, out=None,
               dbias=None, dev_scale=None, planes=True):
    """Fused un-embedding + cross-entropy (`pm_unembed_ce`): H [N,15,d] -> (out, d_logits [N,15,230], logits or None)."""
    N, S, d = H.shape
    out = out if out is not None else torch.empty(4, dtype=F64, device=H.device)
    dl = torch.empty(N, S, C.D_TOKEN_PAIR, dtype=F32, device=H.device)
    lg = torch.empty_like(dl) if want_logits else None
    b = dbias if dbias is not None else (None, None, None)
    wpl = torch.empty(int(lib().pm_unembed_scratch_bytes(d)), dtype=torch.uint8, device=H.device) if planes else None
    call("pm_unembed_ce", ptr(H), ptr(w_pd), ptr(b_pd), ptr(w_pnd), ptr(b_pnd), ptr(w_dur), ptr(b_dur), ptr(plan.tokens),
         ptr(plan.buf), N, plan.E, plan.G, d, S, grad_scale, ptr(dev_scale), ptr(lg), ptr(dl), ptr(b[0]), ptr(b[1]), ptr(b[2]),
         ptr(out), ptr(wpl), stream())
    return out, dl, lg


def kld(mu, log_var, out, beta=0.0, dmu=None, dlog_var=None):
    B, d = mu.shape
    call("pm_kld", ptr(mu), ptr(log_var), B, d, beta, ptr(dmu), ptr(dlog_var), ptr(out), stream())
    return out


def bce_logits(logits, target, out, grad_scale=1.0, want_grad=False):
    dl = torch.empty_like(logits) if want_grad else None
    call("pm_bce_logits", ptr(logits), ptr(target), logits.numel(), grad_scale, ptr(dl), ptr(out), stream())
    return out, dl


def content_accuracy(c_logits, tokens, is_drum):
    """Counts of `_accuracies` (training.py:349-468) on the device: int64 [8] =
    {pitch correct, pitch not-PAD, pitch correct (drums), pitch not-PAD (drums), dur correct, dur not-PAD, note correct, 0}."""
    _chk(c_logits, F32, "c_logits"); _chk(tokens, I32, "tokens")
    drum = is_drum.view(U8) if is_drum.dtype == torch.bool else is_drum
    out = torch.empty(8, dtype=I64, device=c_logits.device)
    call("pm_content_accuracy", ptr(c_logits), ptr(tokens), ptr(drum.contiguous()), c_logits.shape[0], ptr(out), stream())
    return out


def structure_metrics(s_logits, s_target):
    """int64 [4] = {prediction == target, true positives, predicted positives, target positives} (training.py:470-497)."""
    _chk(s_logits, F32, "s_logits"); _chk(s_target, F32, "s_target")
    out = torch.empty(4, dtype=I64, device=s_logits.device)
    call("pm_structure_metrics", ptr(s_logits), ptr(s_target), s_logits.numel(), ptr(out), stream())
    return out


def adam_step(params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, grad_scale=1.0):
    for t, n in ((params, "params"), (grads, "grads"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _chk(t, F32, n)
    call("pm_adam_step", ptr(params), ptr(grads), ptr(exp_avg), ptr(exp_avg_sq), params.numel(), lr, beta1, beta2,
         eps, step, grad_scale, stream())
