"""Host-side bar-graph construction and the batch container of the hot path.

This is the *input contract* of the graph-VAE path: it stays
on the host, exactly as in the reference (`data.py:24-204`), and is restated here
in numpy so that synthetic batches and `Decoder._structure_from_binary`
(reference `model.py:596-607`) do not need torch_geometric.

Edge order is reproduced bit-for-bit (tests/test_graphs.py checks it against
vectors captured from the reference's own `graph_from_tensor`):
per bar: track edges (per track: forward list, then inverse list), onset edges
(per timestep: pair list, then inverse list), next edges (forward only).
"""
from __future__ import annotations

import itertools
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import constants as C


def bar_edges(bar: np.ndarray):
    """Edges of ONE bar. `bar` is a [4,32] 0/1 array with >=1 active cell.

    Returns (src, dst, etype, dist) int64 arrays in the reference's order and the
    node count.  Follows data.py:24-51 (track), :54-80 (onset), :83-121 (next),
    :162-176 (self-loop for an edgeless bar)."""
    bar = np.asarray(bar).astype(bool)
    tr, ts = np.nonzero(bar)                       # row-major == (track, timestep) order
    n = tr.shape[0]
    label = np.zeros(bar.shape, dtype=np.int64)
    label[tr, ts] = np.arange(n)
    src: List[int] = []
    dst: List[int] = []
    typ: List[int] = []
    dis: List[int] = []

    def emit(fwd):
        for (u, v, t, d) in fwd:
            src.append(u); dst.append(v); typ.append(t); dis.append(d)
        for (u, v, t, d) in fwd:                   # inverse list follows the forward list
            src.append(v); dst.append(u); typ.append(t); dis.append(d)

    for track in range(bar.shape[0]):              # data.py:36-49
        tss = ts[tr == track]
        fwd = [(label[track, t1], label[track, t2], C.EDGE_TRACK + track, t2 - t1)
               for t1, t2 in zip(tss[:-1], tss[1:])]
        emit(fwd)
    for t in range(bar.shape[1]):                  # data.py:67-78
        tracks = tr[ts == t]
        fwd = [(label[a, t], label[b, t], C.EDGE_ONSET, 0)
               for a, b in itertools.combinations(tracks, 2)]
        emit(fwd)
    act = np.nonzero(bar.any(axis=0))[0]           # data.py:96-119
    if act.shape[0] > 1:
        for t1, t2 in zip(act[:-1], act[1:]):
            for a in tr[ts == t1]:
                for b in tr[ts == t2]:
                    if a != b:
                        src.append(label[a, t1]); dst.append(label[b, t2])
                        typ.append(C.EDGE_NEXT); dis.append(t2 - t1)
    if not src:                                    # data.py:173-176: fake self loop, type 0
        src, dst, typ, dis = [0], [0], [0], [0]
    return (np.asarray(src, np.int64), np.asarray(dst, np.int64),
            np.asarray(typ, np.int64), np.asarray(dis, np.int64), n)


def dense_bar_edges():
    """BASELINE config 5 'dense stress' bar: all 128 cells active, every ordered
    pair u != v is an edge: type = track if same track, 4 if same
    timestep, else 5; distance = |dt|.  Not producible by `bar_edges`."""
    n = C.N_TRACKS * C.N_TIMESTEPS
    u, v = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    keep = u != v
    u, v = u[keep], v[keep]
    tu, su = u // C.N_TIMESTEPS, u % C.N_TIMESTEPS
    tv, sv = v // C.N_TIMESTEPS, v % C.N_TIMESTEPS
    typ = np.where(tu == tv, tu, np.where(su == sv, C.EDGE_ONSET, C.EDGE_NEXT))
    return (u.astype(np.int64), v.astype(np.int64), typ.astype(np.int64),
            np.abs(su - sv).astype(np.int64), n)


class BarGraphBatch:
    """Disjoint union of the bar graphs of B samples (duck-types the reference's
    PyG `Batch`).

    Compact attributes (what the HIP path consumes):
      edge_index i64 [2,E] · edge_type i32 [E] · edge_dist i32 [E] ·
      tokens i32 [N,16,2] (pitch id, duration id; slot 0 = SOS) ·
      s_tensor f32 [B*nb,4,32] · is_drum bool [N] · bars i64 [N] · batch i64 [N] ·
      num_nodes int · n_bars int.
    Reference-format attributes, materialised lazily (they are 14.7 KB/node):
      edge_attrs f32 [E,33] (data.py:179-182) · c_tensor f32 [N,16,230] (data.py:235-268).
    """

    def __init__(self, **kw):
        self.__dict__.update(kw)

    # -- reference-format views ------------------------------------------------
    @property
    def edge_attrs(self) -> torch.Tensor:
        if "_edge_attrs" not in self.__dict__:
            E = self.edge_type.numel()
            ea = torch.zeros(E, C.N_DISTS + 1, device=self.edge_type.device)
            ea[:, 0] = self.edge_type.float()
            ea[torch.arange(E, device=ea.device), self.edge_dist.long() + 1] = 1.0
            self.__dict__["_edge_attrs"] = ea
        return self.__dict__["_edge_attrs"]

    @property
    def c_tensor(self) -> torch.Tensor:
        if "_c_tensor" not in self.__dict__:
            tok = self.tokens.long()
            N = tok.shape[0]
            c = torch.zeros(N, C.MAX_SIMU_TOKENS, C.D_TOKEN_PAIR, device=tok.device)
            c.scatter_(2, tok[..., 0:1], 1.0)
            c.scatter_(2, tok[..., 1:2] + C.N_PITCH_TOKENS, 1.0)
            self.__dict__["_c_tensor"] = c
        return self.__dict__["_c_tensor"]

    def to(self, device, non_blocking: bool = False) -> "BarGraphBatch":
        out = {}
        for k, v in self.__dict__.items():
            out[k] = v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v
        return BarGraphBatch(**out)

    @property
    def num_graphs(self) -> int:
        return int(self.s_tensor.shape[0])


def batch_flags(tokens: torch.Tensor, edge_index: torch.Tensor, edge_type: torch.Tensor, num_nodes: int):
    """(n_slots, track_unique) of a batch that does not carry them (a foreign PyG batch, a golden fixture): the two
    host-known facts that select the measured path of the native step — active token slots (`PmBatch.n_slots`) and
    "every node receives track edges of at most one relation" (`PmBatch.flags` bit 0, compact GCL).  Same definitions as
    `collate_samples`; on device tensors this costs ONE host read per batch object (callers cache the result)."""
    if tokens.is_cuda and tokens.dtype == torch.int32 and tokens.is_contiguous() and edge_index.dtype == torch.int64 \
            and edge_index.is_contiguous() and edge_type.dtype == torch.int32 and edge_type.is_contiguous() and edge_type.numel() > 0:
        # on the device: two launches and the one host read (csrc/plan.hip pm_batch_flags); same definitions as below
        from ._lib import call, ptr, stream
        scratch = torch.zeros(int(num_nodes) + 3, dtype=torch.int32, device=tokens.device)
        call("pm_batch_flags", ptr(tokens), ptr(edge_index), ptr(edge_type), int(num_nodes), int(edge_type.numel()),
             ptr(scratch), ptr(scratch[int(num_nodes):]), stream())
        n_slots, multi, bad = scratch[int(num_nodes):].tolist()
        if bad:
            raise ValueError("batch holds token ids, edge types or node ids outside their ranges")
        return max(int(n_slots), 1), not bool(multi)
    tok = tokens.long()
    live = (tok[:, 1:, 0] != C.PITCH_PAD) | (tok[:, 1:, 1] != C.DUR_PAD)
    slot_live = live.any(dim=0)
    et = edge_type.long()
    trk = et < C.N_TRACKS
    seen = torch.zeros(max(int(num_nodes), 1), C.N_TRACKS, dtype=torch.bool, device=et.device)
    seen[edge_index[1].long()[trk], et[trk]] = True
    multi = (seen.sum(dim=1) > 1).any()
    last = (slot_live.long() * torch.arange(1, slot_live.numel() + 1, device=slot_live.device)).max()
    # ids index LDS / global tables inside the kernels: range-check them here (same host read)
    ei = edge_index.long()
    bad = ((tok[..., 0] < 0) | (tok[..., 0] >= C.N_PITCH_TOKENS) | (tok[..., 1] < 0) | (tok[..., 1] >= C.N_DUR_TOKENS)).any() | \
          ((et < 0) | (et >= C.N_EDGE_TYPES)).any() | ((ei < 0) | (ei >= int(num_nodes))).any()
    n_slots, multi, bad = torch.stack([last, multi.long(), bad.long()]).tolist()          # the one sync
    if bad:
        raise ValueError("batch holds token ids, edge types or node ids outside their ranges")
    return max(int(n_slots), 1), not bool(multi)


def graph_from_structure(s_tensor: np.ndarray, dense: bool = False):
    """One sample: `s_tensor` [nb,4,32] 0/1 -> dict of numpy arrays.

    Restates data.py:141-204 (per-bar graphs concatenated with node offsets,
    `bars` = bar id of every node).  Like the reference (data.py:148-153) an empty
    bar gets cell [0,0] switched on **in place**."""
    srcs, dsts, typs, diss, bars, drums = [], [], [], [], [], []
    off = 0
    for b in range(s_tensor.shape[0]):
        bar = s_tensor[b]
        if not bar.any():
            bar[0, 0] = 1
        if dense:
            u, v, t, d, n = dense_bar_edges()
        else:
            u, v, t, d, n = bar_edges(bar)
        srcs.append(u + off); dsts.append(v + off); typs.append(t); diss.append(d)
        tr, _ = np.nonzero(np.asarray(bar).astype(bool))
        drums.append(tr == 0)                      # data.py:184-185: track 0 = drums
        bars.append(np.full(n, b, np.int64))
        off += n
    return dict(src=np.concatenate(srcs), dst=np.concatenate(dsts),
                etype=np.concatenate(typs), edist=np.concatenate(diss),
                bars=np.concatenate(bars), is_drum=np.concatenate(drums),
                num_nodes=off)


def collate_samples(samples: Sequence[dict], n_bars: int) -> BarGraphBatch:
    """Batch of samples -> BarGraphBatch (what PyG's DataLoader collate does to the
    reference's `Data` objects).  Each sample dict has the keys of
    `graph_from_structure` plus `tokens` [n,16,2] and `s_tensor` [nb,4,32]."""
    off = 0
    src, dst, batch = [], [], []
    for i, s in enumerate(samples):
        src.append(s["src"] + off); dst.append(s["dst"] + off)
        batch.append(np.full(s["num_nodes"], i, np.int64))
        off += s["num_nodes"]
    cat = lambda k: np.concatenate([s[k] for s in samples])
    tok = cat("tokens")
    # active token slots: slot s (1..15) is active if any node holds a non-PAD token there; slots beyond the
    # last active one are PAD everywhere, which lets the fused step skip them (same losses and gradients)
    live = (tok[:, 1:, 0] != C.PITCH_PAD) | (tok[:, 1:, 1] != C.DUR_PAD)
    n_slots = int(np.nonzero(live.any(axis=0))[0].max()) + 1 if live.any() else 1
    # every node receives track edges (types 0..3) of at most ONE relation: true for graphs built by the
    # reference's rules; verified here because the compact GCL (K = 4d instead of 7d) relies on it
    et_all, dst_all = cat("etype"), np.concatenate(dst)
    trk = et_all < C.N_TRACKS
    pairs = np.unique(np.stack([dst_all[trk], et_all[trk]], 1), axis=0)
    track_unique = bool(pairs.shape[0] == np.unique(pairs[:, 0]).shape[0])
    return BarGraphBatch(
        n_slots=n_slots, track_unique=track_unique,
        edge_index=torch.from_numpy(np.stack([np.concatenate(src), np.concatenate(dst)])),
        edge_type=torch.from_numpy(cat("etype").astype(np.int32)),
        edge_dist=torch.from_numpy(cat("edist").astype(np.int32)),
        tokens=torch.from_numpy(cat("tokens").astype(np.int32)),
        s_tensor=torch.from_numpy(np.concatenate([s["s_tensor"] for s in samples]).astype(np.float32)),
        is_drum=torch.from_numpy(cat("is_drum")),
        bars=torch.from_numpy(cat("bars")),
        batch=torch.from_numpy(np.concatenate(batch)),
        num_nodes=off, n_bars=n_bars)


def device_batch_from_structure(s_tensor: torch.Tensor, n_bars: int, token_grid: Optional[torch.Tensor] = None) -> BarGraphBatch:
    """The batch of `collate_samples(graph_from_structure(...))`, built ON THE DEVICE from the activation grids
    (`ops.graph_build`, csrc/graph.hip): `s_tensor` [B*n_bars,4,32] (0/1, any dtype, cuda).
    `token_grid` (optional) is the dense per-cell payload [B*n_bars,4,32,16,2] of token ids; the nodes' rows are
    gathered from it.  Bit-identical to the host construction (tests/test_graphs_gpu.py)."""
    from . import ops
    s = s_tensor.to(torch.float32).contiguous().clone()
    g = ops.graph_build(s, n_bars)
    kw = dict(n_slots=15, track_unique=True, edge_index=g["edge_index"], edge_type=g["edge_type"],
              edge_dist=g["edge_dist"], s_tensor=s, is_drum=g["is_drum"], bars=g["bars"], batch=g["batch"],
              num_nodes=g["num_nodes"], n_bars=n_bars, node_cell=g["node_cell"])
    if token_grid is not None:
        kw["tokens"] = token_grid.reshape(-1, C.MAX_SIMU_TOKENS, 2).to(torch.int32)[g["node_cell"].long()].contiguous()
    return BarGraphBatch(**kw)
