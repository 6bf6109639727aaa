"""ctypes mirror of the native training-step API (include/polyphemus_hip.h, `pm_vae_step_*`):
the model layout (offsets of every parameter / running statistic in the flat buffers, named
after the reference's modules) and the batch descriptor."""
from __future__ import annotations

import ctypes as C
import os

import torch

from ._lib import lib

PM_MAX_LAYERS = 16


_PLANES = os.environ.get("PM_GCL_PLANES", "1") != "0"


class PmLin(C.Structure):
    _fields_ = [("w", C.c_int64), ("b", C.c_int64)]


class PmBn(C.Structure):
    _fields_ = [("w", C.c_int64), ("b", C.c_int64), ("rm", C.c_int64), ("rv", C.c_int64)]


class PmGcn(C.Structure):
    _fields_ = [("nn_w", C.c_int64), ("nn_b", C.c_int64), ("weight", C.c_int64 * PM_MAX_LAYERS),
                ("bias", C.c_int64 * PM_MAX_LAYERS), ("norm", PmBn * PM_MAX_LAYERS)]


# (field, kind, reference module path) in the order of struct PmVaeLayout
_LAYOUT = [
    ("enc_conv0", "lin", "encoder.s_encoder.cnn_encoder.conv.0"), ("enc_bn1", "bn", "encoder.s_encoder.cnn_encoder.conv.1"),
    ("enc_conv4", "lin", "encoder.s_encoder.cnn_encoder.conv.4"), ("enc_bn5", "bn", "encoder.s_encoder.cnn_encoder.conv.5"),
    ("enc_lin1", "lin", "encoder.s_encoder.cnn_encoder.lin.1"), ("enc_lin4", "lin", "encoder.s_encoder.cnn_encoder.lin.4"),
    ("enc_s_bars", "lin", "encoder.s_encoder.bars_encoder"),
    ("enc_pitch_nd", "lin", "encoder.c_encoder.non_drums_pitch_emb"), ("enc_pitch_d", "lin", "encoder.c_encoder.drums_pitch_emb"),
    ("enc_dur", "lin", "encoder.c_encoder.dur_emb"), ("enc_bn_nd", "bn", "encoder.c_encoder.bn_non_drums"),
    ("enc_bn_d", "bn", "encoder.c_encoder.bn_drums"), ("enc_bn_dur", "bn", "encoder.c_encoder.bn_dur"),
    ("enc_chord", "lin", "encoder.c_encoder.chord_encoder"), ("enc_gcn", "gcn", "encoder.c_encoder.graph_encoder"),
    ("enc_gate", "lin", "encoder.c_encoder.graph_attention.gate_nn.0.layers.0"),
    ("enc_gate_bn", "bn", "encoder.c_encoder.graph_attention.gate_nn.1"), ("enc_c_bars", "lin", "encoder.c_encoder.bars_encoder"),
    ("enc_merge", "lin", "encoder.linear_merge"), ("enc_bn_merge", "bn", "encoder.bn_linear_merge"),
    ("enc_mu", "lin", "encoder.linear_mu"), ("enc_lv", "lin", "encoder.linear_log_var"),
    ("dec_lin", "lin", "decoder.lin_decoder"), ("dec_bn", "bn", "decoder.batch_norm"),
    ("dec_s_bars", "lin", "decoder.s_decoder.bars_decoder"), ("dec_s_lin1", "lin", "decoder.s_decoder.cnn_decoder.lin.1"),
    ("dec_s_lin4", "lin", "decoder.s_decoder.cnn_decoder.lin.4"), ("dec_conv1", "lin", "decoder.s_decoder.cnn_decoder.conv.1"),
    ("dec_bn2", "bn", "decoder.s_decoder.cnn_decoder.conv.2"), ("dec_conv4", "lin", "decoder.s_decoder.cnn_decoder.conv.4"),
    ("dec_c_bars", "lin", "decoder.c_decoder.bars_decoder"), ("dec_gcn", "gcn", "decoder.c_decoder.graph_decoder"),
    ("dec_chord", "lin", "decoder.c_decoder.chord_decoder"), ("dec_pitch_d", "lin", "decoder.c_decoder.drums_pitch_emb"),
    ("dec_pitch_nd", "lin", "decoder.c_decoder.non_drums_pitch_emb"), ("dec_dur", "lin", "decoder.c_decoder.dur_emb"),
]
_KIND = {"lin": PmLin, "bn": PmBn, "gcn": PmGcn}


class PmVaeLayout(C.Structure):
    _fields_ = ([("d", C.c_int32), ("n_bars", C.c_int32), ("n_layers", C.c_int32), ("flags", C.c_int32)] +
                [(f, _KIND[k]) for f, k, _ in _LAYOUT] + [("dropout", C.c_float), ("reserved", C.c_int32)])


class PmBatch(C.Structure):
    _fields_ = [("edge_index", C.c_void_p), ("edge_type", C.c_void_p), ("edge_dist", C.c_void_p), ("bars", C.c_void_p),
                ("batch", C.c_void_p), ("is_drum", C.c_void_p), ("tokens", C.c_void_p), ("s_tensor", C.c_void_p),
                ("N", C.c_int32), ("E", C.c_int32), ("G", C.c_int32), ("B", C.c_int32), ("n_slots", C.c_int32),
                ("flags", C.c_int32), ("ce_scale", C.c_void_p)]


def build_layout(vae) -> PmVaeLayout:
    """Offsets (in floats) of every tensor of `vae` inside vae.flat_params / vae.flat_buffers."""
    po, bo, cfg = vae._offsets, vae._buf_offsets, vae.cfg
    if cfg["gnn_n_layers"] > PM_MAX_LAYERS:
        raise ValueError(f"native step supports at most {PM_MAX_LAYERS} GNN layers")
    lay = PmVaeLayout()
    lay.d, lay.n_bars, lay.n_layers = cfg["d"], cfg["n_bars"], cfg["gnn_n_layers"]
    has_bn = bool(cfg["batch_norm"])
    lay.flags = 0 if has_bn else 1
    lay.dropout = float(cfg["dropout"] or 0.0)
    # batch_norm = False: the norm layers of the GCN stacks and of the CNNs do not exist and the nn.Sequential indices of the
    # layers behind them shift (model.py:218-238,278-292)
    shift = {} if has_bn else {"encoder.s_encoder.cnn_encoder.conv.4": "encoder.s_encoder.cnn_encoder.conv.3",
                               "decoder.s_decoder.cnn_decoder.conv.4": "decoder.s_decoder.cnn_decoder.conv.3"}
    optional = ("encoder.s_encoder.cnn_encoder.conv.1", "encoder.s_encoder.cnn_encoder.conv.5",
                "decoder.s_decoder.cnn_decoder.conv.2")

    def bn(path):
        if not has_bn and (path in optional or ".norm_layers." in path):
            return PmBn(0, 0, 0, 0)
        return PmBn(po[path + ".weight"], po[path + ".bias"], bo[path + ".running_mean"], bo[path + ".running_var"])

    for field, kind, path in _LAYOUT:
        path = shift.get(path, path)
        if kind == "lin":
            setattr(lay, field, PmLin(po[path + ".weight"], po[path + ".bias"]))
        elif kind == "bn":
            setattr(lay, field, bn(path))
        else:
            g = PmGcn()
            g.nn_w, g.nn_b = po[path + ".layers.0.nn.weight"], po[path + ".layers.0.nn.bias"]
            for i in range(cfg["gnn_n_layers"]):
                g.weight[i], g.bias[i] = po[f"{path}.layers.{i}.weight"], po[f"{path}.layers.{i}.bias"]
                assert po[f"{path}.layers.{i}.root"] == g.weight[i] + 6 * cfg["d"] ** 2, "weight|root must be adjacent"
                g.norm[i] = bn(f"{path}.norm_layers.{i}.module")
            setattr(lay, field, g)
    L = lib()
    L.pm_vae_layout_bytes.restype = C.c_int64
    if L.pm_vae_layout_bytes() != C.sizeof(PmVaeLayout):
        raise RuntimeError("PmVaeLayout: ctypes mirror and C struct differ in size")
    return lay


def make_batch(edge_index, bars, batch, s_tensor, tokens, is_drum_u8, et, ed, n_slots: int, track_unique: bool,
               keep_logits: bool = False, ext_loss: bool = False) -> PmBatch:
    """Batch descriptor of the native step; every tensor is a checked device tensor of the dtype the C side reads
    (int64 edge_index / bars / batch, int32 ids, uint8 is_drum, fp32 s_tensor: `HipTrainer._prep_inputs`)."""
    for t, dt in ((edge_index, torch.int64), (bars, torch.int64), (batch, torch.int64), (s_tensor, torch.float32),
                  (tokens, torch.int32), (is_drum_u8, torch.uint8), (et, torch.int32), (ed, torch.int32)):
        if t.dtype != dt or not t.is_contiguous() or not t.is_cuda:
            raise ValueError(f"native step: expected a contiguous cuda {dt} tensor, got {t.dtype} on {t.device}")
    b = PmBatch()
    b.edge_index, b.edge_type, b.edge_dist = edge_index.data_ptr(), et.data_ptr(), ed.data_ptr()
    b.bars, b.batch, b.is_drum = bars.data_ptr(), batch.data_ptr(), is_drum_u8.data_ptr()
    b.tokens, b.s_tensor = tokens.data_ptr(), s_tensor.data_ptr()
    b.N, b.E, b.G = bars.shape[0], edge_index.shape[1], s_tensor.shape[0]
    if not 1 <= int(n_slots) <= 15:
        raise ValueError(f"n_slots must be in 1..15, got {n_slots}")
    b.n_slots = int(n_slots)
    # bit 0: one track relation per node (compact GCL); bit 1: GCL GEMM operands as pre-split bf16 planes
    # bit 2: the fused un-embedding + CE also stores the content logits (step_outputs, evaluation)
    # bit 3: the caller computes the loss (model(graph) + autograd): logits only, no loss kernels in the forward
    b.flags = (1 if track_unique else 0) | (2 if _PLANES else 0) | (4 if keep_logits else 0) | (8 if ext_loss else 0)
    return b


# --------------------------------------------------------------------------- the C++ step behind one object
def prepare_inputs(graph, want_token_counts: bool = False):
    """Compact device inputs of a batch — token ids, edge ids, the two host-known facts that pick the native path (active
    slots, compact GCL) — converted once and cached on the graph.  Accepts the compact `BarGraphBatch` attributes or the
    reference's PyG format (edge_attrs [E,33], one-hot c_tensor [N,16,230]; data.py:179-182,235-268)."""
    from . import ops
    c = graph.__dict__.get("_pm_inputs") if hasattr(graph, "__dict__") else None
    if c is not None and (not want_token_counts or len(c) > 9):
        return c
    has = lambda k: k in getattr(graph, "__dict__", {}) or (hasattr(graph, "keys") and k in graph.keys())
    if has("edge_type") and has("edge_dist"):
        et, ed = graph.edge_type.to(torch.int32).contiguous(), graph.edge_dist.to(torch.int32).contiguous()
    else:
        et, ed = ops.edge_attrs_to_ids(graph.edge_attrs.float().contiguous())
    tok = graph.tokens.to(torch.int32).contiguous() if has("tokens") else ops.tokens_from_onehot(
        graph.c_tensor.float().contiguous())
    drum = graph.is_drum.contiguous()
    drum = drum.view(torch.uint8) if drum.dtype == torch.bool else drum
    ei, bars, bat = (t.to(torch.int64).contiguous() for t in (graph.edge_index, graph.bars, graph.batch))
    for t in (et, ed, tok, drum, ei, bars, bat):
        if not t.is_cuda:
            raise RuntimeError("the native step needs the batch on the GPU (graph.to('cuda')); there is no CPU path")
    n_slots, unique = getattr(graph, "n_slots", None), getattr(graph, "track_unique", None)
    if n_slots is None or unique is None:
        from .graphs import batch_flags
        n_slots, unique = batch_flags(tok, ei, et, tok.shape[0])
    c = (et, ed, tok, drum, ei, bars, bat, int(n_slots), bool(unique))
    if want_token_counts:             # non-PAD (pitch, duration) tokens of slots 1..15, on the device
        nv = torch.stack([(tok[:, 1:, 0] != 130).sum(), (tok[:, 1:, 1] != 98).sum()]).to(torch.float32)
        c = c + (nv,)
    try:
        graph.__dict__["_pm_inputs"] = c
    except Exception:
        pass
    return c


class NativeStep:
    """One model's handle on the C++ training step (`pm_vae_step_*`, csrc/vae_step.hip): layout, host state blob, workspace
    arena and plan buffer; the calls of a step in order.  Used by `HipTrainer` (the step's own fused loss and Adam) and by
    the drop-in module's autograd bridge (`model.VAE.forward`: the caller's loss, `pm_vae_step_set_output_grads`)."""

    def __init__(self, vae):
        from ._lib import call, plan_layout, ptr, stream, PLAN_FIELDS
        self._call, self._plan_layout, self._ptr, self._stream, self._fields = call, plan_layout, ptr, stream, PLAN_FIELDS
        self.vae = vae
        self.layout = build_layout(vae)
        self.flat_ptr = vae.flat_params.data_ptr()
        self.state = C.create_string_buffer(int(lib().pm_vae_step_state_bytes()))
        self.ws = None
        self.plan_buf = None
        self.plan_off = None
        self.bt = None
        dev = vae.flat_params.device
        self.loss_buf = torch.zeros(4, dtype=torch.float64, device=dev)
        ci = vae._counter_index
        emb = ["encoder.c_encoder.bn_drums", "encoder.c_encoder.bn_non_drums", "encoder.c_encoder.bn_dur"]
        inc = torch.ones(vae.flat_counters.numel(), dtype=torch.int64, device=dev)
        sel = torch.zeros(2, vae.flat_counters.numel(), dtype=torch.int64, device=dev)
        for k, row in zip(emb, ((1, 0), (0, 1), (1, 1))):
            i = ci[k + ".num_batches_tracked"]
            inc[i] = 0
            sel[0, i], sel[1, i] = row
        self._nbt_inc, self._nbt_sel = inc, sel

    @property
    def addr(self) -> int:
        return C.addressof(self.state)

    def forward(self, graph, eps, grads, keep_logits: bool, beta: float = 0.0, fix_structure: bool = False,
                n_slots=None, ce_scale=None, want_token_counts: bool = False, ext_loss: bool = False):
        """plan -> encoder -> reparametrisation -> decoder -> the step's own losses into `self.loss_buf` (+ their gradients
        with respect to the outputs; the bias gradients of the un-embedding land in `grads`).  `n_slots`: token slots the
        decoder head covers (default: the batch's active slots; 15 = every slot, as the reference computes them)."""
        vae, L = self.vae, lib()
        if vae.flat_params.data_ptr() != self.flat_ptr:
            raise RuntimeError("the model's flat parameter buffer moved after the native step was built; rebuild it")
        inputs = prepare_inputs(graph, want_token_counts)
        et, ed, tok, drum, ei, bars, bat, act_slots, unique = inputs[:9]
        s_tensor = graph.s_tensor
        if s_tensor.dtype != torch.float32 or not s_tensor.is_contiguous():
            s_tensor = s_tensor.float().contiguous()
        s_tensor = s_tensor.reshape(-1, 4, 32)
        bt = make_batch(ei, bars, bat, s_tensor, tok, drum, et, ed, act_slots if n_slots is None else n_slots, unique,
                        keep_logits=keep_logits, ext_loss=ext_loss)
        if ce_scale is not None:
            bt.ce_scale = ce_scale.data_ptr()
        bt.B = bt.G // vae.cfg["n_bars"]
        need = int(L.pm_vae_step_workspace_bytes(C.byref(self.layout), bt.N, bt.E, bt.G, bt.B, bt.n_slots))
        if self.ws is None or self.ws.numel() < need:
            self.ws = torch.empty(need, dtype=torch.uint8, device=grads.device)
        off = self._plan_layout(bt.N, bt.E, bt.G)
        if self.plan_buf is None or self.plan_buf.numel() < off[-1]:
            self.plan_buf = torch.empty(off[-1], dtype=torch.int32, device=grads.device)
        if eps is None:
            eps = torch.randn(bt.B, vae.cfg["d"], device=grads.device)
        self.bt, self.plan_off, self._keep = bt, off, (s_tensor, eps, grads, ce_scale, inputs)   # (alive until the backward is through)
        ptr = self._ptr
        self._call("pm_vae_step_forward", C.addressof(self.layout), ptr(vae.flat_params), ptr(vae.flat_buffers), ptr(grads),
                   C.addressof(bt), ptr(self.plan_buf), ptr(eps), float(vae.msg_dropout), vae._next_seed(), vae._next_seed(),
                   float(beta), int(fix_structure), ptr(self.ws), self.ws.numel(), self.addr, ptr(self.loss_buf), self._stream())
        return self.loss_buf

    def info(self) -> dict:
        info = (C.c_int32 * 16)()
        self._call("pm_vae_step_info", self.addr, C.cast(info, C.c_void_p))
        keys = ("compact", "planes", "n_slots", "b_frag", "N", "E", "G", "B", "fused_ce", "side_stream", "deterministic",
                "gcl_fused", "dagg_bn", "chord_tables", "h2", "pad_skip")
        return dict(zip(keys, (int(v) for v in info)))

    def outputs(self):
        """`((s_logits, c_logits), mu, log_var)` of the last forward; c_logits is [N, S, 230] for the S slots it covered."""
        i, vae = self.info(), self.vae
        dev, d, nb = vae.flat_params.device, vae.cfg["d"], vae.cfg["n_bars"]
        s_logits = torch.empty(i["B"], nb, 4, 32, device=dev)
        c_logits = torch.empty(i["N"], i["n_slots"], 230, device=dev)
        mu, lv = torch.empty(i["B"], d, device=dev), torch.empty(i["B"], d, device=dev)
        ptr = self._ptr
        self._call("pm_vae_step_outputs", self.addr, ptr(s_logits), ptr(c_logits), ptr(mu), ptr(lv), self._stream())
        return (s_logits, c_logits), mu, lv

    def output_views(self):
        """`((s_logits, c_logits), mu, log_var)` of the last forward as VIEWS of the workspace arena (no copies: c_logits is
        224 MB at 15 slots) — valid until the next forward of this model overwrites the arena."""
        off, num = (C.c_int64 * 8)(), (C.c_int64 * 8)()
        self._call("pm_vae_step_output_views", self.addr, C.cast(off, C.c_void_p), C.cast(num, C.c_void_p))
        i, vae = self.info(), self.vae
        d, nb = vae.cfg["d"], vae.cfg["n_bars"]
        shapes = ((i["B"], nb, 4, 32), (i["N"], i["n_slots"], 230), (i["B"], d), (i["B"], d))
        s_logits, c_logits, mu, lv = (self.ws[int(off[k]):int(off[k]) + 4 * int(num[k])].view(torch.float32).view(shapes[k])
                                      for k in range(4))
        return (s_logits, c_logits), mu, lv

    def set_output_grads(self, ds_logits, dc_logits, dmu, dlv):
        ptr = self._ptr
        # (d_c_logits is READ where it lies by the backward calls: keep it alive until the next forward)
        self._grad_refs = (ds_logits, dc_logits, dmu, dlv)
        self._call("pm_vae_step_set_output_grads", self.addr, ptr(ds_logits), ptr(dc_logits), ptr(dmu), ptr(dlv), self._stream())

    def backward_decoder(self):
        self._call("pm_vae_step_backward_decoder", self.addr, self._stream())

    def backward_encoder_heads(self):
        self._call("pm_vae_step_backward_encoder_heads", self.addr, self._stream())

    def backward_encoder(self):
        self._call("pm_vae_step_backward_encoder", self.addr, self._stream())

    def backward_encoder_tail(self):
        self._call("pm_vae_step_backward_encoder_tail", self.addr, self._stream())

    def bump_counters(self):
        """num_batches_tracked (int64 bookkeeping of nn.BatchNorm): +1 per forward; the embedding norms only when their group
        is non-empty, bn_dur once per non-empty group (model.py:362,375)."""
        vae, ptr = self.vae, self._ptr
        i = self._fields.index("group_cnt")
        off = self.plan_off
        self._call("pm_bn_counters_update", ptr(vae.flat_counters), ptr(self._nbt_inc), ptr(self._nbt_sel),
                   ptr(self.plan_buf[off[i]:off[i] + 2]), vae.flat_counters.numel(), self._stream())

    def plan_word(self, field: str, index: int) -> int:
        """One int of the plan (a host read: debugging only)."""
        j = self._fields.index(field)
        return int(self.plan_buf[self.plan_off[j] + index].item())
