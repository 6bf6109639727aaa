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
               keep_logits: bool = False) -> PmBatch:
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
    b.flags = (1 if track_unique else 0) | (2 if _PLANES else 0) | (4 if keep_logits else 0)
    return b
