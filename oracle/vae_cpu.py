"""CPU restatement of the Polyphemus graph-VAE hot path — ORACLE / TEST
INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module; the product (`polyphemus_amd/`) never does and fails loudly
without its HIP extension.

It executes the *reference's op sequence* (per-relation Python loop, boolean-mask
edge selection, gather, `Linear` on one-hot distances, ReLU, message dropout,
scatter-mean, 6+1 matmuls per layer, PyG-style attention pooling, stock
`torch.optim.Adam`) as plain functions over a reference-keyed `state_dict`
(SURVEY App. C), in fp32 torch on the CPU.  Gradients come from torch autograd.

Parity status: pinned by `tests/golden/{lmd2_tiny,nb3_tiny}.npz`, which were
captured from the reference's own `model.py` / `training.py` imported unchanged
over `oracle/pyg_shim` (`oracle/make_golden.py`); `tests/test_oracle_golden.py`
checks every output, loss, gradient, buffer and Adam-updated parameter.  The
real torch_geometric 2.0.2 is not installable here, so the residual risk is the
shim's fidelity to PyG — stated in DESIGN.md.

Every function cites the reference lines it follows (`/root/reference/...`).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

N_PITCH, N_DUR, N_SLOTS, N_REL = 131, 99, 15, 6        # constants.py:28,40,48,58
PITCH_PAD, DUR_PAD = 130, 98                           # constants.py:25,38

Params = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------- utils
# Element dropout of the `cfg.dropout` layers.  The reference draws these masks from torch's global RNG (F.dropout /
# nn.Dropout); to compare a run with dropout against the HIP path the tests install ELEM_KEEP(site, rows, cols) -> 0/1
# mask [len(rows), cols] (the counter hash of csrc/common.h replayed in numpy), exactly like `keep_mask` does for the
# message dropout.  `rows` are the global row ids of the tensor's rows (node ids where the reference splits a tensor
# into drums / non-drums before the layer).  None = torch's own F.dropout.
ELEM_KEEP: Optional[Callable] = None


def _drop(x, p: float, training: bool, site: str, rows=None):
    if not training or p == 0:
        return x
    if ELEM_KEEP is None:
        return F.dropout(x, p, True)
    flat = x.reshape(x.shape[0], -1)
    rows = torch.arange(flat.shape[0]) if rows is None else rows
    keep = ELEM_KEEP(site, rows, flat.shape[1]).to(x.dtype)
    return (flat * keep / (1.0 - p)).reshape(x.shape)


def _bn(x, P: Params, key: str, training: bool, eps=1e-5, momentum=0.1):
    """nn.BatchNorm1d/2d forward incl. running-stat update (in place in `P`)."""
    if x.shape[0] == 0:
        return x                                       # empty group: torch passes it through (SURVEY B-5)
    out = F.batch_norm(x, P[key + ".running_mean"], P[key + ".running_var"],
                       P[key + ".weight"], P[key + ".bias"], training, momentum, eps)
    if training:
        P[key + ".num_batches_tracked"] += 1
    return out


def _lin(x, P: Params, key: str):
    return F.linear(x, P[key + ".weight"], P[key + ".bias"])


# --------------------------------------------------------------------------- GCN
def gcl_forward(x, edge_index, edge_type, edge_attr, P: Params, key: str, training: bool,
                msg_dropout: float, keep_mask: Optional[Callable] = None):
    """`GCL.forward` + `GCL.message` + PyG propagate (model.py:55-135).

    out = sum_r mean_{e: type r, dst=n} dropout(relu(x[src_e] * edge_nn(onehot dist_e))) @ W_r
          + x @ root + bias, accumulated in relation order 0..5, then root, then bias."""
    N, d_out = x.shape[0], P[key + ".weight"].shape[2]
    out = torch.zeros(N, d_out, dtype=x.dtype)                             # model.py:79
    eids = torch.arange(edge_index.shape[1])
    for r in range(N_REL):                                                 # model.py:103
        m = edge_type == r                                                 # float column vs int, model.py:104
        ei, attr = edge_index[:, m], edge_attr[m, :]                       # model.py:30-38
        x_j = x.index_select(0, ei[0])                                     # propagate: gather sources
        w = F.linear(attr, P[key + ".nn.weight"], P[key + ".nn.bias"])     # model.py:127
        w = w[..., :x.shape[1]].reshape(-1, x.shape[1])                    # model.py:128-129
        msg = F.relu(x_j * w)                                              # model.py:131-132
        if training and msg_dropout > 0:                                   # model.py:133
            if keep_mask is not None:
                msg = msg * keep_mask(key, eids[m], msg.shape[1]) / (1.0 - msg_dropout)
            else:
                msg = F.dropout(msg, p=msg_dropout, training=True)
        h = torch.zeros(N, msg.shape[1], dtype=msg.dtype).index_add_(0, ei[1], msg)   # scatter-sum onto dst
        cnt = torch.zeros(N, dtype=msg.dtype).index_add_(0, ei[1], torch.ones(ei.shape[1], dtype=msg.dtype))
        h = h / cnt.clamp(min=1).unsqueeze(1)                              # reduce='mean'
        out = out + h @ P[key + ".weight"][r]                              # model.py:112
    out = out + x @ P[key + ".root"]                                       # model.py:116
    out = out + P[key + ".bias"]                                           # model.py:119
    return out


def gcn_forward(x, graph, P: Params, key: str, cfg, training: bool, msg_dropout: float,
                keep_mask=None):
    """`GCN.forward` (model.py:190-208)."""
    edge_index, edge_attrs = graph.edge_index, graph.edge_attrs
    edge_type, edge_attr = edge_attrs[:, 0], edge_attrs[:, 1:]             # model.py:193-194
    for i in range(cfg["gnn_n_layers"]):
        residual = x
        x = _drop(x, cfg["dropout"], training, f"{key.split('.')[0][:3]}_gcn.{i}")     # model.py:199
        x = gcl_forward(x, edge_index, edge_type, edge_attr, P, f"{key}.layers.{i}", training,
                        msg_dropout, keep_mask)
        if cfg["batch_norm"]:
            x = _bn(x, P, f"{key}.norm_layers.{i}.module", training)       # model.py:203
        x = F.relu(x)
        x = residual + x                                                   # model.py:205-206
    return x


# --------------------------------------------------------------------------- encoder
def cnn_encoder(s, P: Params, key: str, cfg, training: bool):
    """`CNNEncoder.forward` (model.py:211-256)."""
    p = cfg["dropout"]
    x = s.unsqueeze(1)
    if cfg["batch_norm"]:
        x = F.conv2d(x, P[key + ".conv.0.weight"], P[key + ".conv.0.bias"], padding=1)
        x = F.relu(_bn(x, P, key + ".conv.1", training))
        x = F.max_pool2d(x, (1, 4), stride=(1, 4))
        x = F.conv2d(x, P[key + ".conv.4.weight"], P[key + ".conv.4.bias"], padding=1)
        x = F.relu(_bn(x, P, key + ".conv.5", training))
    else:
        x = F.relu(F.conv2d(x, P[key + ".conv.0.weight"], P[key + ".conv.0.bias"], padding=1))
        x = F.max_pool2d(x, (1, 4), stride=(1, 4))
        x = F.relu(F.conv2d(x, P[key + ".conv.3.weight"], P[key + ".conv.3.bias"], padding=1))
    x = x.flatten(1)
    x = _drop(x, p, training, "enc_cnn_in")
    x = F.relu(_lin(x, P, key + ".lin.1"))
    x = _drop(x, p, training, "enc_cnn_mid")
    return _lin(x, P, key + ".lin.4")


def structure_encoder(graph, P: Params, cfg, training: bool):
    """`StructureEncoder.forward` (model.py:434-445)."""
    d, nb = cfg["d"], cfg["n_bars"]
    out = cnn_encoder(graph.s_tensor.view(-1, 4, cfg["resolution"] * 4), P,
                      "encoder.s_encoder.cnn_encoder", cfg, training)
    return _lin(out.view(-1, nb * d), P, "encoder.s_encoder.bars_encoder")


def attention_pool(x, seg, P: Params, key: str, cfg, training: bool):
    """PyG `GlobalAttention(gate_nn)` (model.py:335-340,408-409):
    gate = BN1d(1)(Linear(d->1)(x)); softmax over the nodes of each bar with PyG's
    `exp(g - segmax) / (segsum + 1e-16)`; out[b] = sum_i gate_i * x_i."""
    size = int(seg[-1].item()) + 1
    g = _drop(x, cfg["dropout"], training, "enc_gate")                     # MLP.forward, model.py:160
    g = _lin(g, P, key + ".gate_nn.0.layers.0")
    g = _bn(g, P, key + ".gate_nn.1", training).view(-1, 1)
    gmax = torch.full((size, 1), float("-inf"), dtype=g.dtype).scatter_reduce(
        0, seg.view(-1, 1), g, reduce="amax", include_self=True)
    e = (g - gmax.index_select(0, seg)).exp()
    ssum = torch.zeros(size, 1, dtype=e.dtype).index_add_(0, seg, e)
    gate = e / (ssum.index_select(0, seg) + 1e-16)
    return torch.zeros(size, x.shape[1], dtype=x.dtype).index_add_(0, seg, gate * x)


def content_encoder(graph, P: Params, cfg, training: bool, msg_dropout: float, keep_mask=None):
    """`ContentEncoder.forward` (model.py:344-417)."""
    d, nb, p = cfg["d"], cfg["n_bars"], cfg["dropout"]
    k = "encoder.c_encoder"
    c = graph.c_tensor[:, 1:, :]                                           # drop SOS, model.py:349
    is_drum = graph.is_drum
    drums, non_drums = c[is_drum], c[torch.logical_not(is_drum)]           # model.py:352-353

    def embed(t, pitch_key, bn_key):                                       # model.py:356-377
        sz = t.size()
        pe = _lin(t[..., :N_PITCH], P, f"{k}.{pitch_key}")
        pe = _bn(pe.view(-1, d // 2), P, f"{k}.{bn_key}", training).view(sz[0], sz[1], d // 2)
        de = _lin(t[..., N_PITCH:], P, f"{k}.dur_emb")
        de = _bn(de.view(-1, d // 2), P, f"{k}.bn_dur", training).view(sz[0], sz[1], d // 2)
        return torch.cat((pe, de), dim=-1)

    drums = embed(drums, "drums_pitch_emb", "bn_drums")                    # drums first: bn_dur sees drums first
    non_drums = embed(non_drums, "non_drums_pitch_emb", "bn_non_drums")
    drums = F.relu(_lin(drums.view(-1, d * N_SLOTS), P, f"{k}.chord_encoder"))     # model.py:381-390
    non_drums = F.relu(_lin(non_drums.view(-1, d * N_SLOTS), P, f"{k}.chord_encoder"))
    node = torch.arange(c.size(0))
    drums = _drop(drums, p, training, "enc_chord", node[is_drum])
    non_drums = _drop(non_drums, p, training, "enc_chord", node[torch.logical_not(is_drum)])
    out = torch.zeros((c.size(0), d), dtype=drums.dtype)                   # model.py:394-397
    out[is_drum] = drums
    out[torch.logical_not(is_drum)] = non_drums
    distinct_bars = graph.bars + nb * graph.batch                          # model.py:403
    out = gcn_forward(out, graph, P, f"{k}.graph_encoder", cfg, training, msg_dropout, keep_mask)
    out = attention_pool(out, distinct_bars, P, f"{k}.graph_attention", cfg, training)
    return _lin(out.view(-1, nb * d), P, f"{k}.bars_encoder")              # model.py:412-414


def encoder_forward(graph, P: Params, cfg, training: bool, msg_dropout: float = 0.1, keep_mask=None):
    """`Encoder.forward` (model.py:466-483) -> (mu, log_var)."""
    p = cfg["dropout"]
    z_s = structure_encoder(graph, P, cfg, training)
    z_c = content_encoder(graph, P, cfg, training, msg_dropout, keep_mask)
    z_g = torch.cat((z_c, z_s), dim=1)
    z_g = _drop(z_g, p, training, "enc_merge_in")
    z_g = _lin(z_g, P, "encoder.linear_merge")
    z_g = F.relu(_bn(z_g, P, "encoder.bn_linear_merge", training))
    z_g = _drop(z_g, p, training, "enc_merge_out")
    return _lin(z_g, P, "encoder.linear_mu"), _lin(z_g, P, "encoder.linear_log_var")


# --------------------------------------------------------------------------- decoder
def cnn_decoder(x, P: Params, key: str, cfg, training: bool):
    """`CNNDecoder.forward` (model.py:259-299)."""
    p = cfg["dropout"]
    x = _drop(x, p, training, "dec_cnn_in")
    x = F.relu(_lin(x, P, key + ".lin.1"))
    x = _drop(x, p, training, "dec_cnn_mid")
    x = F.relu(_lin(x, P, key + ".lin.4"))
    x = x.view(-1, 16, 4, 8)
    x = F.interpolate(x, scale_factor=(1, 4), mode="nearest")
    x = F.conv2d(x, P[key + ".conv.1.weight"], P[key + ".conv.1.bias"], padding=1)
    if cfg["batch_norm"]:
        x = F.relu(_bn(x, P, key + ".conv.2", training))
        x = F.conv2d(x, P[key + ".conv.4.weight"], P[key + ".conv.4.bias"], padding=1)
    else:
        x = F.relu(x)
        x = F.conv2d(x, P[key + ".conv.3.weight"], P[key + ".conv.3.bias"], padding=1)
    return x.unsqueeze(1)


def structure_decoder(z_s, P: Params, cfg, training: bool):
    """`StructureDecoder.forward` (model.py:500-505)."""
    d, nb = cfg["d"], cfg["n_bars"]
    out = _lin(z_s, P, "decoder.s_decoder.bars_decoder")
    out = cnn_decoder(out.reshape(-1, d), P, "decoder.s_decoder.cnn_decoder", cfg, training)
    return out.view(z_s.size(0), nb, 4, -1)


def content_decoder(z_c, graph, P: Params, cfg, training: bool, msg_dropout: float, keep_mask=None):
    """`ContentDecoder.forward` (model.py:536-578)."""
    d, nb, p = cfg["d"], cfg["n_bars"], cfg["dropout"]
    k = "decoder.c_decoder"
    out = _lin(z_c, P, f"{k}.bars_decoder")
    distinct_bars = graph.bars + nb * graph.batch                          # model.py:542
    _, counts = torch.unique(distinct_bars, return_counts=True)
    out = torch.repeat_interleave(out.view(-1, d), counts, dim=0)          # model.py:543-545
    out = gcn_forward(out, graph, P, f"{k}.graph_decoder", cfg, training, msg_dropout, keep_mask)
    out = _lin(out, P, f"{k}.chord_decoder").view(-1, N_SLOTS, d)          # model.py:549-550
    is_drum = graph.is_drum
    drums, non_drums = out[is_drum], out[torch.logical_not(is_drum)]
    node = torch.arange(out.shape[0])
    non_drums = _drop(non_drums, p, training, "dec_chord", node[torch.logical_not(is_drum)])
    drums = _drop(drums, p, training, "dec_chord", node[is_drum])
    drums = torch.cat((_lin(drums[..., :d // 2], P, f"{k}.drums_pitch_emb"),
                       _lin(drums[..., d // 2:], P, f"{k}.dur_emb")), dim=-1)          # model.py:561-563
    non_drums = torch.cat((_lin(non_drums[..., :d // 2], P, f"{k}.non_drums_pitch_emb"),
                           _lin(non_drums[..., d // 2:], P, f"{k}.dur_emb")), dim=-1)  # model.py:566-568
    res = torch.zeros((int(graph.num_nodes), N_SLOTS, N_PITCH + N_DUR), dtype=drums.dtype)
    res[is_drum] = drums
    res[torch.logical_not(is_drum)] = non_drums
    return res


def decoder_forward(z, graph, P: Params, cfg, training: bool, msg_dropout: float = 0.1, keep_mask=None):
    """`Decoder.forward` with a given structure (model.py:634-655) -> (s_logits, c_logits)."""
    d = cfg["d"]
    z = _lin(z, P, "decoder.lin_decoder")
    z = F.relu(_bn(z, P, "decoder.batch_norm", training))
    z = _drop(z, cfg["dropout"], training, "dec_in")
    z_s, z_c = z[:, :d], z[:, d:]
    s_logits = structure_decoder(z_s, P, cfg, training)
    c_logits = content_decoder(z_c, graph, P, cfg, training, msg_dropout, keep_mask)
    return s_logits, c_logits


def vae_forward(graph, P: Params, cfg, training: bool, eps: Optional[torch.Tensor] = None,
                msg_dropout: float = 0.1, keep_mask=None):
    """`VAE.forward` (model.py:665-678); `eps` replaces `torch.randn_like` when given."""
    mu, log_var = encoder_forward(graph, P, cfg, training, msg_dropout, keep_mask)
    z = torch.exp(0.5 * log_var)
    z = z * (torch.randn_like(z) if eps is None else eps)
    z = z + mu
    s_logits, c_logits = decoder_forward(z, graph, P, cfg, training, msg_dropout, keep_mask)
    return (s_logits, c_logits), mu, log_var


def binary_from_logits(s_logits, thresh: float = 0.5):
    """`Decoder._binary_from_logits` (model.py:609-623): >= thresh -> 1, < thresh -> 0, `.bool()` (a NaN stays NaN and
    becomes True); empty bars get cell [0,0]."""
    s = ~(torch.sigmoid(s_logits) < thresh)
    empty = ~s.any(dim=-1).any(dim=-1)
    idx = torch.nonzero(empty, as_tuple=True)
    s[idx + (0, 0)] = True
    return s


def mtp_from_logits(c_logits, s_tensor):
    """`mtp_from_logits` (utils.py:59-79): [B,nb,4,32,15,230]; active cells (row-major, the node order) take the nodes'
    logits, the others the hard silence: row 0 one-hot pitch EOS (129), rows 1.. one-hot pitch PAD (130)."""
    n_slots, d_tok = c_logits.shape[-2], c_logits.shape[-1]
    on = s_tensor.bool().reshape(-1)
    mtp = torch.zeros(on.numel(), n_slots, d_tok, dtype=c_logits.dtype)
    silence = torch.zeros(n_slots, d_tok, dtype=c_logits.dtype)
    silence[0, 129] = 1.0                                                    # constants.py:24
    silence[1:, 130] = 1.0                                                   # constants.py:25
    mtp[on] = c_logits                                                       # raises when the counts differ (utils.py:74)
    mtp[~on] = silence
    return mtp.reshape(*s_tensor.shape, n_slots, d_tok)


# --------------------------------------------------------------------------- loss / optimiser
def losses(s_tensor, s_logits, c_tensor, c_logits, mu, log_var, beta: float = 0.0,
           structure_loss_on_logits: bool = False):
    """`PolyphemusTrainer._losses` (training.py:298-347), including its quirk that the
    structure BCE is evaluated on the *target* (training.py:307 overwrites the logits,
    SURVEY B-1) and that beta stays 0 (SURVEY B-3).  Returns (tot, dict of tensors)."""
    c_tensor = c_tensor[..., 1:, :]
    c_logits = c_logits.reshape(-1, c_logits.size(-1))
    c_tensor = c_tensor.reshape(-1, c_tensor.size(-1))
    if structure_loss_on_logits:
        s_in = s_logits.reshape(-1, *s_logits.shape[2:])
    else:
        s_in = s_tensor.reshape(-1, *s_logits.shape[2:])                   # training.py:307
    s_loss = F.binary_cross_entropy_with_logits(s_in.reshape(-1), s_tensor.reshape(-1).to(s_in.dtype),
                                                reduction="none").mean()
    pitch_true = c_tensor[:, :N_PITCH].argmax(dim=1)
    pitch_loss = F.cross_entropy(c_logits[:, :N_PITCH], pitch_true, ignore_index=PITCH_PAD)
    dur_true = c_tensor[:, N_PITCH:].argmax(dim=1)
    dur_loss = F.cross_entropy(c_logits[:, N_PITCH:], dur_true, ignore_index=DUR_PAD)
    kld = (-0.5 * torch.sum(1 + log_var - mu.pow(2) - log_var.exp(), dim=1)).mean()
    rec = pitch_loss + dur_loss + s_loss
    tot = rec + beta * kld
    return tot, {"tot": tot, "pitch": pitch_loss, "dur": dur_loss, "structure": s_loss,
                 "reconstruction": rec, "kld": kld, "beta*kld": beta * kld}


def accuracies(s_tensor, s_logits, c_tensor, c_logits, is_drum, structure_on_logits: bool = False):
    """`PolyphemusTrainer._accuracies` (training.py:349-497): note / pitch / pitch_drums / pitch_non_drums / dur
    accuracies over the non-PAD tokens of slots 1..15 and the structure accuracy / precision / recall / f1 —
    with the same quirk as `_losses` (training.py:356): the structure "logits" are the target itself."""
    c_true = c_tensor[..., 1:, :]
    p_rec, p_true = c_logits[..., :N_PITCH].softmax(-1).argmax(-1), c_true[..., :N_PITCH].argmax(-1)
    d_rec, d_true = c_logits[..., N_PITCH:].softmax(-1).argmax(-1), c_true[..., N_PITCH:].argmax(-1)
    np_, nd_ = p_true != PITCH_PAD, d_true != DUR_PAD
    cp, cd = (p_rec == p_true) & np_, (d_rec == d_true) & nd_
    drum = is_drum.bool()
    s_in = (s_logits if structure_on_logits else s_tensor).reshape(-1).float()
    pred = (torch.sigmoid(s_in) >= 0.5).float()                                # training.py:472-474
    tgt = s_tensor.reshape(-1).float()
    tp = tgt[pred == 1].sum()
    prec, rec = tp / pred.sum(), tp / tgt.sum()
    return {"note": ((cp & cd).sum() / np_.sum()).item(), "pitch": (cp.sum() / np_.sum()).item(),
            "pitch_drums": (cp[drum].sum() / np_[drum].sum()).item(),
            "pitch_non_drums": (cp[~drum].sum() / np_[~drum].sum()).item(),
            "dur": (cd.sum() / nd_.sum()).item(), "s_acc": ((pred == tgt).sum() / tgt.numel()).item(),
            "s_precision": prec.item(), "s_recall": rec.item(), "s_f1": (2 * rec * prec / (rec + prec)).item()}


def exp_decay_lr(update_steps: int, peak_lr, warmup_steps, final_lr_scale, decay_steps):
    """`ExpDecayLRScheduler.step` value after `update_steps` calls (training.py:43-75)."""
    if update_steps <= warmup_steps:
        return peak_lr
    k = -math.log(final_lr_scale) / decay_steps
    return peak_lr * math.exp(-k * (update_steps - warmup_steps))


PARAM_SUFFIXES = (".weight", ".bias", ".root")


def split_state(sd: Params, param_names):
    """Clone a state_dict into leaf parameters (requires_grad) and buffers; the shared
    edge network (`layers.{i}.nn.*`) is ONE leaf aliased under every key."""
    P: Params = {}
    names = list(param_names)
    for n in names:
        P[n] = sd[n].clone().requires_grad_(True)
    for k, v in sd.items():
        if k in P:
            continue
        if ".layers." in k and ".nn." in k:                                # alias of layers.0.nn.*
            head, tail = k.split(".layers.")
            P[k] = P[f"{head}.layers.0.nn.{tail.split('.nn.')[1]}"]
        else:
            P[k] = v.clone()
    return P, names


def train_step(graph, P: Params, names, cfg, opt: torch.optim.Optimizer, eps=None,
               msg_dropout: float = 0.1, keep_mask=None):
    """One reference training step on the CPU (training.py:137-166): forward, `_losses`,
    backward, `optimizer.step()`, `zero_grad()`."""
    (s_logits, c_logits), mu, log_var = vae_forward(graph, P, cfg, True, eps, msg_dropout, keep_mask)
    tot, parts = losses(graph.s_tensor, s_logits, graph.c_tensor, c_logits, mu, log_var)
    tot.backward()
    grads = {n: (None if P[n].grad is None else P[n].grad.clone()) for n in names}
    opt.step()
    opt.zero_grad()
    return (s_logits, c_logits, mu, log_var), parts, grads
