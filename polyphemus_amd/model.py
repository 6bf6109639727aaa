"""Drop-in `VAE` for the Polyphemus graph-VAE hot path, running on MI355X HIP kernels.

Mirrors the reference's `nn.Module` surface (reference model.py:658-678):

    vae = VAE(dropout=0, batch_norm=True, gnn_n_layers=8, d=512, n_bars=2, resolution=8, device=dev)
    (s_logits, c_logits), mu, log_var = vae(graph)      # graph: BarGraphBatch or a PyG-style Batch
    mu, log_var = vae.encoder(graph);   s_logits, c_logits = vae.decoder(z, graph_or_None)
    vae.state_dict() / load_state_dict()                # 255 reference keys

The module tree below only *holds* parameters (same names, shapes, registration order and
default initialisation as the reference, so `torch.manual_seed(0)` gives the same weights and
reference checkpoints load unchanged).  All arithmetic is done by `engine.Engine` through the C
ABI of libpolyphemus_hip.so; on a CPU tensor, or without the built extension, forward raises.
Parameters are views into ONE flat fp32 buffer (`VAE.flat_params`) — the unit of the fused Adam
step and of the data-parallel gradient all-reduce.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
from torch import nn

from . import constants as C
from . import ops
from ._lib import HipExtensionError
from .engine import Engine
from .graphs import device_batch_from_structure, BarGraphBatch, collate_samples, graph_from_structure


# --------------------------------------------------------------------------- parameter containers
class GCL(nn.Module):
    """Parameter holder of one relational graph-conv layer (reference model.py:41-53 on PyG RGCNConv:
    weight [R,in,out], root [in,out], bias [out] — glorot / glorot / zeros — plus the shared edge
    network `nn`, re-initialised once per constructed layer as `reset(self.nn)` does)."""

    def __init__(self, in_channels, out_channels, num_relations, edge_nn, dropout=0.1):
        super().__init__()
        self.in_channels, self.out_channels, self.num_relations = in_channels, out_channels, num_relations
        self.weight = nn.Parameter(torch.empty(num_relations, in_channels, out_channels))
        self.root = nn.Parameter(torch.empty(in_channels, out_channels))
        self.bias = nn.Parameter(torch.empty(out_channels))
        for t in (self.weight, self.root):
            bound = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
            t.data.uniform_(-bound, bound)
        self.bias.data.zero_()
        self.nn = edge_nn
        self.dropout = dropout                       # message dropout, always 0.1 in the reference (SURVEY B-2)
        edge_nn.reset_parameters()


class _PygBatchNorm(nn.Module):
    """`torch_geometric.nn.norm.BatchNorm`: BatchNorm1d under `.module`."""

    def __init__(self, c):
        super().__init__()
        self.module = nn.BatchNorm1d(c)


class GCN(nn.Module):
    def __init__(self, input_dim, hidden_dim, n_layers, num_relations, num_dists=32, batch_norm=False, dropout=0.1):
        super().__init__()
        self.layers, self.norm_layers = nn.ModuleList(), nn.ModuleList()
        edge_nn = nn.Linear(num_dists, input_dim)
        self.batch_norm, self.p = batch_norm, dropout
        for i in range(n_layers):
            self.layers.append(GCL(input_dim if i == 0 else hidden_dim, hidden_dim, num_relations, edge_nn))
            if batch_norm:
                self.norm_layers.append(_PygBatchNorm(hidden_dim))


class MLP(nn.Module):
    def __init__(self, input_dim=256, hidden_dim=256, output_dim=256, num_layers=2, activation=True, dropout=0.1):
        super().__init__()
        self.layers = nn.ModuleList()
        if num_layers == 1:
            self.layers.append(nn.Linear(input_dim, output_dim))
        else:
            self.layers.append(nn.Linear(input_dim, hidden_dim))
            for _ in range(num_layers - 2):
                self.layers.append(nn.Linear(hidden_dim, hidden_dim))
            self.layers.append(nn.Linear(hidden_dim, output_dim))
        self.activation, self.p = activation, dropout


class GlobalAttention(nn.Module):
    def __init__(self, gate_nn):
        super().__init__()
        self.gate_nn = gate_nn


class CNNEncoder(nn.Module):
    def __init__(self, output_dim, dense_dim, batch_norm, dropout):
        super().__init__()
        if batch_norm:
            self.conv = nn.Sequential(nn.Conv2d(1, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(True),
                                      nn.MaxPool2d((1, 4), stride=(1, 4)), nn.Conv2d(8, 16, 3, padding=1),
                                      nn.BatchNorm2d(16), nn.ReLU(True))
        else:
            self.conv = nn.Sequential(nn.Conv2d(1, 8, 3, padding=1), nn.ReLU(True),
                                      nn.MaxPool2d((1, 4), stride=(1, 4)), nn.Conv2d(8, 16, 3, padding=1),
                                      nn.ReLU(True))
        self.lin = nn.Sequential(nn.Dropout(dropout), nn.Linear(16 * 4 * 8, dense_dim), nn.ReLU(True),
                                 nn.Dropout(dropout), nn.Linear(dense_dim, output_dim))


class CNNDecoder(nn.Module):
    def __init__(self, input_dim, dense_dim, batch_norm, dropout):
        super().__init__()
        self.lin = nn.Sequential(nn.Dropout(dropout), nn.Linear(input_dim, dense_dim), nn.ReLU(True),
                                 nn.Dropout(dropout), nn.Linear(dense_dim, 16 * 4 * 8), nn.ReLU(True))
        if batch_norm:
            self.conv = nn.Sequential(nn.Upsample(scale_factor=(1, 4), mode="nearest"),
                                      nn.Conv2d(16, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(True),
                                      nn.Conv2d(8, 1, 3, padding=1))
        else:
            self.conv = nn.Sequential(nn.Upsample(scale_factor=(1, 4), mode="nearest"),
                                      nn.Conv2d(16, 8, 3, padding=1), nn.ReLU(True), nn.Conv2d(8, 1, 3, padding=1))


class StructureEncoder(nn.Module):
    def __init__(self, **kw):
        super().__init__()
        self.__dict__.update(kw)
        self.cnn_encoder = CNNEncoder(self.d, self.d, self.batch_norm, self.dropout)
        self.bars_encoder = nn.Linear(self.n_bars * self.d, self.d)


class ContentEncoder(nn.Module):
    def __init__(self, **kw):
        super().__init__()
        self.__dict__.update(kw)
        d = self.d
        self.dropout_layer = nn.Dropout(p=self.dropout)
        self.non_drums_pitch_emb = nn.Linear(C.N_PITCH_TOKENS, d // 2)
        self.drums_pitch_emb = nn.Linear(C.N_PITCH_TOKENS, d // 2)
        self.dur_emb = nn.Linear(C.N_DUR_TOKENS, d // 2)
        self.bn_non_drums = nn.BatchNorm1d(d // 2)
        self.bn_drums = nn.BatchNorm1d(d // 2)
        self.bn_dur = nn.BatchNorm1d(d // 2)
        self.chord_encoder = nn.Linear(d * C.N_SLOTS, d)
        self.graph_encoder = GCN(d, d, self.gnn_n_layers, C.N_EDGE_TYPES, batch_norm=self.batch_norm,
                                 dropout=self.dropout)
        self.graph_attention = GlobalAttention(nn.Sequential(
            MLP(input_dim=d, output_dim=1, num_layers=1, activation=False, dropout=self.dropout),
            nn.BatchNorm1d(1)))
        self.bars_encoder = nn.Linear(self.n_bars * d, d)


class StructureDecoder(nn.Module):
    def __init__(self, **kw):
        super().__init__()
        self.__dict__.update(kw)
        self.bars_decoder = nn.Linear(self.d, self.d * self.n_bars)
        self.cnn_decoder = CNNDecoder(self.d, self.d, self.batch_norm, self.dropout)


class ContentDecoder(nn.Module):
    def __init__(self, **kw):
        super().__init__()
        self.__dict__.update(kw)
        d = self.d
        self.bars_decoder = nn.Linear(d, d * self.n_bars)
        self.graph_decoder = GCN(d, d, self.gnn_n_layers, C.N_EDGE_TYPES, batch_norm=self.batch_norm,
                                 dropout=self.dropout)
        self.chord_decoder = nn.Linear(d, d * C.N_SLOTS)
        self.drums_pitch_emb = nn.Linear(d // 2, C.N_PITCH_TOKENS)
        self.non_drums_pitch_emb = nn.Linear(d // 2, C.N_PITCH_TOKENS)
        self.dur_emb = nn.Linear(d // 2, C.N_DUR_TOKENS)
        self.dropout_layer = nn.Dropout(p=self.dropout)


# --------------------------------------------------------------------------- graph preparation
def prepare_graph(graph, n_bars: int) -> ops.Plan:
    """Device-side plan of a batch, cached on the graph object.  Accepts the compact
    BarGraphBatch attributes or the reference's PyG format (edge_attrs [E,33] with the type in
    column 0 and a one-hot distance, c_tensor one-hot [N,16,230]; data.py:179-182,235-268)."""
    plan = graph.__dict__.get("_pm_plan") if hasattr(graph, "__dict__") else None
    if plan is not None:
        return plan
    ei = graph.edge_index
    if not ei.is_cuda:
        raise HipExtensionError("the HIP VAE needs the batch on the GPU (graph.to('cuda')); there is no CPU path")
    has = lambda k: hasattr(graph, "__dict__") and k in graph.__dict__ or (hasattr(graph, "keys") and k in graph.keys())
    if has("edge_type") and has("edge_dist"):
        et, ed = graph.edge_type.to(torch.int32), graph.edge_dist.to(torch.int32)
    else:
        et, ed = ops.edge_attrs_to_ids(graph.edge_attrs.float().contiguous())
    tokens = graph.tokens.to(torch.int32).contiguous() if has("tokens") else ops.tokens_from_onehot(
        graph.c_tensor.float().contiguous())
    G = int(graph.s_tensor.shape[0])
    plan = ops.plan_build(ei.contiguous(), et.contiguous(), ed.contiguous(), graph.bars.contiguous(),
                          graph.batch.contiguous(), graph.is_drum.contiguous(), tokens, n_bars, G)
    try:
        graph.__dict__["_pm_plan"] = plan
    except Exception:
        pass
    return plan


# --------------------------------------------------------------------------- autograd bridges
class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vae, plan, s_tensor, *params):
        eng = vae.engine
        mu, lv, saved = eng.encoder_forward(plan, s_tensor, vae.training, vae._next_seed())
        ctx.vae, ctx.saved = vae, saved
        ctx.set_materialize_grads(False)
        return mu, lv

    @staticmethod
    def backward(ctx, dmu, dlv):
        vae = ctx.vae
        G = vae._grad_views("encoder.")
        z = lambda t, ref: torch.zeros_like(ref) if t is None else t.contiguous().float()
        vae.engine.encoder_backward(ctx.saved, z(dmu, ctx.saved["mu"]), z(dlv, ctx.saved["mu"]), G)
        return (None, None, None) + tuple(G[n] for n in vae._names("encoder."))


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vae, plan, z, *params):
        s_logits, c_logits, saved = vae.engine.decoder_forward(plan, z.contiguous().float(), vae.training,
                                                               vae._next_seed())
        ctx.vae, ctx.saved = vae, saved
        ctx.set_materialize_grads(False)
        return s_logits, c_logits

    @staticmethod
    def backward(ctx, ds, dc):
        vae = ctx.vae
        G = vae._grad_views("decoder.")
        ds = None if ds is None else ds.contiguous().float()
        dc = None if dc is None else dc.contiguous().float()
        dz = vae.engine.decoder_backward(ctx.saved, ds, dc, G)
        return (None, None, dz) + tuple(G[n] for n in vae._names("decoder."))


class _ReparamFn(torch.autograd.Function):
    """z = exp(0.5*log_var) * eps + mu (reference model.py:671-673)."""

    @staticmethod
    def forward(ctx, mu, log_var, eps):
        ctx.save_for_backward(log_var, eps)
        return ops.reparam_fwd(mu.contiguous(), log_var.contiguous(), eps.contiguous())

    @staticmethod
    def backward(ctx, dz):
        log_var, eps = ctx.saved_tensors
        dmu, dlv = torch.zeros_like(log_var), torch.zeros_like(log_var)
        ops.reparam_bwd(dz.contiguous(), log_var, eps, dmu, dlv)
        return dmu, dlv, None


def _draw_eps(n_samples: int, d: int, device) -> torch.Tensor:
    """The reparametrisation noise of `VAE.forward` (model.py:672: `torch.randn_like(mu)`); a function of its own so that
    the parity tests can inject the reference's draw."""
    return torch.randn(n_samples, d, device=device)


class _VaeStepFn(torch.autograd.Function):
    """`VAE.forward` in training mode as ONE autograd node over the C++ step (csrc/vae_step.hip, the sequence bench.py
    measures): the forward is `pm_vae_step_forward` with the logits kept; the caller's loss (`_losses` of the unchanged
    training.py:298-347, under autocast / GradScaler) produces the gradients of the four outputs, which
    `pm_vae_step_set_output_grads` hands to the four native backward calls.  Parameter gradients come back as views of one
    flat buffer laid out like `flat_params`."""

    @staticmethod
    def forward(ctx, vae, graph, eps, *params):
        step = vae._native_step()
        n_slots = None if vae.active_slots_only else 15
        # (a fresh zeroed gradient buffer per call, 10 us of memset at d = 256: autograd may keep the returned views as
        #  `p.grad` — with gradient accumulation beyond this step — so the buffer cannot be the previous call's)
        grads = torch.zeros_like(vae.flat_params)
        # ext_loss: the caller computes the loss — the forward stores the logits and runs none of its own loss kernels
        step.forward(graph, eps, grads, keep_logits=True, n_slots=n_slots, ext_loss=True)
        # outputs as views of the step's arena (no 224 MB copy of the logits); vae.outputs_as_views = False: fresh copies, for a
        # caller that keeps the outputs of one call alive across the next
        (s_logits, c_logits), mu, lv = step.output_views() if vae.outputs_as_views else step.outputs()
        S = c_logits.shape[1]
        if S < 15:                                           # slots that hold PAD in every node: not computed (opt-in)
            full = c_logits.new_zeros(c_logits.shape[0], 15, c_logits.shape[2])
            full[:, :S] = c_logits
            c_logits = full
        step.bump_counters()
        ctx.vae, ctx.step, ctx.grads, ctx.S = vae, step, grads, S
        ctx.ticket = vae._step_ticket = vae._step_ticket + 1
        ctx.set_materialize_grads(False)
        return s_logits, c_logits, mu, lv

    @staticmethod
    def backward(ctx, ds, dc, dmu, dlv):
        vae, step = ctx.vae, ctx.step
        if ctx.ticket != vae._step_ticket:
            raise RuntimeError("backward through a model(graph) call that is not the model's latest: the native step keeps "
                               "the activations of ONE forward (set vae.native_step = False to keep several graphs alive)")
        f = lambda t: None if t is None else t.contiguous().float()
        dc = f(dc)
        if dc is not None and ctx.S < 15:
            dc = dc[:, :ctx.S].contiguous()
        step.set_output_grads(f(ds), dc, f(dmu), f(dlv))
        step.backward_decoder()
        step.backward_encoder_heads()
        step.backward_encoder()
        step.backward_encoder_tail()
        vae._step_ticket += 1                                # (a second backward through the same forward has nothing to read)
        g, P = ctx.grads, dict(vae.named_parameters())
        return (None, None, None) + tuple(g[vae._offsets[n]:vae._offsets[n] + P[n].numel()].view(P[n].shape)
                                          for n in vae._param_names)


# --------------------------------------------------------------------------- encoder / decoder / VAE
class Encoder(nn.Module):
    def __init__(self, **kw):
        super().__init__()
        self.__dict__.update(kw)
        self.s_encoder = StructureEncoder(**kw)
        self.c_encoder = ContentEncoder(**kw)
        self.dropout_layer = nn.Dropout(p=self.dropout)
        self.linear_merge = nn.Linear(2 * self.d, self.d)
        self.bn_linear_merge = nn.BatchNorm1d(self.d)
        self.linear_mu = nn.Linear(self.d, self.d)
        self.linear_log_var = nn.Linear(self.d, self.d)

    def forward(self, graph):
        vae = self.__dict__["_vae"]()
        plan = vae._prepare(graph)
        s = graph.s_tensor.float().contiguous()
        with torch.autocast("cuda", enabled=False):          # the kernels are fp32 (training.py:137 turns autocast on)
            mu, lv = _EncoderFn.apply(vae, plan, s, *vae._tensors("encoder."))
        graph.distinct_bars = graph.bars + self.n_bars * graph.batch            # reference side effect, model.py:403
        return mu, lv


class Decoder(nn.Module):
    def __init__(self, **kw):
        super().__init__()
        self.__dict__.update(kw)
        self.lin_decoder = nn.Linear(self.d, 2 * self.d)
        self.batch_norm = nn.BatchNorm1d(2 * self.d)
        self.dropout = nn.Dropout(p=self.dropout)
        self.s_decoder = StructureDecoder(**kw)
        self.c_decoder = ContentDecoder(**kw)
        self.sigmoid_thresh = 0.5

    # helpers of the generation path (reference model.py:596-632).  A cuda structure is turned into its batch of bar
    # graphs ON THE DEVICE (csrc/graph.hip: the reference's node numbering and edge order, one host read of (N, E));
    # a CPU structure takes the numpy restatement of `graph_from_tensor`.
    def _structure_from_binary(self, s_tensor):
        if s_tensor.is_cuda:
            b = device_batch_from_structure(s_tensor.reshape(-1, 4, 32), self.n_bars)
            s_tensor.copy_(b.s_tensor.view_as(s_tensor).to(s_tensor.dtype))      # empty bars get [0,0] (data.py:152-153)
            b.tokens = torch.zeros(b.num_nodes, 16, 2, dtype=torch.int32, device=s_tensor.device)
            return b
        s_np = s_tensor.detach().cpu().numpy().astype(bool)
        samples = []
        for i in range(s_np.shape[0]):
            g = graph_from_structure(s_np[i])
            g["tokens"] = np.zeros((g["num_nodes"], 16, 2), np.int32)
            g["s_tensor"] = s_np[i].astype(np.float32)
            samples.append(g)
        s_tensor.copy_(torch.from_numpy(s_np).to(s_tensor.device))              # empty bars get [0,0] (data.py:152-153)
        return collate_samples(samples, self.n_bars).to(next(self.parameters()).device)

    def _binary_from_logits(self, s_logits):
        # model.py:609-623.  A cuda tensor takes one kernel (csrc/generate.hip, no `nonzero` host sync); a host tensor
        # (generate.py prepares its structures on the CPU) the reference's own few torch lines.
        if s_logits.is_cuda:
            return ops.binary_from_logits(s_logits.detach().contiguous().float(), self.sigmoid_thresh)
        s = ~(torch.sigmoid(s_logits) < self.sigmoid_thresh)
        empty = ~s.any(dim=-1).any(dim=-1)
        idx = torch.nonzero(empty, as_tuple=True)
        s[idx + (0, 0)] = True
        return s

    def _structure_from_logits(self, s_logits):
        return self._structure_from_binary(self._binary_from_logits(s_logits))

    def forward(self, z, s=None):
        vae = self.__dict__["_vae"]()
        if s is None:
            # generation: the structure comes from the thresholded structure logits, built on the host
            # exactly as the reference does (model.py:646-650); needs a first structure-only pass.
            with torch.no_grad():
                s_logits0 = vae.engine.structure_only(z.contiguous().float(), vae.training)
            s_bin = self._binary_from_logits(s_logits0.detach())
            # the content decoder runs on THIS thresholded structure; callers that lay the result out (generate_music)
            # must use the same one: the logits this forward returns come from a second pass of the head GEMMs, whose
            # split-K float atomics may differ in the last bits and flip a cell whose logit is ~0
            self.__dict__["_last_structure"] = s_bin.clone()
            s = self._structure_from_binary(s_bin)
        plan = vae._prepare(s)
        with torch.autocast("cuda", enabled=False):
            s_logits, c_logits = _DecoderFn.apply(vae, plan, z, *vae._tensors("decoder."))
        s.distinct_bars = s.bars + self.n_bars * s.batch                          # model.py:542
        return s_logits, c_logits


class VAE(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        self.cfg = {k: kwargs[k] for k in ("dropout", "batch_norm", "gnn_n_layers", "d", "n_bars", "resolution")}
        if kwargs["d"] % 8 or kwargs["resolution"] != 8:
            raise ValueError("the HIP path needs d % 8 == 0 and resolution == 8 (4x32 bar grids)")
        self.encoder = Encoder(**kwargs)
        self.decoder = Decoder(**kwargs)
        import weakref
        self.encoder.__dict__["_vae"] = weakref.ref(self)
        self.decoder.__dict__["_vae"] = weakref.ref(self)
        self.msg_dropout = 0.1                # GCL message dropout, hard-wired in the reference (SURVEY B-2)
        self.seed = 0x5EED                    # BASE seed of the counter-based dropout streams (checkpointed as is)
        self.rank_salt = 0                    # data parallel: mixed into every derived seed, so ranks draw different masks;
        self._step = 0                        # set by the trainer from the rank, never stored in a checkpoint
        # `model(graph)` in training mode runs the C++ step (one autograd node, _VaeStepFn); False: the Python orchestration
        # of the same kernels (three autograd nodes over engine.Engine; several forwards may then be alive at once)
        self.native_step = True
        # True: the decoder head covers only the batch's active token slots — c_logits of slots that hold PAD in every node
        # come back as zeros (the reference's loss ignores them: same loss, same gradients); default: all 15, as the reference
        self.active_slots_only = False
        self.outputs_as_views = True                         # model(graph) in training mode returns views of the native step's arena
        self._step_ticket = 0
        self._flatten()

    # ---- flat parameter / buffer storage ---------------------------------------------------
    def _flatten(self):
        """(Re)pack every parameter and float buffer into flat fp32 buffers and re-point `.data`."""
        with torch.no_grad():
            named = list(self.named_parameters())
            dev = named[0][1].device
            offs, off = {}, 0
            for n, p in named:
                offs[n] = off
                off += (p.numel() + 3) // 4 * 4
            flat = torch.zeros(off, dtype=torch.float32, device=dev)
            for n, p in named:
                flat[offs[n]:offs[n] + p.numel()].copy_(p.data.reshape(-1))
                p.data = flat[offs[n]:offs[n] + p.numel()].view(p.shape)
            bufs = [(n, b) for n, b in self.named_buffers() if b.dtype.is_floating_point]
            boffs, boff = {}, 0
            for n, b in bufs:
                boffs[n] = boff
                boff += (b.numel() + 3) // 4 * 4
            bflat = torch.zeros(max(boff, 4), dtype=torch.float32, device=dev)
            for n, b in bufs:
                bflat[boffs[n]:boffs[n] + b.numel()].copy_(b.data.reshape(-1))
                b.data = bflat[boffs[n]:boffs[n] + b.numel()].view(b.shape)
            # integer buffers (BatchNorm.num_batches_tracked): one int64 vector, so a step bumps them with one launch
            ibufs = [(n, b) for n, b in self.named_buffers() if not b.dtype.is_floating_point and b.numel() == 1]
            cflat = torch.zeros(max(len(ibufs), 1), dtype=torch.int64, device=dev)
            for i, (n, b) in enumerate(ibufs):
                cflat[i] = b.data.reshape(()).to(torch.int64)
                b.data = cflat[i]
        self.__dict__["flat_counters"] = cflat
        self.__dict__["_counter_index"] = {n: i for i, (n, _) in enumerate(ibufs)}
        self.__dict__["flat_params"], self.__dict__["flat_buffers"] = flat, bflat
        self.__dict__["_offsets"], self.__dict__["_param_names"] = offs, [n for n, _ in named]
        self.__dict__["_buf_offsets"] = boffs
        tensors = {n: p for n, p in named}
        tensors.update({n: b for n, b in self.named_buffers()})
        for k in list(tensors):                        # shared edge network: alias layers.{i}.nn.* -> layers.0.nn.*
            if ".layers.0.nn." in k:
                head, tail = k.split(".layers.0.nn.")
                for i in range(1, self.cfg["gnn_n_layers"]):
                    tensors[f"{head}.layers.{i}.nn.{tail}"] = tensors[k]
        self.__dict__["engine"] = Engine(self.cfg, tensors)
        self.__dict__["engine"].msg_dropout = self.msg_dropout
        self.__dict__["_native"] = None                      # (the flat buffers moved: the C++ step's layout is rebuilt on use)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._flatten()
        return out

    def _check_flat(self):
        flat = self.flat_params
        for n, p in self.named_parameters():
            if p.data_ptr() != flat.data_ptr() + 4 * self._offsets[n]:
                self._flatten()
                return

    def _names(self, prefix: str) -> List[str]:
        return [n for n in self._param_names if n.startswith(prefix)]

    def _tensors(self, prefix: str):
        P = dict(self.named_parameters())
        return [P[n] for n in self._names(prefix)]

    def _grad_views(self, prefix: str) -> Dict[str, torch.Tensor]:
        """Zeroed gradient tensors for the parameters under `prefix`, as views of one flat buffer laid
        out like `flat_params` (so e.g. GCL weight|root stay adjacent for the fused 7d x d GEMM)."""
        names = self._names(prefix)
        lo = self._offsets[names[0]]
        last = names[-1]
        P = dict(self.named_parameters())
        hi = self._offsets[last] + (P[last].numel() + 3) // 4 * 4
        flat = torch.zeros(hi - lo, dtype=torch.float32, device=self.flat_params.device)
        G = {n: flat[self._offsets[n] - lo:self._offsets[n] - lo + P[n].numel()].view(P[n].shape) for n in names}
        for k in list(G):
            if ".layers.0.nn." in k:
                head, tail = k.split(".layers.0.nn.")
                for i in range(1, self.cfg["gnn_n_layers"]):
                    G[f"{head}.layers.{i}.nn.{tail}"] = G[k]
        return G

    def _next_seed(self) -> int:
        self._step += 1
        return (((self.seed ^ self.rank_salt) & 0xFFFFFFFF) * 0x9E3779B1 + self._step * 0x85EBCA77) & 0xFFFFFFFF

    def _prepare(self, graph) -> ops.Plan:
        self._check_flat()
        self.engine.msg_dropout = self.msg_dropout
        return prepare_graph(graph, self.cfg["n_bars"])

    def __getstate__(self):
        # the native step's handle holds ctypes structs with raw device pointers: not picklable, not copyable — rebuilt on use
        # (copy.deepcopy(vae) / torch.save(vae) after a native forward raised before round 6)
        st = dict(self.__dict__)
        st["_native"] = None
        return st

    def _native_step(self):
        from .native import NativeStep
        st = self.__dict__.get("_native")
        if st is None or st.flat_ptr != self.flat_params.data_ptr():
            st = NativeStep(self)
            self.__dict__["_native"] = st
        return st

    # ---- reference surface ------------------------------------------------------------------
    def forward(self, graph):
        if (self.native_step and self.training and torch.is_grad_enabled() and self.flat_params.is_cuda
                and not self.engine.syncing(True)):
            # the training step of the unchanged training.py:137-166: forward here, `_losses` and `backward()` in the caller
            self._check_flat()
            n_samples = int(graph.s_tensor.numel() // (128 * self.cfg["n_bars"]))
            eps = _draw_eps(n_samples, self.cfg["d"], self.flat_params.device)
            with torch.autocast("cuda", enabled=False):      # the kernels are fp32 (training.py:137 turns autocast on)
                s_logits, c_logits, mu, log_var = _VaeStepFn.apply(self, graph, eps, *self._tensors(""))
            graph.distinct_bars = graph.bars + self.cfg["n_bars"] * graph.batch        # reference side effect, model.py:403,542
            return (s_logits, c_logits), mu, log_var
        mu, log_var = self.encoder(graph)
        eps = torch.randn_like(mu)
        z = _ReparamFn.apply(mu, log_var, eps)
        out = self.decoder(z, graph)
        return out, mu, log_var
