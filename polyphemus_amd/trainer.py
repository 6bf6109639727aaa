"""Sync-free training step of the HIP graph-VAE (the measured hot path).

Mirrors one iteration of `PolyphemusTrainer.train` (reference training.py:129-172):
forward, `_losses`, backward, `optimizer.step()`, `zero_grad()`, `lr_scheduler.step()` —
but without autograd bookkeeping and without the reference's 7 `.item()` host syncs:

    plan build -> encoder -> reparametrisation -> decoder -> fused CE/KLD/BCE loss (+ dlogits)
    -> decoder backward -> reparam backward -> encoder backward
    -> [data parallel: RCCL all-reduce of the flat gradient, two buckets overlapped with the
        encoder backward] -> fused Adam on the flat parameter buffer

Reference quirks reproduced by default: the structure BCE is evaluated on the
target (no gradient reaches the structure decoder, B-1), beta stays 0 (B-3), message dropout
p = 0.1 is always on in training (B-2).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from . import ops
from .model import VAE, prepare_graph
from .parallel import GradBuckets, broadcast_


class ExpDecayLR:
    """`ExpDecayLRScheduler` (training.py:43-75): peak_lr during warm-up, then exponential decay."""

    def __init__(self, peak_lr, warmup_steps, final_lr_scale, decay_steps):
        self.peak_lr, self.warmup_steps = peak_lr, warmup_steps
        self.decay_factor = -math.log(final_lr_scale) / decay_steps
        self.update_steps = 0

    def step(self) -> float:
        self.update_steps += 1
        if self.update_steps <= self.warmup_steps:
            return self.peak_lr
        return self.peak_lr * math.exp(-self.decay_factor * (self.update_steps - self.warmup_steps))


class HipTrainer:
    def __init__(self, vae: VAE, lr=5e-6, betas=(0.9, 0.98), eps=1e-9, lr_scheduler: Optional[dict] = None,
                 structure_loss_on_logits: bool = False, beta: float = 0.0, process_group=None):
        self.vae = vae
        self.lr, self.betas, self.eps = lr, betas, eps
        self.sched = ExpDecayLR(**lr_scheduler) if lr_scheduler else None
        self.fix_structure_loss = structure_loss_on_logits
        self.beta = beta
        self.pg = process_group
        flat = vae.flat_params
        self.grads = torch.zeros_like(flat)
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.step_count = 0
        self.loss_buf = torch.zeros(4, dtype=torch.float64, device=flat.device)
        P = dict(vae.named_parameters())
        self._G: Dict[str, torch.Tensor] = {}
        for n in vae._param_names:
            o = vae._offsets[n]
            self._G[n] = self.grads[o:o + P[n].numel()].view(P[n].shape)
        for k in list(self._G):
            if ".layers.0.nn." in k:
                head, tail = k.split(".layers.0.nn.")
                for i in range(1, vae.cfg["gnn_n_layers"]):
                    self._G[f"{head}.layers.{i}.nn.{tail}"] = self._G[k]
        dec_lo = vae._offsets[vae._names("decoder.")[0]]              # [0, dec_lo) = encoder grads
        self.buckets = GradBuckets(self.grads, [dec_lo], process_group)  # bucket 0 encoder, bucket 1 decoder
        self.world = self.buckets.world
        broadcast_([vae.flat_params, vae.flat_buffers], 0, process_group)

    # ------------------------------------------------------------------------------------------
    def train_step(self, graph, eps: Optional[torch.Tensor] = None):
        """One optimizer step on `graph` (device batch).  Returns the device tensor
        [pitch, dur, structure, kld] of loss values (float64, no host sync)."""
        vae, eng = self.vae, self.vae.engine
        if not vae.training:
            raise RuntimeError("train_step needs vae.train()")
        vae._check_flat()
        if self.grads.data_ptr() == 0 or self.grads.device != vae.flat_params.device:
            raise RuntimeError("trainer was built before the model was moved; rebuild it")
        eng.msg_dropout = vae.msg_dropout
        graph.__dict__.pop("_pm_plan", None)                 # the plan is part of the step (new batch every step)
        plan = prepare_graph(graph, vae.cfg["n_bars"])
        self.grads.zero_()
        G = self._G
        s_tensor = graph.s_tensor.float().contiguous()
        mu, lv, esv = eng.encoder_forward(plan, s_tensor, True, vae._next_seed())
        if eps is None:
            eps = torch.randn_like(mu)
        z = ops.reparam_fwd(mu, lv, eps)
        s_logits, c_logits, dsv = eng.decoder_forward(plan, z, True, vae._next_seed())
        # ---- loss (training.py:298-347) with gradients w.r.t. the logits
        out, dc = ops.content_ce(c_logits, plan, grad_scale=1.0, want_grad=True, out=self.loss_buf)
        dmu, dlv = torch.zeros_like(mu), torch.zeros_like(lv)
        ops.kld(mu, lv, out, beta=self.beta, dmu=dmu, dlog_var=dlv)
        if self.fix_structure_loss:
            _, ds = ops.bce_logits(s_logits.reshape(-1), s_tensor.reshape(-1), out, 1.0, want_grad=True)
            ds = ds.view_as(s_logits)
        else:                                                # training.py:307: BCE of the target against itself
            ops.bce_logits(s_tensor.reshape(-1), s_tensor.reshape(-1), out, 1.0, want_grad=False)
            ds = None
        # ---- backward
        dz = eng.decoder_backward(dsv, ds, dc, G)
        self.buckets.launch(1)                               # decoder gradients: overlapped with the encoder backward
        ops.reparam_bwd(dz, lv, eps, dmu, dlv)
        eng.encoder_backward(esv, dmu, dlv, G)
        self.buckets.launch(0)                               # encoder gradients
        mean_scale = self.buckets.wait()
        # ---- optimizer (training.py:160-172)
        self.step_count += 1
        ops.adam_step(vae.flat_params, self.grads, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0],
                      self.betas[1], self.eps, self.step_count, grad_scale=mean_scale)
        if self.sched is not None:
            self.lr = self.sched.step()
        return out

    def losses_dict(self, out: torch.Tensor) -> dict:
        """Host copy of the loss vector in the reference's dict layout (this DOES sync)."""
        p, d, s, k = out.tolist()
        rec = p + d + s
        return {"tot": rec + self.beta * k, "pitch": p, "dur": d, "structure": s, "reconstruction": rec, "kld": k,
                "beta*kld": self.beta * k}
