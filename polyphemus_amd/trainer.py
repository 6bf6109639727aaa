"""Sync-free training step of the HIP graph-VAE (the measured hot path).

Mirrors one iteration of `PolyphemusTrainer.train` (reference training.py:129-172):
forward, `_losses`, backward, `optimizer.step()`, `zero_grad()`, `lr_scheduler.step()` —
but without autograd bookkeeping and without the reference's 7 `.item()` host syncs:

    plan build -> encoder -> reparametrisation -> decoder -> fused CE/KLD/BCE loss (+ dlogits)
    -> decoder backward -> reparam backward -> encoder backward
    -> [data parallel: RCCL all-reduce of the flat gradient in three buckets, two of them overlapped
        with the encoder backward] -> fused Adam on the flat parameter buffer

Two orchestrations of the SAME kernels:
  * native (default): four C calls (`pm_vae_step_forward`, `..._backward_decoder`,
    `..._backward_encoder`, `..._backward_encoder_tail`, csrc/vae_step.hip) issue the ~340 launches of a step from C++;
  * python (`native=False`): the same sequence through `engine.Engine` (the executable
    specification the autograd drop-in path uses); kept for cross-checking.

Reference quirks reproduced by default: the structure BCE is evaluated on the
target (no gradient reaches the structure decoder, B-1), beta stays 0 (B-3), message dropout
p = 0.1 is always on in training (B-2).
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Dict, Optional

import torch

from . import ops
from ._lib import call, lib, ptr, stream
from .model import VAE, prepare_graph
from .native import NativeStep, prepare_inputs
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


def bucket_boundaries(vae: VAE):
    """Element offsets that split the flat gradient buffer into the three exchange buckets, in flat order: 0 = structure encoder
    + embeddings + chord encoder (final last: behind `pm_vae_step_backward_encoder_tail`), 1 = graph encoder .. end of the
    encoder (final behind `pm_vae_step_backward_encoder`), 2 = decoder (final behind the decoder backward and the join of its
    weight gradients).  Asserts the parameter order the native split relies on."""
    dec_lo = vae._offsets[vae._names("decoder.")[0]]
    mid_lo = vae._offsets[vae._names("encoder.c_encoder.graph_encoder.")[0]]
    for n in vae._names("encoder."):
        late = n.startswith(("encoder.s_encoder.", "encoder.c_encoder.non_drums", "encoder.c_encoder.drums",
                             "encoder.c_encoder.dur_emb", "encoder.c_encoder.bn_", "encoder.c_encoder.chord_encoder"))
        assert (vae._offsets[n] < mid_lo) == late, f"unexpected parameter order at {n}"
    return [mid_lo, dec_lo]


class HipTrainer:
    def __init__(self, vae: VAE, lr=5e-6, betas=(0.9, 0.98), eps=1e-9, lr_scheduler: Optional[dict] = None,
                 structure_loss_on_logits: bool = False, beta: float = 0.0, process_group=None, native: bool = True,
                 iters_to_accumulate: int = 1, global_token_mean: bool = False, sync_bn: bool = False):
        self.vae = vae
        self.lr, self.betas, self.eps = lr, betas, eps
        self.sched = ExpDecayLR(**lr_scheduler) if lr_scheduler else None
        self.fix_structure_loss = structure_loss_on_logits
        self.beta = beta
        self.pg = process_group
        # Data parallel: the reference's CE losses are means over the non-PAD tokens of the batch (training.py:316-323),
        # so the mean of per-rank gradients weights every rank equally whatever its token count.  With
        # `global_token_mean` each rank's CE gradient is weighted n_local * world / n_global (one all-reduce of two
        # counts per step, no host sync): the averaged gradient is then that of the token mean over the GLOBAL batch.
        self.global_token_mean = bool(global_token_mean)
        # True: the native step also stores the content logits (`step_outputs`); off by default — the fused un-embedding +
        # cross-entropy then writes d(loss)/d(logits) only
        self.keep_logits = False
        # the C++ step covers every constructor switch of the model (batch_norm = False: model.py:176-188,218-238,278-292;
        # cfg.dropout: the element dropout layers of model.py:160,199,244-247,267-270,389-390,473,479,558-559,640);
        # `native=False` selects the Python orchestration of the same kernels (engine.py), kept for cross-checking
        self.sync_bn = bool(sync_bn)
        import torch.distributed as _dist
        _world = _dist.get_world_size(process_group) if (_dist.is_available() and _dist.is_initialized()) else 1
        # Synchronised BatchNorm: every training-mode norm takes its statistics over the GLOBAL batch, so that
        # — together with global_token_mean — a data-parallel step equals the single-device step on the concatenated
        # batch.  Runs through the Python orchestration (one small all-reduce per norm and direction, one host read per
        # step); an option for parity checks, not for throughput runs.  With ONE rank the statistics are global anyway:
        # the native step stays on.
        self.native = bool(native and not (self.sync_bn and _world > 1))
        if iters_to_accumulate < 1:
            raise ValueError("iters_to_accumulate must be >= 1")
        self.iters_to_accumulate = int(iters_to_accumulate)            # training.py:83,149,158
        self.micro_batches = 0                                         # `tot_batches` of the reference
        flat = vae.flat_params
        if not flat.is_cuda:
            raise RuntimeError("HipTrainer needs the model on the GPU (vae.to('cuda')) before it is built")
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
        self.buckets = GradBuckets(self.grads, bucket_boundaries(vae), process_group)
        self.world = self.buckets.world
        # gradient accumulation: running sum of grads / k; all-reduced (one bucket) and consumed by Adam every k-th batch
        self.grad_accum = torch.zeros_like(flat) if self.iters_to_accumulate > 1 else None
        self._accum_bucket = GradBuckets(self.grad_accum, [], process_group) if self.grad_accum is not None else None
        broadcast_([vae.flat_params, vae.flat_buffers], 0, process_group)
        vae.engine.set_sync_bn(process_group, self.sync_bn and self.world > 1)
        # every rank its own message-dropout stream (same seed = same masks on every replica): the model keeps its BASE
        # seed — what checkpoints store — and the rank only salts the seeds derived from it (VAE._next_seed), so building a
        # second trainer on the same model, resuming from a rank-0 checkpoint or resuming with another world size all give
        # each rank a stream of its own
        vae.rank_salt = 0
        if self.world > 1:
            import torch.distributed as dist
            vae.rank_salt = (0x9E3779B9 * (dist.get_rank(process_group) + 1)) & 0xFFFFFFFF
        # native step plumbing (polyphemus_amd/native.py: layout, host state blob, arena, plan buffer)
        self._flat_ptr = flat.data_ptr()
        self.step = NativeStep(vae) if self.native else None
        self.loss_buf = self.step.loss_buf if self.step is not None else self.loss_buf

    # ------------------------------------------------------------------------------------------
    @property
    def _plan_buf(self):
        return self.step.plan_buf

    def _native_forward_backward(self, graph, eps):
        vae, step = self.vae, self.step
        ce_scale = None
        gtm = self.global_token_mean and self.world > 1
        if gtm:
            import torch.distributed as dist
            inputs = prepare_inputs(graph, want_token_counts=True)
            tot = inputs[9].clone()
            dist.all_reduce(tot, group=self.pg)
            ce_scale = (inputs[9] * float(self.world) / tot).contiguous()      # stays on the device
        step.forward(graph, eps, self.grads, keep_logits=self.keep_logits, beta=self.beta,
                     fix_structure=self.fix_structure_loss, ce_scale=ce_scale, want_token_counts=gtm)
        if prepare_inputs(graph)[8] and os.environ.get("PM_DEBUG", "0") not in ("", "0"):
            # the plan kernels count the nodes that break the one-track-relation-per-node rule the compact GCL
            # relies on (plan.hip k_node_class, cnt[4]); a host read, hence only under PM_DEBUG
            bad = step.plan_word("trk_cnt", 4)
            if bad:
                raise RuntimeError(f"{bad} nodes receive track edges of more than one track but the batch was "
                                   "flagged track_unique (graphs.batch_flags); the compact GCL would be wrong")
        step.backward_decoder()
        # the decoder's last weight gradients run on the library's second stream beside the head chains; the encoder's head
        # chain ends with the caller's stream waiting for them, so the decoder bucket goes out behind it — in front of the
        # encoder's GCN stack, which overlaps the exchange
        step.backward_encoder_heads()
        self.buckets.launch(2)                               # decoder gradients: overlapped with the encoder backward
        step.backward_encoder()
        self.buckets.launch(1)                               # graph encoder .. encoder head: overlapped with the tail
        step.backward_encoder_tail()
        self.buckets.launch(0)                               # chord encoder, embeddings, structure encoder
        step.bump_counters()
        return step.loss_buf

    def step_info(self) -> dict:
        """Which variant of the native step the last `train_step` ran (`pm_vae_step_info`): compact GCL (K = 4d),
        bf16-planes GEMM operands, active token slots S, fragment-major weight planes (B-direct GEMM), batch sizes, the
        library's effective switches, the fp16 pair format of the GCL products (bit 0 encoder, bit 1 decoder stack)."""
        return self.step.info()

    def step_outputs(self):
        """`((s_logits, c_logits), mu, log_var)` of the last native `train_step` — what `VAE.forward` returns
        (model.py:676-678) — copied out of the workspace arena.  c_logits holds the active slots only: [N, S, 230]
        (the remaining slots are PAD in every node of the batch; the fused step never computes them)."""
        i = self.step_info()              # (the library's effective switches, not the environment at call time)
        if not self.keep_logits and i["fused_ce"]:
            raise RuntimeError("step_outputs needs trainer.keep_logits = True before the step (the fused un-embedding + "
                               "cross-entropy does not store the logits otherwise)")
        return self.step.outputs()

    def _python_forward_backward(self, graph, eps):
        vae, eng = self.vae, self.vae.engine
        eng.msg_dropout = vae.msg_dropout
        graph.__dict__.pop("_pm_plan", None)                 # the plan is part of the step (new batch every step)
        plan = prepare_graph(graph, vae.cfg["n_bars"])
        G = self._G
        s_tensor = graph.s_tensor.float().contiguous()
        mu, lv, esv = eng.encoder_forward(plan, s_tensor, True, vae._next_seed())
        if eps is None:
            eps = torch.randn_like(mu)
        z = ops.reparam_fwd(mu, lv, eps)
        s_logits, c_logits, dsv = eng.decoder_forward(plan, z, True, vae._next_seed())
        ce_scale = 1.0
        if self.global_token_mean and self.world > 1:          # weight n_local * world / n_global (a host read: this
            import torch.distributed as dist                   # orchestration is the parity path, not the measured one)
            tok = plan.tokens
            n_loc = (tok[:, 1:, 0] != 130).sum().double().reshape(1)
            n_all = n_loc.clone()
            dist.all_reduce(n_all, group=self.pg)
            ce_scale = float(n_loc.item()) * self.world / float(n_all.item())
        out, dc = ops.content_ce(c_logits, plan, grad_scale=ce_scale, want_grad=True, out=self.loss_buf)
        dmu, dlv = torch.zeros_like(mu), torch.zeros_like(lv)
        ops.kld(mu, lv, out, beta=self.beta, dmu=dmu, dlog_var=dlv)
        if self.fix_structure_loss:
            _, ds = ops.bce_logits(s_logits.reshape(-1), s_tensor.reshape(-1), out, 1.0, want_grad=True)
            ds = ds.view_as(s_logits)
        else:                                                # training.py:307: BCE of the target against itself
            ops.bce_logits(s_tensor.reshape(-1), s_tensor.reshape(-1), out, 1.0, want_grad=False)
            ds = None
        dz = eng.decoder_backward(dsv, ds, dc, G)
        self.buckets.launch(2)
        ops.reparam_bwd(dz, lv, eps, dmu, dlv)
        eng.encoder_backward(esv, dmu, dlv, G)
        self.buckets.launch(1)
        self.buckets.launch(0)
        return out

    def train_step(self, graph, eps: Optional[torch.Tensor] = None):
        """One batch of the training loop (training.py:137-172) on `graph` (device batch): forward, losses, backward
        and — every `iters_to_accumulate`-th call — the Adam update and the LR-schedule step.  Returns the device
        tensor [pitch, dur, structure, kld] of this batch's loss values (float64, no host sync)."""
        vae = self.vae
        if not vae.training:
            raise RuntimeError("train_step needs vae.train()")
        if vae.flat_params.data_ptr() != self._flat_ptr:
            raise RuntimeError("the model's flat parameter buffer moved after the trainer was built; rebuild it")
        self.grads.zero_()
        k = self.iters_to_accumulate
        self.buckets.hold = k > 1                      # micro-batches of an accumulation are not all-reduced one by one
        out = (self._native_forward_backward if self.native else self._python_forward_backward)(graph, eps)
        self.micro_batches += 1
        grads = self.grads
        if k > 1:
            # training.py:149: backward of tot_loss / k, summed into .grad; the update waits for the k-th batch (:158)
            ops.grad_accumulate(self.grads, self.grad_accum, 1.0 / k, self.micro_batches % k == 1)
            if self.micro_batches % k != 0:
                return out
            self._accum_bucket.launch(0)
            mean_scale = self._accum_bucket.wait()
            grads = self.grad_accum
        else:
            mean_scale = self.buckets.wait()
        # ---- optimizer (training.py:160-172)
        self.step_count += 1
        ops.adam_step(vae.flat_params, grads, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0],
                      self.betas[1], self.eps, self.step_count, grad_scale=mean_scale)
        if self.sched is not None:
            self.lr = self.sched.step()
        return out

    # ---- evaluation (training.py:250-296 `evaluate`, :298-347 `_losses`, :349-497 `_accuracies`) --------------
    def evaluate_batch(self, graph, eps: Optional[torch.Tensor] = None):
        """One batch of `PolyphemusTrainer.evaluate`: eval-mode forward, the 7 losses and the 9 accuracies of the
        reference (same keys), computed by the loss / metric kernels with ONE host sync at the end (the reference
        takes 16).  Quirk kept (SURVEY B-1): the structure terms are evaluated on the target itself unless the
        trainer was built with `structure_loss_on_logits=True`."""
        vae = self.vae
        was_training = vae.training
        vae.eval()
        try:
            with torch.no_grad():
                mu, lv = vae.encoder(graph)
                e = eps if eps is not None else torch.randn_like(mu)
                z = ops.reparam_fwd(mu.contiguous(), lv.contiguous(), e)
                s_logits, c_logits = vae.decoder(z, graph)
                plan = prepare_graph(graph, vae.cfg["n_bars"])
                s_t = graph.s_tensor.float().contiguous()
                out = torch.zeros(4, dtype=torch.float64, device=mu.device)
                ops.content_ce(c_logits.contiguous(), plan, want_grad=False, out=out)
                ops.kld(mu.contiguous(), lv.contiguous(), out, beta=self.beta)
                s_in = s_logits.reshape(-1).contiguous() if self.fix_structure_loss else s_t.reshape(-1)
                ops.bce_logits(s_in, s_t.reshape(-1), out, 1.0, want_grad=False)
                cc = ops.content_accuracy(c_logits.contiguous(), plan.tokens, plan.is_drum)
                sc = ops.structure_metrics(s_in, s_t.reshape(-1))
                host = torch.cat([out, cc.double(), sc.double()]).tolist()          # the one sync
        finally:
            vae.train(was_training)
        p, d, s, k = host[:4]
        c, m = host[4:12], host[12:16]
        div = lambda a, b: a / b if b else float("nan")
        prec, rec = div(m[1], m[2]), div(m[1], m[3])
        losses = {"tot": p + d + s + self.beta * k, "pitch": p, "dur": d, "structure": s, "reconstruction": p + d + s,
                  "kld": k, "beta*kld": self.beta * k}
        accs = {"note": div(c[6], c[1]), "pitch": div(c[0], c[1]), "pitch_drums": div(c[2], c[3]),
                "pitch_non_drums": div(c[0] - c[2], c[1] - c[3]), "dur": div(c[4], c[5]),
                "s_acc": m[0] / max(s_t.numel(), 1), "s_precision": prec, "s_recall": rec,
                "s_f1": div(2 * rec * prec, rec + prec)}
        return losses, accs

    def evaluate(self, loader):
        """`PolyphemusTrainer.evaluate(loader)` (training.py:250-296): per-batch losses / accuracies averaged over the
        batches of `loader` (plain means of the per-batch values, like the reference's `mean(l)`); restores the
        training mode it found."""
        losses: Dict[str, list] = {}
        accs: Dict[str, list] = {}
        for graph in loader:
            lb, ab = self.evaluate_batch(graph)
            for k, v in lb.items():
                losses.setdefault(k, []).append(v)
            for k, v in ab.items():
                accs.setdefault(k, []).append(v)
        mean = lambda l: sum(l) / len(l)
        return {k: mean(l) for k, l in losses.items()}, {k: mean(l) for k, l in accs.items()}

    # ---- checkpoint interop (training.py:503-519 saves `optimizer.state_dict()` of torch.optim.Adam) ----------
    def optimizer_state_dict(self) -> dict:
        """The fused Adam's state in `torch.optim.Adam.state_dict()` layout: parameter ids follow
        `named_parameters()` order, `exp_avg` / `exp_avg_sq` are per-parameter copies of the flat
        moment buffers; loadable by `torch.optim.Adam(vae.parameters(), ...).load_state_dict`."""
        vae = self.vae
        P = dict(vae.named_parameters())
        state = {}
        for i, n in enumerate(vae._param_names):
            o, k = vae._offsets[n], P[n].numel()
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.exp_avg[o:o + k].view(P[n].shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + k].view(P[n].shape).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(vae._param_names)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd: dict) -> None:
        """Inverse of `optimizer_state_dict`; also accepts a checkpoint written by the reference's
        `torch.optim.Adam` (parameters that never received a gradient have no entry there: their moments stay 0)."""
        vae = self.vae
        P = dict(vae.named_parameters())
        group = sd["param_groups"][0]
        if len(group["params"]) != len(vae._param_names):
            raise ValueError("optimizer state does not match the model's parameter list")
        self.lr, self.betas, self.eps = float(group["lr"]), tuple(group["betas"]), float(group["eps"])
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        steps = set()
        for i, n in enumerate(vae._param_names):
            st = sd["state"].get(group["params"][i])
            if st is None:
                continue
            o, k = vae._offsets[n], P[n].numel()
            self.exp_avg[o:o + k].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): the fused Adam keeps one step count")
        self.step_count = steps.pop() if steps else 0

    def save_checkpoint(self, path: str, **extra) -> None:
        """The reference's checkpoint file (`_save_model`, training.py:498-519): a `torch.save`d dict with
        'model_state_dict' (the 255 reference keys), 'optimizer_state_dict' (torch.optim.Adam layout) and
        'tot_batches'; `extra` carries the bookkeeping entries of the reference's loop (epoch, lrs, ...).
        `generate.load_model` of the reference reads 'model_state_dict' from it."""
        ckpt = dict(extra)
        ckpt.update(tot_batches=self.micro_batches, dropout_stream={"seed": self.vae.seed, "step": self.vae._step},
                    model_state_dict={k: v.detach().cpu().clone() for k, v in self.vae.state_dict().items()},
                    optimizer_state_dict=self.optimizer_state_dict())
        torch.save(ckpt, path)

    def load_checkpoint(self, path: str) -> dict:
        """Restore model, Adam moments, step count and LR-schedule position from `save_checkpoint` / a reference
        checkpoint; returns the remaining entries."""
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        self.vae.load_state_dict(ckpt.pop("model_state_dict"))
        self.load_optimizer_state_dict(ckpt.pop("optimizer_state_dict"))
        self.micro_batches = int(ckpt.get("tot_batches", self.step_count * self.iters_to_accumulate))
        ds = ckpt.pop("dropout_stream", None)           # position of the counter-based dropout stream: a resumed run
        if ds is not None:                              # continues with fresh masks instead of replaying the old ones
            self.vae.seed, self.vae._step = int(ds["seed"]), int(ds["step"])
        if self.sched is not None:
            self.sched.update_steps = self.step_count
        return ckpt

    def losses_dict(self, out: torch.Tensor) -> dict:
        """Host copy of the loss vector in the reference's dict layout (this DOES sync)."""
        p, d, s, k = out.tolist()
        rec = p + d + s
        return {"tot": rec + self.beta * k, "pitch": p, "dur": d, "structure": s, "reconstruction": rec, "kld": k,
                "beta*kld": self.beta * k}
