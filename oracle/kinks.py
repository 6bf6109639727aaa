"""ReLU-kink bookkeeping for the fp64 oracle — ORACLE / TEST INFRASTRUCTURE ONLY (see vae_cpu.py's header: only tests/,
__graft_entry__.smoke(), bench.py's cpu_baseline leg and tools/ diagnostics may import anything under oracle/).

The training loss is piecewise smooth in the parameters: every ReLU of the step (model.py:132,205,...) has a kink at 0,
and the gradient jumps when a pre-activation changes sign.  An fp32 implementation whose activations are within ~1e-6
of the exact ones can therefore sit on the other side of a kink than the fp64 oracle at the handful of elements whose
pre-activation is that close to zero — and one flipped element of an [N, d] layer moves the upstream gradient by
~1/sqrt(N d) in relative L2 (percent-level on the small batches of the smoke test), although both gradients are valid
one-sided derivatives of the same function.  `kink_gradients` makes that testable:

  * it runs the oracle step once (g0) and records every ReLU site's pre-activations;
  * the NEAR-KINK SET K = the elements with 0 < |pre| < tau * rms(site) (exact zeros are excluded: they come from exact
    zeros upstream and every implementation treats them alike);
  * for each k in K it re-runs the step with that single ReLU decision inverted: delta_k = g(flip k) - g0.

`explain(g, ...)` then fits g - g0 = sum_k m_k delta_k by least squares: a gradient that is correct up to ReLU decisions
inside the tau-band has all m_k in {0, 1} (within `binary_tol`) and a residual at the usual fp32-vs-fp64 level; a wrong
kernel or a race leaves a residual.  Follows: oracle/vae_cpu.py (the op sequence), reference training.py:137-166."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as _F

from . import vae_cpu


class _FProxy:
    """torch.nn.functional with `relu` replaced (installed as vae_cpu.F for the duration of a probe)."""

    def __init__(self, relu):
        self.relu = relu

    def __getattr__(self, k):
        return getattr(_F, k)


class ReluProbe:
    """Context manager: records the pre-activations of every ReLU the oracle executes (call order = site index) and
    inverts the decisions listed in `flips` {site: LongTensor of flat element indices}; `forced` {site: bool tensor of the
    site's shape} replaces a site's decisions altogether (relu(x) becomes x * mask: the decisions another implementation took)."""

    def __init__(self, flips: Optional[Dict[int, torch.Tensor]] = None, keep: bool = True,
                 forced: Optional[Dict[int, torch.Tensor]] = None):
        self.flips = flips or {}
        self.forced = forced or {}
        self.keep = keep
        self.disagree: Dict[int, int] = {}     # per forced site: decisions that differ from the oracle's own
        self.pre: List[torch.Tensor] = []
        self.count = 0

    def _relu(self, x, *a, **k):
        i = self.count
        self.count += 1
        if self.keep:
            self.pre.append(x.detach().clone())
        if i in self.forced:
            mask = self.forced[i]
            assert mask.shape == x.shape, (i, tuple(mask.shape), tuple(x.shape))
            self.disagree[i] = int((mask != (x.detach() > 0)).sum())
            return x * mask.to(x.dtype)
        if i not in self.flips:
            return _F.relu(x)
        mask = x.detach() > 0
        idx = self.flips[i]
        flat = mask.reshape(-1)
        flat[idx] = ~flat[idx]
        return x * flat.reshape(x.shape).to(x.dtype)

    def __enter__(self):
        self._saved = vae_cpu.F
        vae_cpu.F = _FProxy(self._relu)
        return self

    def __exit__(self, *exc):
        vae_cpu.F = self._saved
        return False


def fp64_inputs(cpu_batch, sd, names):
    """(batch, leaf parameters) of the oracle in fp64 from a CPU batch and an fp32 state_dict."""
    P, names = vae_cpu.split_state({k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}, names)
    b64 = cpu_batch.to("cpu")
    b64.__dict__["_c_tensor"], b64.__dict__["_edge_attrs"] = cpu_batch.c_tensor.double(), cpu_batch.edge_attrs.double()
    b64.s_tensor = cpu_batch.s_tensor.double()
    return b64, P, names


def relu_sites(cfg) -> List[str]:
    """Names of the oracle's ReLU sites in call order (vae_cpu.vae_forward with batch_norm = True): the index space of
    `ReluProbe.flips / .forced`."""
    assert cfg["batch_norm"]
    L = cfg["gnn_n_layers"]

    def gcn(tag):
        out = []
        for i in range(L):
            out += [f"{tag}.{i}.msg.{r}" for r in range(6)] + [f"{tag}.{i}.norm"]
        return out
    return (["enc_cnn.bn1", "enc_cnn.bn5", "enc_cnn.lin1", "enc_chord.drums", "enc_chord.non_drums"] + gcn("enc_gcn") +
            ["enc_merge", "dec_bn", "dec_cnn.lin1", "dec_cnn.lin4", "dec_cnn.bn2"] + gcn("dec_gcn"))


def _step(cpu_batch, sd, names, cfg, eps, msg_dropout, keep_mask, flips=None, keep=False, forced=None):
    b64, P, names = fp64_inputs(cpu_batch, sd, names)
    with ReluProbe(flips, keep, forced) as probe:
        (s_logits, c_logits), mu, log_var = vae_cpu.vae_forward(b64, P, cfg, True, eps.double(), msg_dropout, keep_mask)
        tot, parts = vae_cpu.losses(b64.s_tensor, s_logits, b64.c_tensor, c_logits, mu, log_var)
    tot.backward()
    grads = {n: (None if P[n].grad is None else P[n].grad.detach().clone()) for n in names}
    return grads, {k: float(v) for k, v in parts.items()}, probe, P


def flat_grad(grads: Dict[str, Optional[torch.Tensor]], names, like: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """Concatenation (fp64) of the gradients of `names`; tensors the loss does not reach (None) count as zeros of the
    shape given by `like`, or are skipped when `like` is None."""
    parts = []
    for n in names:
        g = grads.get(n)
        if g is None:
            if like is None:
                continue
            g = torch.zeros_like(like[n], dtype=torch.float64)
        parts.append(g.detach().double().reshape(-1).cpu())
    return torch.cat(parts)


def kink_gradients(cpu_batch, sd, names, cfg, eps, msg_dropout: float = 0.0, keep_mask=None, tau: float = 2e-5,
                   max_kinks: int = 96):
    """The oracle step plus the single-flip gradient changes of its near-kink ReLU decisions.

    Returns dict(g0 = {name: grad}, losses, used = names with a gradient, kinks = [(site, flat index, |pre|, rms)],
    deltas = [K, n] fp64 over `used`, margins = per-site (shape, smallest nonzero |pre| / rms))."""
    g0, parts, probe, _ = _step(cpu_batch, sd, names, cfg, eps, msg_dropout, keep_mask, keep=True)
    used = [n for n in names if g0[n] is not None]
    cand: List[Tuple[float, int, int, float, float]] = []
    margins = []
    for i, pre in enumerate(probe.pre):
        if pre.numel() == 0:
            margins.append((tuple(pre.shape), float("inf")))
            continue
        a = pre.abs().reshape(-1)
        rms = float((pre ** 2).mean().sqrt())
        nz = a[a > 0]
        margins.append((tuple(pre.shape), float(nz.min()) / rms if nz.numel() and rms > 0 else float("inf")))
        hit = torch.nonzero((a > 0) & (a < tau * rms)).reshape(-1)
        for j in hit.tolist():
            cand.append((float(a[j]) / rms, i, j, float(a[j]), rms))
    cand.sort()
    cand = cand[:max_kinks]
    f0 = flat_grad(g0, used)
    deltas = torch.zeros(len(cand), f0.numel(), dtype=torch.float64)
    for k, (_, site, j, _, _) in enumerate(cand):
        gk, _, _, _ = _step(cpu_batch, sd, names, cfg, eps, msg_dropout, keep_mask, flips={site: torch.tensor([j])})
        deltas[k] = flat_grad(gk, used, like=g0) - f0
    return dict(g0=g0, f0=f0, losses=parts, used=used, kinks=[(s, j, a, r) for _, s, j, a, r in cand], deltas=deltas,
                margins=margins, tau=tau)


def explain(g: torch.Tensor, ref: dict, tol: float = 1e-4, min_delta: float = 1e-4):
    """Attribution of a gradient `g` (flat fp64 over ref['used']) to the near-kink decisions of `ref`.

    Only decisions whose single-flip change is at least `min_delta` of |g0| enter (smaller ones cannot be told from fp32
    rounding; `tol` bounds what they may add up to).  A ReLU decision is taken or not, so the coefficients are BINARY:
    a least-squares fit is rounded to {0, 1} and refined by coordinate descent (single decisions toggled while that
    lowers the residual — the changes of two decisions on one path can be collinear, which leaves the real-valued fit
    ambiguous).  Returns dict(raw = |g - g0| / |g0|, residual = |g - g0 - sum_k m_k delta_k| / |g0| for the binary m,
    flips = [(site, index, 1.0)] of the decisions taken the other way, ok = residual < tol)."""
    f0, D = ref["f0"], ref["deltas"]
    den = float(f0.norm())
    b = g.double() - f0
    raw = float(b.norm()) / den
    keep = [k for k in range(D.shape[0]) if float(D[k].norm()) >= min_delta * den]
    if not keep or raw < 0.3 * min_delta:
        return dict(raw=raw, residual=raw, flips=[], ok=raw < tol)
    A = D[keep].T
    m = torch.linalg.lstsq(A, b.unsqueeze(1), driver="gelsd").solution.reshape(-1).round().clamp(0, 1)
    r = b - A @ m
    best = float(r.norm())
    improved = True
    while improved:
        improved = False
        for j in range(len(keep)):
            cand = r + A[:, j] * (1.0 if m[j] == 1 else -1.0)          # toggle decision j
            c = float(cand.norm())
            if c < best * (1 - 1e-9):
                m[j] = 1 - m[j]
                r, best, improved = cand, c, True
    res = best / den
    flips = [(ref["kinks"][k][0], ref["kinks"][k][1], 1.0) for k, v in zip(keep, m) if v == 1]
    return dict(raw=raw, residual=res, flips=flips, ok=bool(res < tol))
