"""Host mirror of the reference's generation entry points (generate.py:21-37, 90-98) over the device kernels of
csrc/generate.hip and csrc/graph.hip: the decoder pass, the thresholded structure, and the dense multitrack pianoroll,
without the reference's host synchronisations (`nonzero`, per-sample Python graph building, three full-tensor writes).
Converting the pianoroll to MIDI (muspy) is outside the hot path and stays with the caller."""
from __future__ import annotations

import torch

from . import ops


def generate_z(bs: int, d_model: int, device) -> torch.Tensor:
    """generate.py:90-98: a standard-normal latent batch."""
    return torch.randn(bs, d_model, device=device)


def generate_music(vae, z, s_cond=None, s_tensor_cond=None):
    """generate.py:21-37.  `s_cond` is a batch of bar graphs (`vae.decoder._structure_from_binary`) or None (the
    structure then comes from the decoder's own thresholded logits); `s_tensor_cond` [B,n_bars,4,32] is the binary
    structure the pianoroll is laid out on when given.  Returns (mtp [B,n_bars,4,32,15,230], s_tensor bool)."""
    vae.decoder.__dict__.pop("_last_structure", None)
    s_logits, c_logits = vae.decoder(z, s_cond)
    if s_tensor_cond is not None:
        s_tensor = s_tensor_cond
    elif s_cond is None:                 # the structure the decoder itself thresholded and built its graphs from
        s_tensor = vae.decoder.__dict__["_last_structure"].view(s_logits.shape).bool()
    else:
        s_tensor = vae.decoder._binary_from_logits(s_logits)
    mtp = ops.mtp_from_logits(c_logits.detach().contiguous().float(), s_tensor)
    return mtp, s_tensor
