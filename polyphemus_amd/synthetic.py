"""Synthetic LMD-shape samples — the bench / test workload.

`numpy.random.default_rng(seed)`; structure s[nb,4,32] ~ Bernoulli(p); an empty
bar gets cell [0,0]; each active cell holds k ~ U{1..4} notes with
pitch ~ U{0..127}, duration token ~ U{0..95}, laid out
[SOS, n_1..n_k, EOS, PAD...] over the 16 slots (reference on-disk format:
preprocess.py:118-149,210).
"""
from __future__ import annotations

import numpy as np

from . import constants as C
from .graphs import BarGraphBatch, collate_samples, graph_from_structure


def disk_sample(rng: np.random.Generator, n_bars: int, p: float, max_notes: int = 4):
    """One sample in the reference's `.npz` layout (preprocess.py:210):
    c_tensor int16 [4, nb*32, 16, 2], s_tensor bool [4, nb*32].  `max_notes`: notes per active cell are U{1..max_notes}
    (4; 14 fills all 15 token slots behind SOS, constants.py:48)."""
    if not 1 <= max_notes <= C.MAX_SIMU_TOKENS - 2:
        raise ValueError("max_notes must be in 1..14")
    T = n_bars * C.N_TIMESTEPS
    s = rng.random((C.N_TRACKS, T)) < p
    for b in range(n_bars):                         # data.py:152-153 forces [0,0] on
        if not s[:, b * C.N_TIMESTEPS:(b + 1) * C.N_TIMESTEPS].any():
            s[0, b * C.N_TIMESTEPS] = True
    c = np.empty((C.N_TRACKS, T, C.MAX_SIMU_TOKENS, 2), np.int16)
    c[..., 0] = C.PITCH_PAD
    c[..., 1] = C.DUR_PAD
    k = rng.integers(1, max_notes + 1, size=(C.N_TRACKS, T))
    pitch = rng.integers(0, 128, size=(C.N_TRACKS, T, max_notes))
    dur = rng.integers(0, 96, size=(C.N_TRACKS, T, max_notes))
    c[..., 0, 0] = C.PITCH_SOS
    c[..., 0, 1] = C.DUR_SOS
    for j in range(max_notes):
        has = k > j
        c[..., 1 + j, 0] = np.where(has, pitch[..., j], c[..., 1 + j, 0])
        c[..., 1 + j, 1] = np.where(has, dur[..., j], c[..., 1 + j, 1])
    tr, ts = np.nonzero(np.ones_like(k, bool))
    c[tr, ts, 1 + k[tr, ts], 0] = C.PITCH_EOS
    c[tr, ts, 1 + k[tr, ts], 1] = C.DUR_EOS
    return c, s


def sample_from_disk(c_disk: np.ndarray, s_disk: np.ndarray, n_bars: int, dense=False):
    """`.npz` tensors -> per-sample graph dict (restates PolyphemusDataset.__getitem__,
    data.py:218-271, but keeps token *ids* instead of 14.7 KB/node one-hots)."""
    c = c_disk.reshape(C.N_TRACKS, n_bars, C.N_TIMESTEPS, C.MAX_SIMU_TOKENS, 2).transpose(1, 0, 2, 3, 4)
    s = np.ascontiguousarray(s_disk.reshape(C.N_TRACKS, n_bars, C.N_TIMESTEPS).transpose(1, 0, 2)).astype(bool)
    g = graph_from_structure(s, dense=dense)        # may switch [0,0] on in empty bars
    g["tokens"] = c.reshape(-1, C.MAX_SIMU_TOKENS, 2)[s.reshape(-1)].astype(np.int32)
    g["s_tensor"] = s.astype(np.float32)
    return g


def synthetic_batch(batch_size: int, n_bars: int = 2, p: float = 0.25, seed: int = 1234,
                    dense: bool = False, max_notes: int = 4) -> BarGraphBatch:
    """A collated batch of `batch_size` synthetic samples (CPU tensors)."""
    rng = np.random.default_rng(seed)
    samples = []
    for _ in range(batch_size):
        c, s = disk_sample(rng, n_bars, 1.0 if dense else p, max_notes)
        samples.append(sample_from_disk(c, s, n_bars, dense=dense))
    return collate_samples(samples, n_bars)
