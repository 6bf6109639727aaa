"""Dataset + device-side collate: the input side of the hot path.

The reference's `PolyphemusDataset.__getitem__` (data.py:218-271) expands every sample to one-hot `c_tensor`s
(14.7 KB per node) and builds its bar graphs with Python loops on the host (6-170 ms per sample), then PyG's collate
concatenates them (train.py:152).  Here a sample stays in its on-disk form — token ids int16 [nb,4,32,16,2] and the
activation grid [nb,4,32] (preprocess.py:118-149,210) — a batch is two pinned host arrays (4.2 MB + 64 KB at B = 256),
one H2D copy each on a side stream, and the batch of bar graphs is built ON THE DEVICE (`csrc/graph.hip`) with the
reference's node numbering and edge order, so the result is interchangeable bit for bit with the reference's
DataLoader output (tests/test_data_gpu.py on the samples the reference itself collated).
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import constants as C
from .graphs import BarGraphBatch, device_batch_from_structure


class PolyphemusDataset:
    """Mirror of the reference's `PolyphemusDataset(dir, n_bars)` (data.py:206-216) over the same `.npz` files
    (`c_tensor` int [4, nb*32, 16, 2], `s_tensor` bool [4, nb*32]).  `__getitem__` returns the sample re-laid out bar
    by bar (data.py:226-233) as token ids; no one-hots, no graph — those are the device's job.  Files are taken in
    sorted order (the reference uses `os.scandir` order, which is arbitrary)."""

    def __init__(self, dir: str, n_bars: int = 2):
        self.dir = dir
        self.files = sorted(e.name for e in os.scandir(dir) if e.is_file())
        self.len = len(self.files)
        self.n_bars = n_bars

    def __len__(self) -> int:
        return self.len

    def __getitem__(self, idx: int) -> Tuple[np.ndarray, np.ndarray]:
        with np.load(os.path.join(self.dir, self.files[idx])) as data:
            return relayout_sample(data["c_tensor"], data["s_tensor"], self.n_bars)


def relayout_sample(c_disk: np.ndarray, s_disk: np.ndarray, n_bars: int) -> Tuple[np.ndarray, np.ndarray]:
    """(n_tracks x n_bars*32 x ...) -> (n_bars x n_tracks x 32 x ...), data.py:226-233: token grid int16
    [nb,4,32,16,2] and activation grid uint8 [nb,4,32]."""
    if c_disk.shape != (C.N_TRACKS, n_bars * C.N_TIMESTEPS, C.MAX_SIMU_TOKENS, 2):
        raise ValueError(f"c_tensor has shape {c_disk.shape}, expected {(C.N_TRACKS, n_bars * C.N_TIMESTEPS, C.MAX_SIMU_TOKENS, 2)}")
    if s_disk.shape != (C.N_TRACKS, n_bars * C.N_TIMESTEPS):
        raise ValueError(f"s_tensor has shape {s_disk.shape}, expected {(C.N_TRACKS, n_bars * C.N_TIMESTEPS)}")
    c = c_disk.reshape(C.N_TRACKS, n_bars, C.N_TIMESTEPS, C.MAX_SIMU_TOKENS, 2).transpose(1, 0, 2, 3, 4)
    s = s_disk.reshape(C.N_TRACKS, n_bars, C.N_TIMESTEPS).transpose(1, 0, 2)
    return np.ascontiguousarray(c, dtype=np.int16), np.ascontiguousarray(s, dtype=np.uint8)


def collate_on_device(samples: Sequence[Tuple[np.ndarray, np.ndarray]], n_bars: int, device="cuda",
                      staging: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> BarGraphBatch:
    """Batch of `PolyphemusDataset` samples -> the `BarGraphBatch` the reference's DataLoader would deliver
    (PyG collate of the per-sample graphs), built on the device.  `staging` = optional pinned host
    buffers (token grid int16 [>=B,nb,4,32,16,2], structure uint8 [>=B,nb,4,32]) to copy through."""
    B = len(samples)
    if B == 0:
        raise ValueError("empty batch")
    if staging is None:
        staging = _staging(B, n_bars, pin=False)
    tok_h, s_h = staging[0][:B], staging[1][:B]
    for i, (c, s) in enumerate(samples):
        tok_h[i] = torch.from_numpy(c)
        s_h[i] = torch.from_numpy(s)
    # active token slots of the batch (`PmBatch.n_slots`): the last slot that holds a non-PAD token in any ACTIVE cell
    # (an empty bar's cell [0,0] counts: the graph builder switches it on, data.py:152-153) — 4 MB of int16 on the host
    s_np, tok_np = s_h.numpy().astype(bool), tok_h.numpy()
    empty = ~s_np.any(axis=(-1, -2))
    if empty.any():
        s_np = s_np.copy()
        s_np[..., 0, 0] |= empty
    live = ((tok_np[..., 1:, 0] != C.PITCH_PAD) | (tok_np[..., 1:, 1] != C.DUR_PAD)) & s_np[..., None]
    slots = np.nonzero(live.reshape(-1, C.MAX_SIMU_TOKENS - 1).any(axis=0))[0]
    tok = tok_h.to(device, non_blocking=True)
    s = s_h.to(device, non_blocking=True)
    batch = device_batch_from_structure(s.view(B * n_bars, C.N_TRACKS, C.N_TIMESTEPS), n_bars, token_grid=tok)
    batch.n_slots = int(slots.max()) + 1 if slots.size else 1
    return batch


def _staging(B: int, n_bars: int, pin: bool):
    tok = torch.empty(B, n_bars, C.N_TRACKS, C.N_TIMESTEPS, C.MAX_SIMU_TOKENS, 2, dtype=torch.int16)
    s = torch.empty(B, n_bars, C.N_TRACKS, C.N_TIMESTEPS, dtype=torch.uint8)
    return (tok.pin_memory(), s.pin_memory()) if pin else (tok, s)


class DeviceLoader:
    """`DataLoader(dataset, batch_size, shuffle)` of train.py:152-156 with the collate on the device.  Batches are
    staged through two sets of pinned host buffers and copied + built on a side stream one batch ahead of the
    consumer: file reading (`num_workers` threads), the H2D copy and the graph kernels of batch i+1 overlap the
    (asynchronous) training step of batch i.  Partial last batches are kept (`drop_last=False`, the DataLoader default
    the reference uses)."""

    def __init__(self, dataset, batch_size: int, shuffle: bool = False, seed: int = 0, device="cuda",
                 drop_last: bool = False, num_workers: int = 0, rank: Optional[int] = None, world: Optional[int] = None):
        """`batch_size` is PER RANK.  Under data parallelism (`rank` / `world`, default: the torch.distributed default
        group) the loader behaves like `DistributedSampler`: every rank draws the SAME permutation (seed + epoch),
        pads it by wrap-around to a multiple of `world` and keeps every world-th index starting at its rank — disjoint
        shards, the same number of batches on every rank (the gradient all-reduce needs the ranks in lock step)."""
        import torch.distributed as dist
        ddp = dist.is_available() and dist.is_initialized()
        self.rank = int(rank) if rank is not None else (dist.get_rank() if ddp else 0)
        self.world = int(world) if world is not None else (dist.get_world_size() if ddp else 1)
        if not 0 <= self.rank < self.world:
            raise ValueError(f"rank {self.rank} outside world {self.world}")
        self.dataset, self.batch_size, self.shuffle, self.seed = dataset, int(batch_size), shuffle, seed
        self.device, self.drop_last = torch.device(device), drop_last
        self._pool = ThreadPoolExecutor(num_workers) if num_workers > 0 else None
        self.n_bars = dataset.n_bars
        self.epoch = 0
        self._stage = [_staging(self.batch_size, self.n_bars, pin=True) for _ in range(2)]
        self._stream = torch.cuda.Stream(device=self.device)

    def _shard_len(self) -> int:
        return (len(self.dataset) + self.world - 1) // self.world

    def __len__(self) -> int:
        n = self._shard_len()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _index_batches(self) -> List[np.ndarray]:
        n = len(self.dataset)
        order = np.random.default_rng(self.seed + self.epoch).permutation(n) if self.shuffle else np.arange(n)
        if self.world > 1:
            total = self._shard_len() * self.world
            # DistributedSampler's rule: wrap around — repeatedly when the dataset is smaller than the world (np.resize
            # tiles) — so that every rank gets the same number of samples and batches: the all-reduce needs lock step
            order = np.resize(order, total)[self.rank::self.world]
            n = order.shape[0]
        out = [order[i:i + self.batch_size] for i in range(0, n, self.batch_size)]
        if self.drop_last and out and len(out[-1]) < self.batch_size:
            out.pop()
        return out

    def _build(self, idx: np.ndarray, slot: int):
        """enqueue copy + graph construction of one batch on the side stream; returns (batch, ready event)"""
        if self._pool is not None:
            samples = list(self._pool.map(self.dataset.__getitem__, [int(i) for i in idx]))
        else:
            samples = [self.dataset[int(i)] for i in idx]
        with torch.cuda.stream(self._stream):
            batch = collate_on_device(samples, self.n_bars, self.device, self._stage[slot])
            ev = torch.cuda.Event()
            ev.record(self._stream)
        return batch, ev

    def __iter__(self) -> Iterator[BarGraphBatch]:
        batches = self._index_batches()
        self.epoch += 1
        if not batches:
            return
        nxt = self._build(batches[0], 0)
        for j in range(len(batches)):
            batch, ev = nxt
            # `device_batch_from_structure` reads (N, E) back, so the side stream has finished this batch's copy: the
            # other staging slot is free to be refilled while the consumer trains on `batch`
            if j + 1 < len(batches):
                nxt = self._build(batches[j + 1], (j + 1) & 1)
            torch.cuda.current_stream(self.device).wait_event(ev)
            for v in batch.__dict__.values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(torch.cuda.current_stream(self.device))
            yield batch
