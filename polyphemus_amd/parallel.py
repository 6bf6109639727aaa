"""Data-parallel plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

The reference is single-device (train.py:120-122).  Bar-graphs of different samples never share
an edge, so the batch shards by sample with no data-path collective; the only exchange is ONE
sum-all-reduce of the flat fp32 gradient buffer per optimizer step (43 MB at d=256, 169 MB at
d=512), issued as three buckets — decoder gradients as soon as the decoder backward has finished, the graph
encoder .. encoder head after the first half of the encoder backward, the rest at the end — so the two large
buckets overlap the encoder backward.
BatchNorm uses per-replica statistics (standard DDP semantics; >= 16 k nodes per replica).
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun).

    Call it BEFORE anything initialises HIP in this process (`torch.cuda.is_available()`, any tensor on the GPU): the
    dmabuf-IPC switch below is read by the runtime when it starts and has no effect afterwards — RCCL then fails with
    `hipIpcGetMemHandle: invalid argument`.  Launchers that start the ranks (bench.py, tests/util.run_ranks) put the
    variable into the child environment themselves; this default only covers a bare `torchrun`."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # PM_DP_FORCE=1 (tests): take the exchange path with a single rank too, so that the RCCL stream semantics of the
    # bucketed all-reduce can be exercised on a one-GPU box
    if (world > 1 or os.environ.get("PM_DP_FORCE") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL / tensor sharing across processes)
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        elif torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        import datetime
        # a rank that dies before / during the rendezvous fails the others after two minutes instead of the 30-minute default
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    return rank, local, world


def shard_range(n_samples: int, rank: int, world: int) -> range:
    """Contiguous shard of the global batch owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_samples, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


class GradBuckets:
    """Bucketed sum-all-reduce of a flat gradient buffer.  `boundaries` are element offsets that
    split the buffer into buckets; `launch(i)` may be called as soon as bucket i is final."""

    def __init__(self, flat_grads: torch.Tensor, boundaries: Sequence[int], group=None):
        self.flat, self.group = flat_grads, group
        edges = [0] + list(boundaries) + [flat_grads.numel()]
        self.views: List[torch.Tensor] = [flat_grads[a:b] for a, b in zip(edges[:-1], edges[1:])]
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.active = self.world > 1 or (os.environ.get("PM_DP_FORCE") == "1" and dist.is_available() and dist.is_initialized())
        self._work = []
        self.hold = False        # True: launches are skipped (micro-batches of a gradient accumulation)
        self.disabled = False    # True: no exchange at all (bench.py: cost of the step without the all-reduce)
        # trace = True: every launch / completion of the next exchange is stamped (bench.py's `dp` block; off in timed
        # regions: a stamp on a GPU stream is an event record, i.e. a barrier packet)
        self.trace = False
        self._marks: List[dict] = []
        self._wait0 = None

    def _stamp(self):
        """a point in the order of the COMPUTE stream (HIP event) for GPU buffers, the host clock otherwise"""
        if self.flat.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            return ev
        import time
        return time.perf_counter()

    def launch(self, i: int) -> None:
        if self.active and not self.hold and not self.disabled:
            if self.trace:
                if not self._work:
                    self._marks = []
                self._marks.append({"bucket": i, "bytes": int(self.views[i].numel() * self.views[i].element_size()),
                                    "launch": self._stamp()})
            self._work.append(dist.all_reduce(self.views[i], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self) -> float:
        """Wait for every launched bucket; returns the factor that turns the sum into the mean."""
        tr = self.trace and len(self._marks) == len(self._work)
        if tr:
            self._wait0 = self._stamp()
        for k, w in enumerate(self._work):
            w.wait()                      # (RCCL: the compute stream waits for the bucket; gloo: the host does)
            if tr:
                self._marks[k]["done"] = self._stamp()
        self._work = []
        return 1.0 / self.world

    def timeline(self) -> List[dict]:
        """Per bucket of the last traced exchange, in milliseconds on the compute stream's clock: when it was launched
        (relative to the first launch), the `window` of compute that ran between its launch and the point where the
        step needs the gradients (wait), and the time the step then stood `exposed`, waiting for this bucket after the
        previous one had arrived.  window >> 0 and exposed ~ 0 is what "overlapped with the backward" means."""
        if not self._marks or self._wait0 is None or "done" not in self._marks[-1]:
            return []
        if self.flat.is_cuda:
            torch.cuda.synchronize()
            dt = lambda a, b: a.elapsed_time(b)
        else:
            dt = lambda a, b: 1e3 * (b - a)
        out, prev = [], self._wait0
        for m in self._marks:
            out.append({"bucket": m["bucket"], "bytes": m["bytes"], "launched_at_ms": round(dt(self._marks[0]["launch"], m["launch"]), 3),
                        "window_ms": round(dt(m["launch"], self._wait0), 3), "exposed_ms": round(max(dt(prev, m["done"]), 0.0), 3)})
            prev = m["done"]
        return out


def broadcast_(tensors: Sequence[torch.Tensor], src: int = 0, group=None) -> None:
    """Make every rank start from rank `src`'s parameters / buffers."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for t in tensors:
            dist.broadcast(t, src=src, group=group)


class NativeComm:
    """RCCL communicator behind the C ABI (`pm_comm_*`, `pm_allreduce`, csrc/comm.hip): what a host that is not
    PyTorch binds.  `NativeComm.from_torch_group()` bootstraps one next to an initialised torch.distributed group (the
    128-byte id travels as a broadcast object); `NativeComm.single()` is the one-rank communicator."""

    def __init__(self, id128: bytes, rank: int, world: int):
        import ctypes
        from ._lib import call
        self.rank, self.world = rank, world
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(id128)
        h = ctypes.c_void_p()
        call("pm_comm_init", ctypes.cast(buf, ctypes.c_void_p), rank, world, ctypes.cast(ctypes.pointer(h), ctypes.c_void_p))
        self._h = h

    @staticmethod
    def unique_id() -> bytes:
        import ctypes
        from ._lib import call
        buf = (ctypes.c_uint8 * 128)()
        call("pm_comm_unique_id", ctypes.cast(buf, ctypes.c_void_p))
        return bytes(buf)

    @classmethod
    def single(cls) -> "NativeComm":
        return cls(cls.unique_id(), 0, 1)

    @classmethod
    def from_torch_group(cls, group=None) -> "NativeComm":
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(box[0], rank, world)

    def all_reduce(self, t: torch.Tensor) -> None:
        """In-place sum of a contiguous fp32 cuda tensor over the ranks, on the current stream."""
        from ._lib import call, ptr, stream
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise ValueError("NativeComm.all_reduce needs a contiguous fp32 cuda tensor")
        call("pm_allreduce", ptr(t), t.numel(), self._h, stream())

    def close(self) -> None:
        from ._lib import call
        if self._h:
            call("pm_comm_destroy", self._h)
            self._h = None
