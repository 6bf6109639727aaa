"""Data-parallel host logic on CPU (gloo, world_size 2): sharding, bucketed gradient
all-reduce, parameter broadcast, and DP-equivalence of the averaged gradient (oracle)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from polyphemus_amd.parallel import GradBuckets, broadcast_, shard_range


def test_shard_range_covers_batch_without_overlap():
    for n, w in ((256, 8), (10, 4), (7, 2), (3, 4)):
        seen = []
        for r in range(w):
            seen += list(shard_range(n, r, w))
        assert seen == list(range(n))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)
        flat = torch.randn(1000)
        mine = flat.clone()
        gb = GradBuckets(flat, [384])
        assert gb.world == world and [v.numel() for v in gb.views] == [384, 616]
        gb.launch(1)                      # decoder bucket first, as the trainer does
        gb.launch(0)
        scale = gb.wait()
        allg = [torch.zeros(1000) for _ in range(world)]
        dist.all_gather(allg, mine)
        ok = torch.allclose(flat, sum(allg)) and scale == 1.0 / world
        p = torch.full((16,), float(rank))
        broadcast_([p], 0)
        ok = ok and bool((p == 0).all())
        # gradient accumulation (trainer.iters_to_accumulate > 1): micro-batch launches are held back, the running sum
        # travels as ONE bucket on the k-th batch
        held = mine.clone()
        gh = GradBuckets(held, [384])
        gh.hold = True
        gh.launch(1); gh.launch(0)
        ok = ok and gh.wait() == 1.0 / world and torch.equal(held, mine)
        acc = mine.clone()
        one = GradBuckets(acc, [])
        one.launch(0)
        one.wait()
        ok = ok and len(one.views) == 1 and torch.allclose(acc, sum(allg))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_and_broadcast_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_single_process_is_a_noop():
    flat = torch.arange(10.0)
    gb = GradBuckets(flat, [4])
    gb.launch(0); gb.launch(1)
    assert gb.wait() == 1.0 and torch.equal(flat, torch.arange(10.0))
