"""Data-parallel host logic on CPU (gloo, world_size 2): sharding, bucketed gradient
all-reduce, parameter broadcast, and DP-equivalence of the averaged gradient (oracle)."""
import datetime

import pytest
import torch
import torch.distributed as dist

from polyphemus_amd.parallel import GradBuckets, broadcast_, shard_range
from util import run_ranks


def test_shard_range_covers_batch_without_overlap():
    for n, w in ((256, 8), (10, 4), (7, 2), (3, 4)):
        seen = []
        for r in range(w):
            seen += list(shard_range(n, r, w))
        assert seen == list(range(n))


def _worker(rank, world):
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
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
        return ok
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_and_broadcast_gloo_world2():
    assert run_ranks(_worker, 2, timeout=90.0) == [True, True]


def test_single_process_is_a_noop():
    flat = torch.arange(10.0)
    gb = GradBuckets(flat, [4])
    gb.launch(0); gb.launch(1)
    assert gb.wait() == 1.0 and torch.equal(flat, torch.arange(10.0))


def _raiser(rank, world):
    if rank == 1:
        raise ValueError("boom on rank 1")
    import time
    time.sleep(30)                       # rank 0 would wait in a collective for ever


def _sleeper(rank, world):
    import time
    time.sleep(60)


def test_harness_reports_a_failing_rank_and_kills_the_others():
    """The round-1 GPU suite hung on a rank that never answered: the harness must turn both cases into a prompt
    exception and leave no child alive."""
    import multiprocessing
    import time
    from util import RanksHung
    t0 = time.time()
    with pytest.raises(AssertionError, match="boom on rank 1"):
        run_ranks(_raiser, 2, timeout=40.0)
    with pytest.raises(RanksHung, match="rank 0"):
        run_ranks(_sleeper, 2, timeout=16.0)
    assert time.time() - t0 < 45.0
    assert not multiprocessing.active_children()


def _bench(*extra, env=None, gpus=2):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--stub-step", "--steps", "3",
                        "--warmup", "1", *extra], capture_output=True, text=True, timeout=240,
                       env=dict(os.environ, **(env or {})))
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r.returncode, lines, r.stderr


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` (no torchrun) must start both ranks itself; the stub step keeps the launcher, the
    rendezvous, the bucketed exchange and the JSON contract and drops only the GPU work."""
    rc, lines, err = _bench()
    assert rc == 0, err
    assert len(lines) == 1, lines                                  # ONE line, from rank 0
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    tl = line["dp"].pop("buckets_timeline")
    assert line["dp"] == {"ranks_seen": 2, "backend": "gloo", "allreduce_checksum_ok": True}
    # one traced exchange: the three buckets in launch order (decoder first), every one launched before the wait
    assert [b["bucket"] for b in tl] == [2, 1, 0] and [b["bytes"] for b in tl] == [1200, 1600, 1200]
    assert all(b["window_ms"] >= 0 and b["exposed_ms"] >= 0 for b in tl) and tl[0]["launched_at_ms"] == 0


def test_bench_launcher_with_eight_ranks():
    """configs[3]'s world size through the self-launcher (stub step on CPU / gloo): eight ranks rendezvous, the bucketed
    exchange adds up over all of them, ONE line comes back."""
    rc, lines, err = _bench(gpus=8)
    assert rc == 0, err
    assert len(lines) == 1 and lines[0]["n_gpus"] == 8
    assert lines[0]["dp"]["ranks_seen"] == 8 and lines[0]["dp"]["allreduce_checksum_ok"] is True
    assert [b["bucket"] for b in lines[0]["dp"]["buckets_timeline"]] == [2, 1, 0]


def test_bench_fails_when_a_rank_dies():
    rc, _, err = _bench(env={"PM_BENCH_FAIL_RANK": "1"})
    assert rc != 0 and "rank 1 exited" in err


def test_device_loader_shards_are_disjoint_and_in_lock_step():
    """World-2 `DeviceLoader` index plan (no GPU needed for the plan itself): same permutation on both ranks, disjoint
    shards, equal batch counts, every sample seen once per epoch (one wrapped duplicate when the size is odd)."""
    import numpy as np
    from polyphemus_amd.data import DeviceLoader

    class DS:
        n_bars = 2

        def __len__(self):
            return 11

    plans = []
    for rank in range(2):
        ld = DeviceLoader.__new__(DeviceLoader)                   # the index plan only: no pinned buffers, no stream
        ld.dataset, ld.batch_size, ld.shuffle, ld.seed, ld.drop_last, ld.epoch = DS(), 2, True, 5, False, 3
        ld.rank, ld.world = rank, 2
        plans.append(ld._index_batches())
        assert len(plans[-1]) == len(ld) == 3
    a, b = (np.concatenate(p) for p in plans)
    assert len(a) == len(b) == 6
    assert len(set(a) & set(b)) <= 1 and set(a) | set(b) == set(range(11))
    order = np.random.default_rng(5 + 3).permutation(11)
    assert list(a) == list(np.concatenate([order, order[:1]])[0::2])


@pytest.mark.parametrize("n,world,bs", [(2051, 8, 256), (5, 8, 2), (8, 8, 1), (17, 8, 4)])
def test_device_loader_world8_shards(n, world, bs):
    """World-8 index plan (BASELINE configs[3]: 2048 samples over 8 ranks), including a dataset SMALLER than the world
    (the wrap-around used to pad by at most one copy of the permutation, leaving ranks with unequal batch
    counts and the gradient all-reduce waiting for a rank that had no batch): every rank the same number of samples
    and batches, shards disjoint up to the wrapped tail, every sample covered."""
    import numpy as np
    from polyphemus_amd.data import DeviceLoader

    class DS:
        n_bars = 2

        def __len__(self):
            return n

    plans = []
    for rank in range(world):
        ld = DeviceLoader.__new__(DeviceLoader)
        ld.dataset, ld.batch_size, ld.shuffle, ld.seed, ld.drop_last, ld.epoch = DS(), bs, True, 11, False, 0
        ld.rank, ld.world = rank, world
        plans.append(ld._index_batches())
        assert len(plans[-1]) == len(ld)
    counts = [sum(len(b) for b in p) for p in plans]
    assert len(set(counts)) == 1 and len({len(p) for p in plans}) == 1            # lock step
    per_rank = -(-n // world)
    assert counts[0] == per_rank
    seen = np.concatenate([np.concatenate(p) for p in plans])
    assert set(seen.tolist()) == set(range(n))
    assert len(seen) - n == per_rank * world - n                                   # only the wrapped tail repeats


@pytest.mark.parametrize("d,nb", [(512, 16), (256, 2)])
def test_gradient_buckets_cover_the_flat_buffer_exactly(d, nb):
    """The three exchange buckets of the data-parallel step (trainer.bucket_boundaries; BASELINE configs[3] / configs[4] at
    d = 512 with 16 bars: 56.9 M parameters, 228 MB) partition the flat gradient buffer: contiguous, disjoint, complete,
    16-byte-aligned starts (RCCL's vector loads), and every parameter's gradient lies in the bucket whose backward call makes
    it final — host-side arithmetic, no GPU."""
    import torch
    from polyphemus_amd.model import VAE
    from polyphemus_amd.parallel import GradBuckets
    from polyphemus_amd.trainer import bucket_boundaries
    vae = VAE(dropout=0, batch_norm=True, gnn_n_layers=8, d=d, n_bars=nb, resolution=8, device=torch.device("cpu"))
    flat = torch.zeros(vae.flat_params.numel())
    bounds = bucket_boundaries(vae)
    gb = GradBuckets(flat, bounds)
    assert len(gb.views) == 3 and sum(v.numel() for v in gb.views) == flat.numel()
    edges = [0] + list(bounds) + [flat.numel()]
    assert edges == sorted(edges) and len(set(edges)) == 4
    for k, v in enumerate(gb.views):
        assert v.data_ptr() == flat.data_ptr() + 4 * edges[k] and edges[k] % 4 == 0      # contiguous; 16-byte aligned start
        v.fill_(float(k + 1))
    assert float(flat.min()) == 1.0 and int((flat == 0).sum()) == 0                      # nothing left uncovered
    P = dict(vae.named_parameters())
    for n in vae._param_names:
        lo, hi = vae._offsets[n], vae._offsets[n] + P[n].numel()
        k = 2 if n.startswith("decoder.") else (1 if lo >= bounds[0] else 0)
        assert edges[k] <= lo and hi <= edges[k + 1], n                                   # no parameter straddles a boundary
        if n.startswith(("encoder.c_encoder.graph_encoder.", "encoder.c_encoder.graph_attention.", "encoder.c_encoder.bars_encoder.",
                         "encoder.linear_", "encoder.bn_linear_merge.")):
            assert k == 1, n
    if (d, nb) == (512, 16):
        assert flat.numel() * 4 > 225e6                                                   # 228 MB exchanged per step


def test_bench_preflight_names_what_keeps_the_communicator_from_forming(monkeypatch, capsys):
    """`bench.py --gpus N`: before the process group exists every rank checks what RCCL depends on and exits legibly (exit code
    4, the facts on stderr) instead of hanging in the first collective: fewer visible devices than local ranks,
    HSA_ENABLE_IPC_MODE_LEGACY not 0, no rendezvous address."""
    import json
    import bench
    for k, v in (("WORLD_SIZE", "2"), ("RANK", "1"), ("LOCAL_RANK", "1"), ("LOCAL_WORLD_SIZE", "2"), ("MASTER_ADDR", "127.0.0.1"),
                 ("MASTER_PORT", "29999")):
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    facts = bench.preflight("gloo")                               # gloo: nothing to object to on a CPU box
    assert facts["world_size"] == 2 and facts["rank"] == 1 and facts["device_count"] == 0
    with pytest.raises(SystemExit) as e:                          # nccl with no visible GPU: one rank per device is impossible
        bench.preflight("nccl")
    assert e.value.code == 4
    err = capsys.readouterr().err
    assert "rank 1" in err and "2 ranks on this node but 0 visible GPU" in err and '"device_count": 0' in err
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "1")
    monkeypatch.delenv("MASTER_ADDR")
    with pytest.raises(SystemExit):
        bench.preflight("nccl")
    err = capsys.readouterr().err
    assert "HSA_ENABLE_IPC_MODE_LEGACY is not 0" in err and "MASTER_ADDR is not set" in err
    json.loads(err[err.index("{"):err.rindex("}") + 1])           # the facts are machine-readable
