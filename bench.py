#!/usr/bin/env python3
"""bench.py — bar-graphs/sec of the graph-VAE training step on MI355X.

One "step" = plan build + VAE forward + loss + backward + (DP: gradient all-reduce) + fused Adam
on one synthetic LMD2 batch (BASELINE.json configs[1]: 2 bars, 4 tracks x 32 timesteps,
batch 256 per GPU, d_hidden 256, 8 GNN layers; weak scaling: every rank gets its own 256 samples,
configs[3] at 8 GPUs).  Inputs are resident in HBM before the timed region; host-side graph
construction stays outside it, as in the reference (data.py runs in DataLoader workers).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0) with the driver's fields plus `roofline` (dominant kernel, timed live
with HIP events on the launch stream), `roofline_segreduce` (the HBM-bound aggregation kernel) and
`cpu_baseline` (the CPU oracle — a port of the reference op sequence — timed on the host cores,
rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md, chip-level parameters (dense fp32 matrix)
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 / fp16 matrix peak; an fp32 product costs SIX bf16 MFMA products on the exact
                                   # three-term bf16 split, THREE fp16 MFMA products in the fp16 pair format (PmH2, csrc/common.h)
# Memory-side bytes per launch (`roofline.traffic`): bench.py cannot read PMC counters itself, so they come from the
# committed result of the separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over this same command
# (tools/pmc_traffic.py writes profiles/pmc_traffic.json: per kernel class the corrected bytes, the workload and a digest
# of the kernel sources they were measured on).  A stale digest or another workload reports null, never an old number.
PMC_TRAFFIC_JSON = os.path.join(ROOT, "profiles", "pmc_traffic.json")


def kernel_source_digest():
    import hashlib
    h = hashlib.sha256()
    for f in ("gemm.hip", "segreduce.hip", "bar.hip", "gcl.hip", "wide.hip", "gcl_tiles.h", "tile_order.h", "common.h"):
        h.update(open(os.path.join(ROOT, "polyphemus_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_class, workload):
    """profiles/pmc_traffic_<workload>.json if there is one, else profiles/pmc_traffic.json (the bench's default workload)"""
    for path in (PMC_TRAFFIC_JSON.replace(".json", f"_{workload}.json"), PMC_TRAFFIC_JSON):
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        if rec.get("source_digest") == kernel_source_digest() and rec.get("workload") == workload:
            return rec.get("bytes_per_launch", {}).get(kernel_class)
    return None


PEAK_HBM_GBS = 8000.0              # HBM3E spec; ~6.3 TB/s achievable


def flops_per_bar(n_nodes, n_bars_total, d, L):
    """Algorithmic flops per bar-graph, forward+backward:
    3 * mean_nodes_per_bar * [(60 + 28 L) d^2 + 6900 d]."""
    return 3.0 * (n_nodes / n_bars_total) * ((60 + 28 * L) * d * d + 6900 * d)


ROW_TILES = 0          # 64-row tiles of the batch's four track groups (set by executed_block_fraction)


def executed_block_fraction(trainer, n_nodes, n_edges, G):
    """Fraction of the compact GCL contraction [N, 4d] x [4d, d] the kernels of gcl.hip / wide.hip EXECUTE: a 64-row
    tile of a track group skips the onset / next block when none of its rows receives such an edge (row classes,
    plan.trk_cnt[8 + 5 * group ..]: class boundaries of the group's node list).  One host read of 28 integers, outside
    the timed region."""
    from polyphemus_amd._lib import PLAN_FIELDS, plan_layout
    off = plan_layout(n_nodes, n_edges, G)
    j = PLAN_FIELDS.index("trk_cnt")
    tc = trainer._plan_buf[off[j]:off[j] + 28].tolist()
    live = total = 0
    global ROW_TILES
    ROW_TILES = sum((tc[g] + 63) // 64 for g in range(4))
    for g in range(4):
        cnt, cb = tc[g], tc[8 + 5 * g: 13 + 5 * g]
        for m0 in range(0, cnt, 64):
            on = m0 < cb[3] and m0 + 64 > cb[1]
            nx = m0 < cb[4] and m0 + 64 > cb[2]
            live += 2 + int(on) + int(nx)
            total += 4
    return live / total if total else 1.0


def standalone_segreduce_fwd(batch, d, p=0.1, reps=24):
    """"when the layer is fused the aggregates never reach HBM; keep the standalone kernel for this
    measurement" — in-process series of `pm_segreduce_fwd_planes` launches (compact [N, 4d] aggregate written as three
    bf16 planes, the form the unfused step uses) on the bench batch, HIP events on the launch stream, untimed part of
    the run.  Algorithmic bytes: x read 4dN + planes written 6 * 4dN + 12 E.

    Two series: (a) ROTATING operands — as many (x, planes) sets as it takes to exceed 768 MB, three times the 256 MB
    Infinity Cache, visited round-robin, so that no launch finds its input or the lines it overwrites in the memory-side
    cache: the HBM figure (`avg_launch_us`, `frac`); (b) the SAME 117 MB set every launch (round 3's measurement): the
    working set fits the Infinity Cache and the rate is partly a cache figure (`same_buffers`)."""
    from polyphemus_amd import ops
    from polyphemus_amd._lib import call, ptr, stream
    plan = ops.plan_build(batch.edge_index, batch.edge_type, batch.edge_dist, batch.bars, batch.batch, batch.is_drum,
                          batch.tokens, batch.n_bars, batch.s_tensor.shape[0])
    N, E = batch.num_nodes, batch.edge_index.shape[1]
    dev = batch.edge_index.device
    nbytes = 4.0 * d * N + 6.0 * 4 * d * N + 12.0 * E
    nset = max(2, int(768e6 // nbytes) + 1)
    xs = [torch.randn(N, d, device=dev) for _ in range(nset)]
    T = ops.edge_table(torch.randn(d, 32, device=dev) * 0.5, torch.randn(d, device=dev) * 0.1)
    Ps = [torch.empty(3, N * 4 * d, dtype=torch.int16, device=dev) for _ in range(nset)]

    def run(i):
        call("pm_segreduce_fwd_planes", ptr(xs[i]), ptr(T), ptr(plan.buf), N, plan.E, plan.G, d, p, 5, 2, 1, ptr(Ps[i]), N * 4 * d, stream())

    def series(rotate):
        for i in range(nset if rotate else 3):
            run(i if rotate else 0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for i in range(reps):
            run(i % nset if rotate else 0)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3

    us, us_same = series(True), series(False)
    return {"kernel": "k_segreduce_fwd (stand-alone, compact aggregate as three bf16 planes)", "avg_launch_us": round(us, 2),
            "algorithmic_bytes_per_launch": nbytes, "achieved": round(nbytes / us / 1e3, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(nbytes / us / 1e3 / PEAK_HBM_GBS, 4),
            "from": f"{reps} back-to-back launches over {nset} rotating operand sets ({nset * nbytes / 1e6:.0f} MB: beyond the "
                    "256 MB Infinity Cache) after the warm-up, events on torch's current stream (the launch stream)",
            "same_buffers": {"avg_launch_us": round(us_same, 2), "achieved": round(nbytes / us_same / 1e3, 1),
                             "frac": round(nbytes / us_same / 1e3 / PEAK_HBM_GBS, 4),
                             "note": "the same 117 MB operand set every launch: fits the Infinity Cache, not an HBM figure"}}


def reference_format(batch):
    """The batch as the reference's DataLoader hands it to `model(graph)` (data.py:179-182,235-268): edge_attrs [E, 33] (type
    as a float in column 0, one-hot distance), c_tensor one-hot [N, 16, 230] (240 MB at configs[1]); no token ids."""
    class Graph:
        pass
    g = Graph()
    E, N = batch.edge_index.shape[1], batch.num_nodes
    ea = torch.zeros(E, 33, device=batch.edge_index.device)
    ea[:, 0] = batch.edge_type.float()
    ea[torch.arange(E, device=ea.device), 1 + batch.edge_dist.long()] = 1.0
    tok = batch.tokens.long()
    c = torch.zeros(N, 16, 230, device=ea.device)
    c.scatter_(2, tok[..., 0:1], 1.0)
    c.scatter_(2, 131 + tok[..., 1:2], 1.0)
    g.edge_index, g.edge_attrs, g.c_tensor, g.s_tensor = batch.edge_index, ea, c, batch.s_tensor
    g.is_drum, g.bars, g.batch, g.num_nodes = batch.is_drum, batch.bars, batch.batch, N
    return g


def reference_losses(s_tensor, s_logits, c_tensor, c_logits, mu, log_var, beta=0.0):
    """`PolyphemusTrainer._losses` (training.py:298-347) in the caller's torch ops, quirks included (the structure BCE on the
    target itself, :307; beta = 0, :116) — and its seven `.item()` host reads."""
    F = torch.nn.functional
    c_tensor = c_tensor[..., 1:, :]
    c_logits = c_logits.reshape(-1, c_logits.size(-1))
    c_tensor = c_tensor.reshape(-1, c_tensor.size(-1))
    s_in = s_tensor.reshape(-1, *s_logits.shape[2:])
    s_loss = F.binary_cross_entropy_with_logits(s_in.reshape(-1), s_tensor.reshape(-1).float(), reduction="none").mean()
    pitch_loss = F.cross_entropy(c_logits[:, :131], c_tensor[:, :131].argmax(dim=1), ignore_index=130)
    dur_loss = F.cross_entropy(c_logits[:, 131:], c_tensor[:, 131:].argmax(dim=1), ignore_index=98)
    kld = (-0.5 * torch.sum(1 + log_var - mu.pow(2) - log_var.exp(), dim=1)).mean()
    rec = pitch_loss + dur_loss + s_loss
    tot = rec + beta * kld
    return tot, {"tot": tot.item(), "pitch": pitch_loss.item(), "dur": dur_loss.item(), "structure": s_loss.item(),
                 "reconstruction": rec.item(), "kld": kld.item(), "beta*kld": beta * kld.item()}


def reference_loop_workload(name, active_slots_only, batch_size=256, d=256, n_bars=2, layers=8, steps=10, warmup=3, seed=1234):
    """The drop-in boundary as the UNCHANGED reference loop drives it (training.py:137-172, train.py:176-181): reference-format
    inputs, `vae(graph)` under fp16 autocast, the reference's `_losses` in torch, `GradScaler.scale(loss).backward()`,
    `scaler.step(torch.optim.Adam)`, `zero_grad` — `model(graph)` and its autograd node run the C++ step (model._VaeStepFn),
    the loss, the unscale and Adam over 152 tensors are the caller's torch code.  A new batch object every step, as a
    DataLoader gives (the one-hot -> id conversion and its host read are inside the timed region)."""
    from polyphemus_amd.model import VAE
    from polyphemus_amd.synthetic import synthetic_batch
    dev = torch.device("cuda", torch.cuda.current_device())
    cfg = dict(dropout=0, batch_norm=True, gnn_n_layers=layers, d=d, n_bars=n_bars, resolution=8)
    torch.manual_seed(0)
    vae = VAE(**cfg, device=dev).to(dev)
    vae.train()
    vae.active_slots_only = bool(active_slots_only)
    opt = torch.optim.Adam(vae.parameters(), lr=5e-6, betas=(0.9, 0.98), eps=1e-9)     # train.py:181, training.json:11-18
    scaler = torch.amp.GradScaler("cuda")                                             # training.py:123
    g = reference_format(synthetic_batch(batch_size, n_bars, p=0.25, seed=seed).to(dev))
    from polyphemus_amd.native import prepare_inputs
    parts = None
    phases = ("inputs: one-hot -> ids (+ one host read)", "forward: vae(graph)", "loss: _losses in torch (7 .item())",
              "backward: torch loss backward + native backward", "optimizer: GradScaler unscale + torch Adam + zero_grad")
    acc = [0.0] * len(phases)

    def step(timed=False):
        nonlocal parts
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(phases) + 1)] if timed else None
        mark = (lambda i: ev[i].record()) if timed else (lambda i: None)
        g.__dict__.pop("_pm_inputs", None)
        mark(0)
        prepare_inputs(g)                                                             # (vae(g) would do it: split out to time it)
        mark(1)
        with torch.autocast("cuda", dtype=torch.float16):                             # training.py:137
            (s_logits, c_logits), mu, log_var = vae(g)
            mark(2)
            tot, parts = reference_losses(g.s_tensor, s_logits, g.c_tensor, c_logits, mu, log_var)
        mark(3)
        scaler.scale(tot).backward()                                                  # training.py:153
        mark(4)
        scaler.step(opt)
        scaler.update()
        opt.zero_grad()
        mark(5)
        return ev
    for _ in range(warmup):
        step()
    first = dict(parts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    for _ in range(3):                                                                # GPU time per phase (separate, event-timed steps)
        ev = step(timed=True)
        torch.cuda.synchronize()
        for i in range(len(phases)):
            acc[i] += ev[i].elapsed_time(ev[i + 1]) / 3.0
    G = g.s_tensor.shape[0]
    info = vae._native_step().info()
    out = {"workload": name, "bar-graphs/s": round(G / dt, 1), "ms_per_step": round(1e3 * dt, 3), "steps": steps, "warmup": warmup,
           "batch": batch_size, "d": d, "n_bars": n_bars, "batch_seed": seed, "decoder_head_slots": info["n_slots"],
           "loss_after_warmup": round(first["tot"], 5), "loss_last": round(parts["tot"], 5),
           "gpu_ms_by_phase": {k: round(v, 3) for k, v in zip(phases, acc)},
           "whose": "forward and backward run the C++ step behind model(graph) / autograd (plus torch's backward of the loss); inputs "
                    "conversion is this package's; loss and optimizer are the reference loop's own torch code",
           "kernel_path": {k: v for k, v in info.items() if k not in ("N", "E", "G", "B")}}
    del vae, opt, g
    torch.cuda.empty_cache()
    return out


def generation_workload(batch_size=256, d=256, n_bars=2, layers=8, steps=10, warmup=3):
    """The generation path of generate.py:21-37: z ~ N(0, 1) -> `vae.decoder(z, None)` (structure decoder, threshold, device-side
    graph construction, content decoder; eval mode) -> bar-graphs generated per second."""
    from polyphemus_amd.model import VAE
    dev = torch.device("cuda", torch.cuda.current_device())
    cfg = dict(dropout=0, batch_norm=True, gnn_n_layers=layers, d=d, n_bars=n_bars, resolution=8)
    torch.manual_seed(0)
    vae = VAE(**cfg, device=dev).to(dev)
    vae.eval()
    z = torch.randn(batch_size, d, device=dev)
    with torch.no_grad():
        for _ in range(warmup):
            vae.decoder(z, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            s_logits, c_logits = vae.decoder(z, None)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {"workload": "generation: z -> decoder(z, None) -> (s_logits, c_logits), eval mode (generate.py:21-37)",
           "bar-graphs/s": round(batch_size * n_bars / dt, 1), "ms_per_call": round(1e3 * dt, 3), "steps": steps, "warmup": warmup,
           "batch": batch_size, "d": d, "n_bars": n_bars, "nodes_generated": int(c_logits.shape[0])}
    del vae
    torch.cuda.empty_cache()
    return out


def other_workload(name, batch_size, d, n_bars, dense, steps=5, warmup=2, layers=8, seed=1234, p=0.25, max_notes=4):
    """The remaining single-GPU configurations of BASELINE.json through the same step (fresh model and trainer, untimed part
    of the run): whole-step rate only."""
    from polyphemus_amd.model import VAE
    from polyphemus_amd.synthetic import synthetic_batch
    from polyphemus_amd.trainer import HipTrainer
    dev = torch.device("cuda", torch.cuda.current_device())
    cfg = dict(dropout=0, batch_norm=True, gnn_n_layers=layers, d=d, n_bars=n_bars, resolution=8)
    torch.manual_seed(0)
    vae = VAE(**cfg, device=dev).to(dev)
    vae.train()
    tr = HipTrainer(vae, lr=5e-6, betas=(0.9, 0.98), eps=1e-9)
    batch = synthetic_batch(batch_size, n_bars, p=p, seed=seed, dense=dense, max_notes=max_notes).to(dev)
    for _ in range(warmup):
        tr.train_step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    G = batch.s_tensor.shape[0]
    out = {"workload": name, "bar-graphs/s": round(G / dt, 1), "ms_per_step": round(1e3 * dt, 3), "steps": steps, "warmup": warmup,
           "batch": batch_size, "d": d, "n_bars": n_bars, "batch_seed": seed, "nodes": batch.num_nodes,
           "edges": int(batch.edge_index.shape[1]), "kernel_path": {k: v for k, v in tr.step_info().items() if k not in ("N", "E", "G", "B")}}
    del tr, vae, batch
    torch.cuda.empty_cache()
    return out


def host_cores() -> int:
    """Cores this process may actually use: min(affinity mask, cgroup CPU quota).  os.cpu_count() reports
    every core of the host even inside a CPU-limited container, and oversubscribing OpenMP threads
    (256 threads on a 16-CPU quota) slows the oracle by orders of magnitude."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline_subprocess(cfg, timeout_s=300):
    """Run the CPU leg in a child process (never touches the GPU) under a hard timeout."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--d", str(cfg["d"]), "--n-bars",
           str(cfg["n_bars"]), "--layers", str(cfg["gnn_n_layers"])]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(host_cores()))
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        for ln in reversed(r.stdout.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"value": None, "error": (r.stderr or r.stdout)[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "error": f"cpu baseline exceeded {timeout_s}s"}


def cpu_baseline(cfg, seconds_budget=20.0):
    """The CPU oracle (oracle/vae_cpu.py: reference op sequence, fp32, stock torch Adam) on bounded samples of the same
    workload, all host cores: B = 64 samples of the same synthetic distribution for a stable rate
    (`value`), B = 8 (BASELINE configs[0], the reference's own CPU-runnable case) beside it."""
    rate = cpu_baseline_at(cfg, 64, seconds_budget)
    small = cpu_baseline_at(cfg, 8, seconds_budget / 4)
    rate["configs0_B8"] = {"value": small["value"], "unit": small["unit"], "sample": small["sample"]}
    if cfg["d"] != 512:            # ... and the reference's own width (training.json: d = 512), B = 32, beside the d = 512 GPU workload
        wide = cpu_baseline_at(dict(cfg, d=512), 32, seconds_budget / 2)
        rate["training_json_d512_B32"] = {"value": wide["value"], "unit": wide["unit"], "sample": wide["sample"]}
    return rate


def cpu_baseline_at(cfg, B, seconds_budget):
    from oracle import vae_cpu
    from polyphemus_amd.model import VAE
    from polyphemus_amd.synthetic import synthetic_batch
    torch.set_num_threads(host_cores())
    batch = synthetic_batch(B, cfg["n_bars"], p=0.25, seed=1234)
    torch.manual_seed(0)
    ref = VAE(**cfg, device=torch.device("cpu"))
    names = [n for n, _ in ref.named_parameters()]
    P, names = vae_cpu.split_state({k: v.detach().clone() for k, v in ref.state_dict().items()}, names)
    opt = torch.optim.Adam([P[n] for n in names], lr=5e-6, betas=(0.9, 0.98), eps=1e-9)
    _ = batch.c_tensor, batch.edge_attrs                      # reference-format inputs, built outside the timing
    vae_cpu.train_step(batch, P, names, cfg, opt)             # warm-up
    times = []
    t_end = time.time() + seconds_budget
    while len(times) < 5 and (time.time() < t_end or len(times) < 2):
        t0 = time.time()
        vae_cpu.train_step(batch, P, names, cfg, opt)
        times.append(time.time() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": B * cfg["n_bars"] / med, "unit": "bar-graphs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/vae_cpu.py train step (fwd+loss+bwd+Adam, message dropout on), B={B} 2-bar samples "
                      f"({batch.num_nodes} nodes), d={cfg['d']}, L={cfg['gnn_n_layers']}, median of {len(times)} steps "
                      f"= {med * 1e3:.0f} ms"}


def launch_ranks(argv, n):
    """`python bench.py --gpus N` without a launcher: start N rank processes of this script (one per GPU, fresh
    interpreters, rendezvous on 127.0.0.1 at a free port) BEFORE this process touches the GPU, forward rank 0's output,
    and fail if any rank fails (the survivors are killed, never left waiting in a collective)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PM_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        live = set(range(n))
        while live:
            for r in list(live):
                code = procs[r].poll()
                if code is not None:
                    live.discard(r)
                    if code != 0:
                        rc = rc or code
                        sys.stderr.write(f"[bench] rank {r} exited with code {code}; stopping the other ranks\n")
                        for q in live:
                            procs[q].terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return rc


def preflight(backend):
    """Multi-GPU runs: fail FAST and LEGIBLY when the communicator cannot form, instead of hanging in the first collective.
    Before the process group exists: the facts RCCL depends on (one visible device per local rank — counting devices does not
    initialise the GPU on this image —, dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0 must be in the environment BEFORE the runtime
    starts, the rendezvous address).  Returns the facts; `preflight_collective` then proves the communicator with one small
    all-reduce under a watchdog and reports `ranks_seen`."""
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    lws = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    facts = {"rank": rank, "world_size": world, "local_rank": local, "local_world_size": lws, "backend": backend,
             "device_count": torch.cuda.device_count(), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
             "MASTER_ADDR": os.environ.get("MASTER_ADDR"), "MASTER_PORT": os.environ.get("MASTER_PORT"),
             "HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES"), "ROCR_VISIBLE_DEVICES": os.environ.get("ROCR_VISIBLE_DEVICES")}
    problems = []
    if backend == "nccl":
        if facts["device_count"] < lws:
            problems.append(f"{lws} ranks on this node but {facts['device_count']} visible GPU(s): RCCL refuses two ranks on one device")
        if facts["HSA_ENABLE_IPC_MODE_LEGACY"] != "0":
            problems.append("HSA_ENABLE_IPC_MODE_LEGACY is not 0: this host supports dmabuf IPC only, RCCL would fail with "
                            "`hipIpcGetMemHandle: invalid argument` (export it before the ranks start)")
    if world > 1 and not facts["MASTER_ADDR"]:
        problems.append("MASTER_ADDR is not set (launch with torch.distributed.run --master-addr 127.0.0.1)")
    if problems:
        print(f"[bench preflight] rank {rank}: the {world}-rank communicator cannot form: " + "; ".join(problems) +
              f" | facts: {json.dumps(facts)}", file=sys.stderr, flush=True)
        raise SystemExit(4)
    return facts


def preflight_collective(facts, dev, timeout_s=90.0):
    """One 4-byte all-reduce under a watchdog: every rank must arrive; prints `ranks_seen` (rank 0) or the reason and exits."""
    import threading
    import torch.distributed as dist
    done = threading.Event()

    def watchdog():
        if not done.wait(timeout_s):
            print(f"[bench preflight] rank {facts['rank']}: the first all-reduce did not complete within {timeout_s:.0f} s — a rank is "
                  f"missing or RCCL could not connect the ranks | facts: {json.dumps(facts)}", file=sys.stderr, flush=True)
            os._exit(5)
    threading.Thread(target=watchdog, daemon=True).start()
    one = torch.ones(1, device=dev)
    dist.all_reduce(one)
    seen = int(one.item())
    done.set()
    if seen != facts["world_size"]:
        print(f"[bench preflight] rank {facts['rank']}: ranks_seen {seen} != WORLD_SIZE {facts['world_size']} | facts: {json.dumps(facts)}",
              file=sys.stderr, flush=True)
        raise SystemExit(6)
    if facts["rank"] == 0:
        print(f"[bench preflight] ranks_seen={seen} backend={dist.get_backend()} devices={facts['device_count']} "
              f"HSA_ENABLE_IPC_MODE_LEGACY={facts['HSA_ENABLE_IPC_MODE_LEGACY']}", file=sys.stderr, flush=True)
    return seen


def stub_main(args):
    """`--stub-step` (tests only, no GPU, no HIP): the launcher, rendezvous, barrier / max-over-ranks timing and JSON
    contract of the real bench with the training step replaced by a bucketed gloo all-reduce of a small CPU buffer."""
    import torch.distributed as dist
    from polyphemus_amd import parallel
    rank, local, world = parallel.init_from_env("gloo")
    flat = torch.full((1000,), float(rank + 1))
    gb = parallel.GradBuckets(flat, [300, 700])

    def step():
        flat.fill_(float(rank + 1))
        for i in (2, 1, 0):
            gb.launch(i)
        return gb.wait()

    for _ in range(args.warmup):
        step()
    gb.trace = True
    step()
    timeline = gb.timeline()
    gb.trace = False
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        scale = step()
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = bool((flat == world * (world + 1) / 2).all()) and scale == 1.0 / world
    if os.environ.get("PM_BENCH_FAIL_RANK") == str(rank):
        raise SystemExit(3)                                # (test hook: a dying rank must fail the whole launch)
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": args.steps / float(t), "unit": "steps/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * float(t) / args.steps,
                          "dp": {"ranks_seen": world, "backend": "gloo", "allreduce_checksum_ok": ok,
                                 "buckets_timeline": timeline}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="samples per GPU")
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--n-bars", type=int, default=2)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--dense", action="store_true", help="BASELINE configs[4] dense-graph stress")
    ap.add_argument("--seed", type=int, default=1234, help="seed of the synthetic batch (rank r uses seed + r)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the `other_workloads` block of the default run")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--stub-step", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torch.distributed.run: be the launcher (nothing above has touched the GPU)
        raise SystemExit(launch_ranks(sys.argv[1:], args.gpus))
    if args.stub_step:
        return stub_main(args)
    if args.cpu_baseline_only:
        cfg = dict(dropout=0, batch_norm=True, gnn_n_layers=args.layers, d=args.d, n_bars=args.n_bars, resolution=8)
        print(json.dumps(cpu_baseline(cfg)))
        return

    from polyphemus_amd import ops, parallel
    from polyphemus_amd.model import VAE
    from polyphemus_amd.synthetic import synthetic_batch
    from polyphemus_amd.trainer import HipTrainer

    backend = os.environ.get("PM_DIST_BACKEND", "nccl")
    facts = preflight(backend) if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None
    rank, local, world = parallel.init_from_env(backend)
    if world != max(args.gpus, 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    dev = torch.device("cuda", local % torch.cuda.device_count())     # (PM_DIST_BACKEND=gloo: ranks may share a GPU)
    torch.cuda.set_device(dev)
    if facts is not None:
        preflight_collective(facts, dev)
    cfg = dict(dropout=0, batch_norm=True, gnn_n_layers=args.layers, d=args.d, n_bars=args.n_bars, resolution=8)

    torch.manual_seed(0)
    vae = VAE(**cfg, device=dev).to(dev)
    vae.train()
    tj = dict(peak_lr=1e-4, final_lr_scale=0.01, warmup_steps=8000, decay_steps=800000)      # training.json:19-24
    trainer = HipTrainer(vae, lr=5e-6, betas=(0.9, 0.98), eps=1e-9, lr_scheduler=tj)        # training.json:11-18
    batch = synthetic_batch(args.batch, args.n_bars, p=0.25, seed=args.seed + rank, dense=args.dense).to(dev)
    n_nodes, n_edges, G = batch.num_nodes, batch.edge_index.shape[1], batch.s_tensor.shape[0]

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    import ctypes
    from polyphemus_amd._lib import lib
    L = lib()
    from polyphemus_amd._lib import PROF_NCLASS as NCLS
    tiles = ("64x64x16", "128x128x16", "64x64x32", "128x128x32", "x6:128x128x16", "x6:128x64x16", "x6:64x64x32",
             "x6:128x128x32", "planes:64x64x32", "planesB:64x128x32", "planes:128x128x32")
    lay = ("NN", "NT", "TN")
    names = [f"gemm_{lay[c % 3]}_{tiles[c // 3]}" for c in range(33)] + ["segreduce_fwd", "segreduce_bwd", "gcl_fwd", "gcl_dagg", "gcl_dw", "gemm_NN_rows_w", "gemm_TN_rows_tn"]
    names[9 * 3] = "gemm_NN_planesB:64x128x64"            # (the B-direct forward product takes k-tiles of 64)

    def prof_collect():
        ms_a, work_a, cnt_a = (ctypes.c_double * NCLS)(), (ctypes.c_double * NCLS)(), (ctypes.c_int64 * NCLS)()
        L.pm_prof_end(ctypes.cast(ms_a, ctypes.c_void_p), ctypes.cast(work_a, ctypes.c_void_p), ctypes.cast(cnt_a, ctypes.c_void_p))
        return {names[c]: dict(launches=int(cnt_a[c]), total_ms=ms_a[c], avg_us=1e3 * ms_a[c] / cnt_a[c], work=work_a[c])
                for c in range(NCLS) if cnt_a[c] > 0}

    # Untimed steps: W warm-up steps, the last SURVEY of them with HIP events around EVERY GEMM / segment-reduce launch
    # (in-library, on the launch stream): the per-class table and the choice of the dominant class.  An event pair costs
    # ~8 us of GPU idle time (all ~115 bracketed launches of a step: 12 % of the step), so the timed region brackets only
    # the dominant GEMM class and the segment-reduce forward, every EVENT_STRIDE-th launch of each.
    # ---- data-parallel evidence (world > 1): who is here, does the exchange add up, what does it cost
    dp = None
    if world > 1:
        import torch.distributed as dist
        probe = torch.arange(1024, dtype=torch.float32, device=dev) * float(rank + 1)
        dist.all_reduce(probe)
        ok = bool(torch.equal(probe, torch.arange(1024, dtype=torch.float32, device=dev) * (world * (world + 1) / 2)))
        for _ in range(3):
            trainer.train_step(batch)
        reps = 5
        sync()
        t0 = time.perf_counter()
        for _ in range(reps):                                   # the exchange alone: the three buckets back to back
            for i in (2, 1, 0):
                trainer.buckets.launch(i)
            trainer.buckets.wait()
        sync()
        t_alone = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            trainer.train_step(batch)
        sync()
        t_dp = (time.perf_counter() - t0) / reps
        trainer.buckets.trace = True                            # one more step with every bucket launch / completion stamped
        trainer.train_step(batch)
        sync()
        timeline = trainer.buckets.timeline()
        trainer.buckets.trace = False
        trainer.buckets.disabled = True                         # same step without the exchange (ranks drift apart ...)
        t0 = time.perf_counter()
        for _ in range(reps):
            trainer.train_step(batch)
        sync()
        t_nodp = (time.perf_counter() - t0) / reps
        trainer.buckets.disabled = False
        parallel.broadcast_([vae.flat_params, vae.flat_buffers, trainer.exp_avg, trainer.exp_avg_sq], 0)   # ... re-joined here
        tt = torch.tensor([t_alone, t_dp, t_nodp], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_alone, t_dp, t_nodp = tt.tolist()
        exposed = max(t_dp - t_nodp, 0.0)
        dp = {"ranks_seen": dist.get_world_size(), "backend": dist.get_backend(), "allreduce_checksum_ok": ok,
              "gradient_bytes": int(trainer.grads.numel() * 4), "buckets": [int(v.numel() * 4) for v in trainer.buckets.views],
              "allreduce_alone_ms": round(1e3 * t_alone, 3), "step_ms_with_exchange": round(1e3 * t_dp, 3),
              "step_ms_without_exchange": round(1e3 * t_nodp, 3), "allreduce_exposed_ms": round(1e3 * exposed, 3),
              "overlapped_fraction": round(1.0 - min(exposed / t_alone, 1.0), 3) if t_alone > 0 else None,
              # rank 0's compute-stream clock, one traced step: per bucket the compute that ran between its launch and the
              # point where Adam needs the gradients (window) and how long the step then stood waiting for it (exposed)
              "buckets_timeline": timeline,
              "note": f"max over ranks, {reps} repetitions each, measured before the timed region"}

    SURVEY = 2
    EVENT_STRIDE = int(os.environ.get("PM_BENCH_EVENT_STRIDE", "5" if args.steps >= 5 else "1"))
    losses_first = None
    for i in range(max(args.warmup - SURVEY, 0)):
        o = trainer.train_step(batch)
        if i == 0 and dp is None:
            losses_first = trainer.losses_dict(o)                  # step 1 of the run: default init, reproducible to 1e-9
    L.pm_prof_configure(-1, 1)
    L.pm_prof_begin(SURVEY * 200)
    for _ in range(SURVEY):
        trainer.train_step(batch)
    sync()
    survey = prof_collect()
    step_info = trainer.step_info()
    exec_frac = executed_block_fraction(trainer, n_nodes, n_edges, G) if step_info["compact"] else 1.0
    seg_alone = standalone_segreduce_fwd(batch, args.d) if rank == 0 else None
    mfma_classes = [k for k in survey if k.startswith(("gemm", "gcl"))]       # every kernel class that runs on the matrix cores
    dom = max(mfma_classes, key=lambda k: survey[k]["total_ms"])
    # the aggregation kernel of the forward: the fused layer kernel (gcl.hip) where it runs, else the segment-reduce
    # the aggregation kernel whose HBM view the line carries: the fused GCL forward on sparse graphs; on the dense route the
    # aggregation is a kernel of its own again (csrc/bar.hip, profiler class segreduce_fwd) — and vector-ALU bound, see below
    seg_key = "segreduce_fwd" if (args.dense and "segreduce_fwd" in survey) else ("gcl_fwd" if "gcl_fwd" in survey else "segreduce_fwd")
    L.pm_prof_configure((1 << names.index(dom)) | (1 << names.index(seg_key)), EVENT_STRIDE)
    L.pm_prof_begin(args.steps * 64 + 64)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = trainer.train_step(batch)
    sync()
    elapsed = time.perf_counter() - t0
    gst = prof_collect()
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    tot_nodes = torch.tensor([float(n_nodes), float(G)], dtype=torch.float64, device=dev)
    tiles_all = torch.zeros(world, dtype=torch.float64, device=dev)
    tiles_all[rank] = float(ROW_TILES)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(tot_nodes)
        torch.distributed.all_reduce(tiles_all)
    elapsed = float(t.item())
    losses = trainer.losses_dict(out)
    if world > 1:                                                # every rank must hold the same parameters after the run
        ck = vae.flat_params.double().sum().reshape(1)
        lo, hi = ck.clone(), ck.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        dp["params_in_sync_after_run"] = bool(lo.item() == hi.item())

    if rank == 0:
        gemm_keys = mfma_classes
        ds = gst[dom]                                          # timed region, sampled launches of the dominant class
        tf = ds["work"] / (ds["total_ms"] * 1e-3) / 1e12
        gemm_ms = sum(survey[k]["total_ms"] for k in gemm_keys)
        gemm_tf = sum(survey[k]["work"] for k in gemm_keys) / (gemm_ms * 1e-3) / 1e12
        split = dom.startswith("gcl") or dom[8:].startswith(("planes", "x6"))   # fp32 product = 6 bf16 MFMA products (fp32 accumulate)
        h2 = dom.startswith("gcl") and step_info.get("h2", 0) == 3              # ... or 3 fp16 products (fp16 pair format: the sparse-graph route of d = 128 / 256 / 512)
        nprod = 3.0 if h2 else 6.0
        peak = PEAK_BF16_MFMA_TFLOPS / nprod if split else PEAK_FP32_MFMA_TFLOPS
        insn = ("v_mfma_f32_32x32x16_f16, 3 products per fp32 product (fp16 pair operands)" if h2 else
                "v_mfma_f32_32x32x16_bf16, 6 products per fp32 product") if split else "v_mfma_f32_32x32x2_f32"
        workload_key = f"B{args.batch}_d{args.d}_nb{args.n_bars}_L{args.layers}" + ("_dense" if args.dense else "")
        sampling = (f"HIP events around every {EVENT_STRIDE}-th launch of this kernel inside the timed region "
                    f"({ds['launches']} launches sampled)")
        if args.d == 512:
            gcl_names = {"gcl_fwd": ("k_wide<V_FWDP>: GCL forward product from the A' planes of the bar-resident aggregation (bar.hip -> wide.hip)" if args.dense else
                                     "k_wide<V_FWD>: GCL forward, aggregate built in LDS + weight product in one kernel (wide.hip)"),
                         "gcl_dagg": "k_wide<V_DAGG>: GCL input gradient, dh streamed through the LDS ring (wide.hip)",
                         "gcl_dw": "k_gcl_dw<512>: GCL weight gradient, 128x128 tiles"}
        else:
            gcl_names = {"gcl_fwd": "k_gcl_fwd: GCL forward, aggregate built in LDS + weight product in one kernel",
                         "gcl_dagg": "k_gcl_dagg: GCL input gradient, A-stationary", "gcl_dw": "k_gcl_dw: GCL weight gradient, 128x128 tiles"}
        kname = f"{gcl_names[dom]} ({insn})" if dom in gcl_names else f"k_gemm<{dom[8:]},{dom[5:7]}> ({insn})"
        roof = {"bound": "mfma", "kernel": kname,
                "achieved": round(tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(tf / peak, 4), "traffic": pmc_traffic(dom, workload_key),
                # the same launch against the OTHER roof: memory-side counter bytes per launch / its duration / 8 TB/s — for a
                # kernel that gathers, stores operand planes and multiplies in one launch both fractions are co-limits
                "hbm_counter_frac": (round(pmc_traffic(dom, workload_key) / (ds["avg_us"] * 1e-6) / 1e9 / PEAK_HBM_GBS, 4)
                                     if pmc_traffic(dom, workload_key) else None),
                "peak_note": ((f"dense fp16 MFMA peak / 3 (fp32-equivalent flops; the same kernel on the six-product bf16 split had a "
                               f"roof of {round(PEAK_BF16_MFMA_TFLOPS / 6.0, 1)}: {round(tf / (PEAK_BF16_MFMA_TFLOPS / 6.0), 4)} of that)" if h2 else
                               "dense bf16 MFMA peak / 6 (fp32-equivalent flops)") if split else "dense fp32 MFMA peak")
                             + f"; {round(tf / PEAK_FP32_MFMA_TFLOPS, 3)} of the 157.3 TFLOP/s fp32 MFMA peak",
                "launches_per_step": survey[dom]["launches"] / SURVEY, "avg_launch_us": round(ds["avg_us"], 2),
                "algorithmic_gflop_per_launch": round(ds["work"] / ds["launches"] / 1e9, 3), "sampling": sampling,
                # `achieved` prices the full [N, 4d] x [4d, d] contraction; the GCL kernels skip the onset / next blocks no row
                # of a 64-row tile receives (row classes): what they EXECUTE, and the rate on that
                "executed": ({"fraction_of_algorithmic": round(exec_frac, 4),
                              "gflop_per_launch": round(exec_frac * ds["work"] / ds["launches"] / 1e9, 3),
                              "TFLOP/s": round(exec_frac * tf, 2), "frac": round(exec_frac * tf / peak, 4)}
                             if dom.startswith("gcl") else None),
                "survey_note": f"all_gemm / other_gemm_classes: every launch bracketed in the last {SURVEY} untimed warm-up steps",
                "all_gemm": {"TFLOP/s": round(gemm_tf, 2), "ms_per_step": round(gemm_ms / SURVEY, 3)},
                "other_gemm_classes": {k: {"TFLOP/s": round(survey[k]["work"] / (survey[k]["total_ms"] * 1e-3) / 1e12, 2),
                                           "avg_us": round(survey[k]["avg_us"], 2),
                                           "launches_per_step": survey[k]["launches"] / SURVEY}
                                       for k in gemm_keys if k != dom}}
        ss = gst[seg_key]
        if seg_key == "gcl_fwd":        # algorithmic HBM bytes of the fused layer kernel (its profiler work is the product's flops)
            # x read + h written + A' planes written (three bf16 / two fp16 planes of [N, 4d]) + edge records + the weight planes
            npl = 2.0 if step_info.get("h2", 0) == 3 else 3.0
            seg_bytes = 8.0 * args.d * n_nodes + 8.0 * npl * args.d * n_nodes + 12.0 * n_edges + 14.0 * npl * args.d * args.d
        else:
            seg_bytes = ss["work"] / ss["launches"]
        gbs = seg_bytes / (ss["avg_us"] * 1e-6) / 1e9
        roof_seg = {"bound": "hbm", "kernel": "k_gcl_fwd (aggregate built in LDS + weight product, one kernel)"
                    if seg_key == "gcl_fwd" else ("k_bar_fwd (dense graphs: the bar's rows in LDS, csrc/bar.hip)" if args.dense else "k_segreduce_fwd"),
                    "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                    "traffic": pmc_traffic(seg_key, workload_key),
                    "launches_per_step": survey[seg_key]["launches"] / SURVEY, "avg_launch_us": round(ss["avg_us"], 2),
                    "algorithmic_bytes_per_launch": seg_bytes,
                    "sampling": f"every {EVENT_STRIDE}-th launch inside the timed region ({ss['launches']} sampled)"}
        if seg_key == "gcl_fwd":
            # the fused kernel is no longer bounded by HBM alone: it also carries the layer's forward product
            # (2 * N * 4d * d flops as six bf16 MFMA products each) and the aggregation's vector-ALU work
            fl = 2.0 * n_nodes * 4.0 * args.d * args.d
            ftf = fl / (ss["avg_us"] * 1e-6) / 1e12
            roof_seg["note"] = ("aggregate built in LDS and contracted in the same kernel (csrc/gcl.hip): the [N,4d] planes "
                                "are written for the backward but never read back in the forward; algorithmic bytes = x read "
                                "+ h written + A' planes written + edges + weight planes")
            np_ = 3.0 if step_info.get("h2", 0) == 3 else 6.0
            roof_seg["mfma"] = {"achieved": round(ftf, 2), "peak": round(PEAK_BF16_MFMA_TFLOPS / np_, 1), "unit": "TFLOP/s",
                                "frac": round(ftf / (PEAK_BF16_MFMA_TFLOPS / np_), 4),
                                "algorithmic_gflop_per_launch": round(fl / 1e9, 3),
                                "peak_note": f"dense 16-bit MFMA peak / {int(np_)} (fp32-equivalent flops)"}
        sb = survey.get("segreduce_bwd")
        if sb:
            bgbs = sb["work"] / (sb["total_ms"] * 1e-3) / 1e9
            roof_seg["backward"] = {"achieved": round(bgbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                    "frac": round(bgbs / PEAK_HBM_GBS, 4), "avg_launch_us": round(sb["avg_us"], 2),
                                    "algorithmic_bytes_per_launch": sb["work"] / sb["launches"],
                                    "traffic": pmc_traffic("segreduce_bwd", workload_key),
                                    "from": "survey steps (every launch bracketed)"}
        if args.dense:
            # measured with SQ counters (profiles/r06_v1_sq_counters_dense.txt; bench.py cannot read PMC counters itself): the dense
            # aggregation is bound by vector-ALU ISSUE, one wave-instruction per 4 cycles per SIMD — its HBM fraction is not its roof
            roof_seg["valu_note"] = ("k_bar_fwd / k_bar_bwd are vector-ALU issue bound: SQ_INSTS_VALU 149.3 M / 214.9 M wave-instructions per launch "
                                     "x 4 cycles / 1024 SIMDs = 0.91 / 0.79 of the launch's SIMD cycles (profiles/r06_v1_sq_counters_dense.txt; "
                                     "DESIGN.md section 5): 30 / 38 instructions per 256 edge-channels, 14 of them the dropout hash")
        if seg_alone is not None:
            roof_seg["standalone_fwd"] = seg_alone
        bars_total = float(tot_nodes[1].item())
        value = bars_total * args.steps / elapsed
        fpb = flops_per_bar(float(tot_nodes[0].item()), bars_total, args.d, args.layers)
        line = {
            "metric": "bar-graphs/sec VAE fwd+bwd (+loss +Adam), LMD2 4-track x 32-ts synthetic",
            "value": round(value, 1), "unit": "bar-graphs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 (fp16-pair MFMA products)" if step_info.get("h2", 0) == 3 else "f32 (bf16-triple MFMA products)"),
            "data": "synthetic",
            "dtype_note": ("fp32 storage and accumulation; the products run on the 16-bit matrix pipe with split operands: the three "
                           "GCL products (d = 128 / 256 / 512; since round 6 also the dense-graph route at d = 512) as 3 fp16 MFMA products per fp32 product (fp16 pair format: 22-bit operands, "
                           "power-of-two scales from the tensors' |max|), every other split product as 6 bf16 MFMA products (exact "
                           "three-term split); full-size outputs 2e-6 .. 4e-6 from the fp64 oracle either way (the reference's own "
                           "fp32 arithmetic: 1e-4 .. 3e-4), profiles/r05_*parity*.jsonl"),
            "config": {"workload": ("dense-graph stress, one GPU's shard (BASELINE configs[4])" if args.dense else
                                    "LMD2 2-bar, 4 tracks, 32 ts, batch=256 per GPU, d_hidden=256 (BASELINE configs[1]; "
                                    "configs[3] when n_gpus=8)" if (args.d, args.batch, args.n_bars) == (256, 256, 2) else
                                    "LMD16 16-bar, batch=64, d_hidden=256 (BASELINE configs[2])" if (args.d, args.batch, args.n_bars) == (256, 64, 16) else
                                    "LMD2 2-bar, batch=256, d_hidden=512 (the reference's training.json)" if (args.d, args.batch, args.n_bars) == (512, 256, 2) else
                                    f"LMD {args.n_bars}-bar, batch={args.batch} per GPU, d_hidden={args.d}"),
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world, "n_bars": args.n_bars,
                       "d": args.d, "gnn_n_layers": args.layers, "nodes_per_gpu": n_nodes, "edges_per_gpu": n_edges,
                       "batch_seed": args.seed, "row_tiles_per_rank": [int(v) for v in tiles_all.tolist()],
                       "row_tiles_note": "64-row tiles of the rank's batch (seed + rank); the row-tile kernels hold one "
                                         "workgroup per CU (256): a batch with a few more tiles than CUs runs an XCD's extra "
                                         "tile as two 32-row halves behind two others (csrc/tile_order.h): +0.3-3 % step time "
                                         "at 257-261 tiles (DESIGN.md section 5)",
                       "message_dropout": 0.1, "parallelism": f"dp{world}", "weights": "default init, manual_seed(0)",
                       # which kernel set the step took (pm_vae_step_info: compact GCL, bf16 planes, fragment-major weights,
                       # active slots, the library's effective switches) and the launches per step of every profiled class:
                       # a shape that falls back to the round-1 kernels shows here, not only in the rate
                       "kernel_path": dict({k: v for k, v in step_info.items() if k not in ("N", "E", "G", "B")},
                                           launches_per_step={k: survey[k]["launches"] / SURVEY for k in sorted(survey)}),
                       "step": "plan+fwd+loss+bwd+allreduce+Adam"},
            # SURVEY 8(d)'s operation count (7 products of N d^2 per GCL) and what the step executes: the compact GCL
            # contracts K = 4d (one track block per node) and skips all-zero onset / next blocks tile by tile
            "step_flops_model": {"algorithmic_gflop_per_bar": round(fpb / 1e9, 3),
                                 "algorithmic_TFLOP/s_per_gpu": round(value * fpb / 1e12 / world, 2),
                                 "executed_gflop_per_bar": round(3.0 * (float(tot_nodes[0].item()) / bars_total) * (
                                     (60 + 16 * args.layers * exec_frac) * args.d ** 2 + 6900 * args.d) / 1e9, 3),
                                 "note": "algorithmic = SURVEY 8(d) (14 N d^2 per GCL); executed = compact GCL (8 N d^2) "
                                         "x the fraction of (tile, block) pairs that are not all-zero"},
            "losses_first_step": ({k: round(v, 5) for k, v in losses_first.items()} if losses_first else None),
            "losses": {k: round(v, 5) for k, v in losses.items()},
            "losses_note": ("`losses_first_step`: step 1 from the default init (reproducible; pinned against the oracle in "
                            "tests/test_fullsize_gpu.py).  `losses`: last of warmup + steps Adam steps on ONE fixed batch; "
                            "the unweighted kld goes through a transient at beta = 0 that the fp64 oracle shows as well "
                            "(tests/test_native_step_gpu.py::test_loss_trajectory_follows_the_fp64_oracle)"),
            "roofline": roof, "roofline_segreduce": roof_seg,
        }
        if dp is not None:
            line["dp"] = dp
        default_run = (args.d, args.batch, args.n_bars, args.layers, args.dense) == (256, 256, 2, 8, False)
        if world == 1 and default_run and not args.no_other_workloads:
            # the other single-GPU configurations of BASELINE.json, so that the driver's record carries them (untimed part)
            del trainer, vae
            torch.cuda.empty_cache()
            line["other_workloads"] = [
                # ranks 1..7 of a data-parallel run draw these batches: more 64-row tiles than CUs (the default batch has exactly 256)
                other_workload("BASELINE configs[1] with the batch of seed 1235 (258 row tiles)", 256, 256, 2, False, seed=1235),
                other_workload("BASELINE configs[1] with the batch of seed 1236 (261 row tiles)", 256, 256, 2, False, seed=1236),
                other_workload("LMD16 16-bar, batch=64, d_hidden=256 (BASELINE configs[2])", 64, 256, 16, False),
                other_workload("LMD2 2-bar, batch=256, d_hidden=512 (the reference's training.json)", 256, 512, 2, False),
                other_workload("dense-graph stress, one GPU's shard: batch=64, d_hidden=512 (BASELINE configs[4])", 64, 512, 2, True, steps=3),
                # shapes the synthetic spec of SURVEY App. D does not draw: every token slot in use (up to 14 notes per cell,
                # constants.py:48), denser bars
                other_workload("BASELINE configs[1] with up to 14 notes per cell: all 15 token slots active", 256, 256, 2, False, max_notes=14),
                other_workload("BASELINE configs[1] with denser bars (p = 0.5: 65 nodes, 322 edges per bar)", 256, 256, 2, False, p=0.5),
                # the drop-in boundary driven by the reference's own loop shape (torch loss, GradScaler, torch Adam)
                reference_loop_workload("reference loop shape at BASELINE configs[1]: vae(graph) under fp16 autocast -> _losses (torch) -> "
                                        "GradScaler backward -> torch.optim.Adam; decoder head over all 15 slots (as the reference)", False),
                reference_loop_workload("the same with vae.active_slots_only = True (logits of slots that are PAD in every node are "
                                        "not computed: same loss, same gradients)", True),
                generation_workload(),
            ]
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_subprocess(cfg)
            if line["cpu_baseline"].get("value"):
                line["speedup_vs_cpu_baseline"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
