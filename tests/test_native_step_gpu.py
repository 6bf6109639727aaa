"""The C++-orchestrated training step (csrc/vae_step.hip) against the Python-orchestrated one
(engine.py, the path the golden-vector tests pin) on identical inputs: same kernels in the same
order, so everything must agree to float-atomics reordering noise.  Plus the golden check itself
through the python orchestration (the default `HipTrainer` is native)."""
import json

import pytest
import torch

from polyphemus_amd._lib import PROF_NCLASS
from polyphemus_amd.model import VAE
from polyphemus_amd.synthetic import synthetic_batch
from polyphemus_amd.trainer import HipTrainer
from util import REL_TOL, batch_from_golden, load_case, rel_err, state_dict_from_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("B,nb,d,L,fix", [(8, 2, 64, 2, False), (6, 3, 32, 3, True), (16, 2, 256, 2, False),
                                          (1, 1, 8, 1, False), (3, 2, 24, 2, True), (2, 4, 40, 1, False)])
def test_native_step_equals_python_step(B, nb, d, L, fix):
    cfg = dict(dropout=0, batch_norm=True, gnn_n_layers=L, d=d, n_bars=nb, resolution=8)
    batch = synthetic_batch(B, nb, p=0.25, seed=3 + B).to(DEV)
    eps = torch.randn(B, d, generator=torch.Generator().manual_seed(100 + B)).to(DEV)
    results = []
    for native in (True, False):
        torch.manual_seed(0)
        vae = VAE(**cfg, device=DEV).to(DEV)
        vae.train()                                            # message dropout p = 0.1 stays ON
        tr = HipTrainer(vae, lr=5e-6, native=native, structure_loss_on_logits=fix)
        losses, g1 = [], None
        for i in range(3):
            losses.append(tr.losses_dict(tr.train_step(batch, eps)))
            if i == 0:
                g1 = tr.grads.clone()
        results.append((losses, {k: v.detach().clone() for k, v in vae.state_dict().items()}, g1))
    (la, sa, ga), (lb, sb, gb) = results
    for k in la[0]:                                            # first step: identical inputs and weights
        # (native: active slots + closed-form PAD tail, planes / K = 4d products; B = 2 batches sit at ~5e-6 in the KLD
        #  through their two-row BatchNorm statistics, everything else at ~1e-7)
        assert abs(la[0][k] - lb[0][k]) <= 1e-5 * max(1.0, abs(lb[0][k])), k
    # The two orchestrations differ in GEMM arithmetic (pre-split bf16 planes / K = 4d against fp32 MFMA / K = 7d) and in
    # atomics order; on these small random-init batches the reference arithmetic amplifies such rounding differences
    # (the CPU oracle itself moves by ~1e-2 in relative L2 between 1 and 8 threads), typically to
    # 1e-5 and occasionally to a few 1e-4 of the largest gradient.  A wrong kernel shows up at 1e-2 .. 1.
    assert rel_err(ga, gb) < 2e-3
    for x, y in zip(la[1:], lb[1:]):                           # later steps: Adam has amplified atomics-order noise
        for k in x:
            assert abs(x[k] - y[k]) <= 1e-4 * max(1.0, abs(y[k])), k
    for k in sa:
        if sa[k].dtype.is_floating_point:
            assert float((sa[k] - sb[k]).abs().max()) <= 1e-5 * max(1.0, float(sb[k].abs().max())) + 4e-5, k
        else:
            assert torch.equal(sa[k], sb[k]), k
    # tight check on tensors whose gradient is far above rounding noise
    for k in ("encoder.c_encoder.graph_encoder.layers.0.weight", "decoder.c_decoder.chord_decoder.weight",
              "encoder.c_encoder.chord_encoder.weight", f"decoder.c_decoder.graph_decoder.layers.{L - 1}.nn.weight"):
        a, b = sa[k].double(), sb[k].double()
        assert float((a - b).abs().mean()) < 1e-6, k


@pytest.mark.parametrize("case", ["lmd2_tiny", "nb3_tiny", "d128_l2"])
def test_measured_native_path_matches_reference_golden(case):
    """The variant bench.py measures — compact GCL (K = 4d), GEMM operands as bf16 planes, row classes, active slots,
    and at d = 128 the B-direct forward / input-gradient GEMMs — against the tensors captured from the reference:
    model outputs (model.py:676-678) 1e-4, the 7 losses, every gradient, BatchNorm buffers.  The test asserts WHICH
    variant ran (`pm_vae_step_info` + the launch-class counters of the in-library profiler)."""
    import ctypes
    from polyphemus_amd._lib import lib
    z, cfg = load_case(case)
    vae = VAE(**cfg, device=DEV).to(DEV)
    vae.load_state_dict(state_dict_from_golden(z))
    vae.train()
    vae.msg_dropout = 0.0                                       # the goldens were captured with GCL.dropout = 0
    g = batch_from_golden(z, cfg).to(DEV)
    assert g.track_unique and g.n_slots < 15
    eps = torch.from_numpy(z["in/eps"]).to(DEV)
    optcfg = json.loads(str(z["opt"]))
    tr = HipTrainer(vae, lr_scheduler=optcfg["lr_scheduler"], **optcfg["optimizer"])
    tr.keep_logits = True                                       # (the fused un-embedding + CE also stores the logits)
    L = lib()
    L.pm_prof_configure(-1, 1)
    L.pm_prof_begin(512)
    got = tr.losses_dict(tr.train_step(g, eps))
    ms, work, cnt = (ctypes.c_double * PROF_NCLASS)(), (ctypes.c_double * PROF_NCLASS)(), (ctypes.c_int64 * PROF_NCLASS)()
    L.pm_prof_end(*(ctypes.cast(a, ctypes.c_void_p) for a in (ms, work, cnt)))
    info = tr.step_info()
    d, nl = cfg["d"], cfg["gnn_n_layers"]
    assert info["compact"] == 1 and info["planes"] == 1 and info["n_slots"] == g.n_slots, info
    planes_tn, planesb_nn, planesb_nt, planes_nn, planes_nt = cnt[8 * 3 + 2], cnt[9 * 3], cnt[9 * 3 + 1], cnt[8 * 3], cnt[8 * 3 + 1]
    if d % 128 == 0:                                            # the three layer products: the kernels of gcl.hip
        assert info["b_frag"] == 1 and cnt[35] == 2 * nl and cnt[36] == 2 * nl and cnt[37] == 2 * nl and planesb_nn == 0 and planesb_nt == 0 and planes_tn == 0, (info, list(cnt))
    else:
        assert planes_nn == 2 * nl and planes_nt == 2 * nl and planes_tn == 2 * nl, list(cnt)
    for k, v in json.loads(str(z["train1/losses"])).items():
        assert abs(got[k] - v) <= REL_TOL * max(1.0, abs(v)), k
    (s_logits, c_logits), mu, lv = tr.step_outputs()
    S = info["n_slots"]
    assert rel_err(s_logits, z["train1/s_logits"]) < REL_TOL
    assert rel_err(c_logits, z["train1/c_logits"][:, :S]) < REL_TOL
    assert rel_err(mu, z["train1/mu"]) < REL_TOL and rel_err(lv, z["train1/log_var"]) < REL_TOL
    # Gradients.  The reference's fp32 gradients are themselves only defined up to their distance from exact arithmetic
    # (BatchNorm over near-constant channels amplifies summation-order rounding: up to 2.7e-2 of a tensor's scale on
    # this fixture, tools/golden_fp64_diag.py), so the anchor is the oracle run in fp64 — the same restatement that
    # reproduces the golden bit for bit in fp32 (tests/test_oracle_golden.py): every gradient tensor of the HIP path
    # within 1e-4 of the fp64 result (measured <= 3e-5), and no further from the reference's fp32 tensor than that
    # tensor is from fp64 (+ the 1e-4 bar).  No carve-out for analytically-zero gradients is needed against fp64.
    from util import fp64_oracle_grads
    names = [n for n, _ in vae.named_parameters()]
    g64 = fp64_oracle_grads(z, cfg)
    none = set(str(n) for n in z["train1/grad_none"])
    gmax = max(float(g64[n].abs().max()) for n in names if g64[n] is not None)
    for n in names:
        if n in none:
            assert g64[n] is None and float(tr._G[n].abs().max()) == 0.0, n
            continue
        h, gold, o = tr._G[n].detach().cpu().double(), torch.from_numpy(z[f"train1/grad/{n}"]).double(), g64[n]
        den = max(float(o.abs().max()), 1e-2 * gmax)
        e_hip, e_gold, e_hg = (float((a - b).abs().max()) / den for a, b in ((h, o), (gold, o), (h, gold)))
        assert e_hip < REL_TOL, (n, e_hip)
        assert e_hg <= e_gold + REL_TOL, (n, e_hg, e_gold)
    sd = vae.state_dict()
    for k, v in state_dict_from_golden(z, "train1/sd_after/").items():
        if "running_" in k:
            assert rel_err(sd[k], v) < REL_TOL, k
        elif not v.dtype.is_floating_point:
            assert torch.equal(sd[k].cpu(), v), k


@pytest.mark.parametrize("case", ["lmd2_tiny", "nb3_tiny"])
def test_python_orchestrated_trainer_matches_reference_losses(case):
    z, cfg = load_case(case)
    vae = VAE(**cfg, device=DEV).to(DEV)
    vae.load_state_dict(state_dict_from_golden(z))
    vae.train()
    vae.msg_dropout = 0.0
    g = batch_from_golden(z, cfg).to(DEV)
    eps = torch.from_numpy(z["in/eps"]).to(DEV)
    optcfg = json.loads(str(z["opt"]))
    tr = HipTrainer(vae, lr_scheduler=optcfg["lr_scheduler"], native=False, **optcfg["optimizer"])
    for step in (1, 2):
        got = tr.losses_dict(tr.train_step(g, eps))
        for k, v in json.loads(str(z[f"train{step}/losses"])).items():
            assert abs(got[k] - v) <= REL_TOL * max(1.0, abs(v)), (step, k)


def test_optimizer_state_interop_with_torch_adam():
    """`optimizer_state_dict()` is loadable by torch.optim.Adam (the reference's checkpoint format, training.py:518),
    and a trainer restored from it continues exactly like the original."""
    cfg = dict(dropout=0, batch_norm=True, gnn_n_layers=2, d=32, n_bars=2, resolution=8)
    batch = synthetic_batch(6, 2, p=0.25, seed=9).to(DEV)
    eps = torch.randn(6, 32, device=DEV)
    torch.manual_seed(0)
    vae = VAE(**cfg, device=DEV).to(DEV)
    vae.train()
    vae.msg_dropout = 0.0
    tr = HipTrainer(vae, lr=1e-4)
    for _ in range(2):
        tr.train_step(batch, eps)
    sd = tr.optimizer_state_dict()
    ref = torch.optim.Adam(vae.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-9)
    ref.load_state_dict(sd)                                        # the layout torch expects
    names = [n for n, _ in vae.named_parameters()]
    p0 = dict(vae.named_parameters())[names[0]]
    assert torch.equal(ref.state[p0]["exp_avg"], sd["state"][0]["exp_avg"])
    model_sd = {k: v.clone() for k, v in vae.state_dict().items()}
    model_flat = vae.flat_params.clone()
    tr.train_step(batch, eps)
    want = vae.flat_params.clone()
    torch.manual_seed(0)
    vae2 = VAE(**cfg, device=DEV).to(DEV)
    vae2.load_state_dict(model_sd)
    vae2.train()
    vae2.msg_dropout = 0.0
    vae2.seed, vae2._step = vae.seed, 4                            # (dropout is off: the seeds do not matter)
    tr2 = HipTrainer(vae2, lr=1e-4)
    tr2.load_optimizer_state_dict(sd)
    assert tr2.step_count == 2
    tr2.train_step(batch, eps)
    # same moments + same step count -> same update; elements whose gradient is atomics-order rounding noise may
    # move by up to +-lr with either sign (see test_model_gpu.params_close), everything else agrees closely
    diff = (vae2.flat_params - want).abs()
    assert float(diff.max()) <= 2.5e-4 and float(diff.mean()) < 2e-6
    step = (want - model_flat).abs()
    assert float(((vae2.flat_params - model_flat) * (want - model_flat)).sum() / (step.norm() ** 2)) > 0.98


def test_native_step_on_foreign_graphs_takes_the_seven_block_path():
    """A graph that does NOT follow the reference's construction rules (a node receiving track edges of two relations)
    breaks the premise of the compact GCL (`track_unique` False): the native step must then use the 7-block aggregate
    and agree with the python orchestration (which always does)."""
    cfg = dict(dropout=0, batch_norm=True, gnn_n_layers=2, d=32, n_bars=2, resolution=8)
    batch = synthetic_batch(6, 2, p=0.3, seed=4)
    et = batch.edge_type.clone()
    trk = torch.nonzero(et < 4).flatten()
    et[trk[::3]] = (et[trk[::3]] + 1) % 4                          # re-label a third of the track edges
    batch.edge_type = et
    batch.track_unique = False
    batch.__dict__.pop("_edge_attrs", None)
    batch = batch.to(DEV)
    eps = torch.randn(6, 32, device=DEV)
    res = []
    for native in (True, False):
        torch.manual_seed(0)
        vae = VAE(**cfg, device=DEV).to(DEV)
        vae.train()
        tr = HipTrainer(vae, lr=5e-6, native=native)
        out = tr.losses_dict(tr.train_step(batch, eps))
        res.append((out, tr.grads.clone()))
    (la, ga), (lb, gb) = res
    for k in la:
        assert abs(la[k] - lb[k]) <= 1e-6 * max(1.0, abs(lb[k])), k
    assert rel_err(ga, gb) < 1e-4


def test_debug_mode_rejects_a_wrong_track_unique_flag(monkeypatch):
    """A caller that attaches `track_unique = True` to a batch that breaks the rule would silently get a wrong compact
    GCL; under PM_DEBUG=1 the trainer reads the violation count the plan kernels keep (plan.hip cnt[4]) and raises."""
    monkeypatch.setenv("PM_DEBUG", "1")
    cfg = dict(dropout=0, batch_norm=True, gnn_n_layers=2, d=32, n_bars=2, resolution=8)
    batch = synthetic_batch(6, 2, p=0.3, seed=4)
    et = batch.edge_type.clone()
    trk = torch.nonzero(et < 4).flatten()
    et[trk[::3]] = (et[trk[::3]] + 1) % 4
    batch.edge_type = et
    batch.track_unique = True                                      # wrong on purpose
    batch.__dict__.pop("_edge_attrs", None)
    torch.manual_seed(0)
    vae = VAE(**cfg, device=DEV).to(DEV)
    vae.train()
    tr = HipTrainer(vae, lr=5e-6, native=True)
    with pytest.raises(RuntimeError, match="more than one track"):
        tr.train_step(batch.to(DEV), torch.randn(6, 32, device=DEV))
    good = synthetic_batch(6, 2, p=0.3, seed=4).to(DEV)             # a well-formed batch passes the same check
    tr.train_step(good, torch.randn(6, 32, device=DEV))


def test_native_step_with_all_fifteen_slots_active():
    """`n_slots` = 15 (no PAD-only tail: the chord encoder / decoder take their plain full-width path) must give the
    same step as the batch's own (smaller) active-slot count and as the python orchestration."""
    cfg = dict(dropout=0, batch_norm=True, gnn_n_layers=2, d=32, n_bars=2, resolution=8)
    base = synthetic_batch(6, 2, p=0.3, seed=21)
    assert base.n_slots < 15
    eps = torch.randn(6, 32, device=DEV)
    res = []
    for native, slots in ((True, 15), (True, base.n_slots), (False, base.n_slots)):
        batch = synthetic_batch(6, 2, p=0.3, seed=21)
        batch.n_slots = slots
        torch.manual_seed(0)
        vae = VAE(**cfg, device=DEV).to(DEV)
        vae.train()
        vae.msg_dropout = 0.0
        tr = HipTrainer(vae, lr=5e-6, native=native)
        out = tr.losses_dict(tr.train_step(batch.to(DEV), eps))
        res.append((out, tr.grads.clone()))
    for (la, ga) in res[:2]:
        for k in la:
            assert abs(la[k] - res[2][0][k]) <= 1e-6 * max(1.0, abs(res[2][0][k])), k
        assert rel_err(ga, res[2][1]) < 2e-3


def test_gradient_accumulation_matches_reference_loop():
    """`iters_to_accumulate` = 2 (training.py:149,158): backward of tot_loss / 2 on two different batches, ONE Adam update
    and ONE LR-schedule step per pair — against the drop-in module driven by the reference's own loop with
    torch.optim.Adam (the path pinned to the goldens in test_model_gpu)."""
    from oracle import vae_cpu
    cfg = dict(dropout=0, batch_norm=True, gnn_n_layers=2, d=32, n_bars=2, resolution=8)
    batches = [synthetic_batch(5 + i, 2, p=0.25, seed=40 + i).to(DEV) for i in range(4)]
    epss = [torch.randn(5 + i, 32, generator=torch.Generator().manual_seed(60 + i)).to(DEV) for i in range(4)]
    sched = dict(peak_lr=1e-4, warmup_steps=1, final_lr_scale=0.5, decay_steps=2)
    torch.manual_seed(0)
    ref = VAE(**cfg, device=DEV).to(DEV)
    ref.train()
    ref.msg_dropout = 0.0
    sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
    flat0 = ref.flat_params.clone()
    opt = torch.optim.Adam(ref.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-9)
    acc_grads = []
    for i, (g, e) in enumerate(zip(batches, epss)):
        mu, lv = ref.encoder(g)
        z = torch.exp(0.5 * lv) * e + mu
        s_logits, c_logits = ref.decoder(z, g)
        tot, _ = vae_cpu.losses(g.s_tensor, s_logits, g.c_tensor, c_logits, mu, lv)
        (tot / 2).backward()
        if (i + 1) % 2 == 0:
            acc_grads.append(torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten()
                                        for p in ref.parameters()]))
            opt.step()
            opt.zero_grad()
            for pg in opt.param_groups:
                pg["lr"] = vae_cpu.exp_decay_lr((i + 1) // 2, **sched)
    torch.manual_seed(0)
    vae = VAE(**cfg, device=DEV).to(DEV)
    vae.load_state_dict(sd0)
    vae.train()
    vae.msg_dropout = 0.0
    tr = HipTrainer(vae, lr=1e-4, lr_scheduler=sched, iters_to_accumulate=2)
    order = torch.cat([torch.arange(vae._offsets[n], vae._offsets[n] + p.numel()) for n, p in vae.named_parameters()])
    for i, (g, e) in enumerate(zip(batches, epss)):
        before = vae.flat_params.clone()
        tr.train_step(g, e)
        if (i + 1) % 2:
            assert torch.equal(vae.flat_params, before) and tr.step_count == i // 2      # no update on the odd batch
        else:
            assert tr.step_count == (i + 1) // 2
            if i == 1:                                                  # same weights on both sides for the first pair
                assert rel_err(tr.grad_accum[order.to(DEV)], acc_grads[0]) < 2e-3
    assert tr.micro_batches == 4 and abs(tr.lr - opt.param_groups[0]["lr"]) < 1e-12
    nbt = [v for k, v in vae.state_dict().items() if k.endswith("graph_encoder.norm_layers.0.module.num_batches_tracked")]
    assert int(nbt[0]) == 4                                             # BatchNorm sees every micro-batch
    diff = (vae.flat_params - ref.flat_params).abs()
    step = (ref.flat_params - flat0)
    assert float(diff.max()) <= 4.5e-4 and float(diff.mean()) < 4e-6   # +-lr per update on rounding-noise gradients
    assert float(((vae.flat_params - flat0) * step).sum() / (step.norm() ** 2)) > 0.98
    with pytest.raises(ValueError):
        HipTrainer(vae, iters_to_accumulate=0)


def test_evaluate_loop_and_checkpoint_round_trip(tmp_path):
    """`evaluate(loader)` = means of the per-batch values (training.py:250-296); `save_checkpoint` writes the
    reference's checkpoint layout (training.py:498-519) and `load_checkpoint` resumes from it."""
    import os
    cfg = dict(dropout=0, batch_norm=True, gnn_n_layers=2, d=32, n_bars=2, resolution=8)
    batches = [synthetic_batch(4 + i, 2, p=0.25, seed=70 + i).to(DEV) for i in range(3)]
    sched = dict(peak_lr=1e-4, warmup_steps=1, final_lr_scale=0.5, decay_steps=4)
    torch.manual_seed(0)
    vae = VAE(**cfg, device=DEV).to(DEV)
    vae.train()
    vae.msg_dropout = 0.0
    tr = HipTrainer(vae, lr=1e-4, lr_scheduler=sched)
    epss = [torch.randn(4 + i, 32, generator=torch.Generator().manual_seed(i)).to(DEV) for i in range(3)]
    for b, e in zip(batches[:2], epss):
        tr.train_step(b, e)
    # ---- evaluate
    torch.manual_seed(1)
    per = [tr.evaluate_batch(b) for b in batches]
    torch.manual_seed(1)

    class Loader:                                       # evaluate() draws eps itself: same generator state, same values
        def __iter__(self):
            return iter(batches)
    losses, accs = tr.evaluate(Loader())
    assert vae.training
    assert set(losses) == {"tot", "pitch", "dur", "structure", "reconstruction", "kld", "beta*kld"} and len(accs) == 9
    for k in losses:
        assert abs(losses[k] - sum(p[0][k] for p in per) / 3) < 1e-9, k
    for k in accs:
        want = sum(p[1][k] for p in per) / 3
        assert (accs[k] != accs[k] and want != want) or abs(accs[k] - want) < 1e-12, k
    # ---- checkpoint
    path = os.path.join(tmp_path, "checkpoint")
    tr.save_checkpoint(path, epoch=3, lrs=[1e-4, tr.lr])
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert list(ck["model_state_dict"]) == list(vae.state_dict()) and ck["tot_batches"] == 2 and ck["epoch"] == 3
    torch.optim.Adam(vae.parameters()).load_state_dict(ck["optimizer_state_dict"])     # the reference's resume path
    tr.train_step(batches[2], epss[2])
    want, want_lr = vae.flat_params.clone(), tr.lr
    torch.manual_seed(5)
    vae2 = VAE(**cfg, device=DEV).to(DEV)                       # different init: everything must come from the file
    vae2.train()
    vae2.msg_dropout = 0.0
    tr2 = HipTrainer(vae2, lr=3e-3, lr_scheduler=sched)
    rest = tr2.load_checkpoint(path)
    assert rest["epoch"] == 3 and tr2.step_count == 2 and tr2.micro_batches == 2
    for k, v in vae2.state_dict().items():
        assert torch.equal(v.cpu(), ck["model_state_dict"][k]), k
    tr2.train_step(batches[2], epss[2])
    assert abs(tr2.lr - want_lr) < 1e-15
    diff = (vae2.flat_params - want).abs()
    assert float(diff.max()) <= 2.5e-4 and float(diff.mean()) < 2e-6


def _unfused_ce_worker(rank, world, case):
    """Fresh process (the switch is read when the library loads): the native step with the three un-embedding products +
    the loss kernel instead of the fused un-embedding + cross-entropy kernel."""
    import os
    os.environ["PM_FUSED_CE"] = "0"
    z, cfg = load_case(case)
    vae = VAE(**cfg, device=DEV).to(DEV)
    vae.load_state_dict(state_dict_from_golden(z))
    vae.train()
    vae.msg_dropout = 0.0
    tr = HipTrainer(vae)
    tr.keep_logits = True
    out = tr.losses_dict(tr.train_step(batch_from_golden(z, cfg).to(DEV), torch.from_numpy(z["in/eps"]).to(DEV)))
    (_, c_logits), _, _ = tr.step_outputs()
    return out, tr.grads.detach().cpu().numpy(), c_logits.cpu().numpy()


@pytest.mark.parametrize("case", ["lmd2_tiny", "d128_l2"])
def test_native_step_with_fused_unembed_ce_matches_the_unfused_step(case):
    """The default step (fused un-embedding + cross-entropy, csrc/unembed.hip, SURVEY 8(f).2: at d = 128 the bf16-planes
    kernel, at d = 32 the fp32-MFMA one) against PM_FUSED_CE=0 — three un-embedding products + the loss kernel — on a
    reference-captured batch: same losses (and the goldens'), logits, and gradients.  (lf / gf / cf: the unfused run.)"""
    from util import run_ranks
    (lf, gf, cf), = run_ranks(_unfused_ce_worker, 1, (case,), timeout=120.0)
    z, cfg = load_case(case)
    vae = VAE(**cfg, device=DEV).to(DEV)
    vae.load_state_dict(state_dict_from_golden(z))
    vae.train()
    vae.msg_dropout = 0.0
    tr = HipTrainer(vae)
    tr.keep_logits = True
    lu = tr.losses_dict(tr.train_step(batch_from_golden(z, cfg).to(DEV), torch.from_numpy(z["in/eps"]).to(DEV)))
    (_, cu), _, _ = tr.step_outputs()
    for k, v in json.loads(str(z["train1/losses"])).items():
        assert abs(lf[k] - v) <= REL_TOL * max(1.0, abs(v)), k
        assert abs(lf[k] - lu[k]) <= 1e-6 * max(1.0, abs(lu[k])), k
    assert rel_err(torch.from_numpy(cf), cu) < 1e-5
    assert rel_err(torch.from_numpy(gf), tr.grads) < 1e-4


def _unfused_gcl_worker(rank, world, d, L, seed):
    """Fresh process (the switch is read once): the native step with PM_GCL_FUSED=0 — segment-reduce forward + the three
    grouped planes products instead of the kernels of csrc/gcl.hip; message dropout ON (same counter stream)."""
    import os
    os.environ["PM_GCL_FUSED"] = "0"
    return _synthetic_step(d, L, seed)


def _synthetic_step(d, L, seed):
    cfg = dict(dropout=0, batch_norm=True, gnn_n_layers=L, d=d, n_bars=2, resolution=8)
    batch = synthetic_batch(24, 2, p=0.25, seed=seed).to(DEV)
    eps = torch.randn(24, d, generator=torch.Generator().manual_seed(seed)).to(DEV)
    torch.manual_seed(0)
    vae = VAE(**cfg, device=DEV).to(DEV)
    vae.train()
    tr = HipTrainer(vae, lr=5e-6)
    tr.keep_logits = True
    out = tr.losses_dict(tr.train_step(batch, eps))
    (s_logits, c_logits), mu, _ = tr.step_outputs()
    info = tr.step_info()
    # (where the decoder head's gradients sit in the flat buffer: they are formed before the backward meets a ReLU)
    lo = vae._offsets["decoder.c_decoder.chord_decoder.weight"]
    info["head_lo"] = int(lo)
    return out, tr.grads.detach().cpu().numpy(), c_logits.cpu().numpy(), mu.cpu().numpy(), info


@pytest.mark.parametrize("d,L", [(128, 2), (256, 3)])
def test_gcl_kernels_match_the_segment_reduce_plus_grouped_product_step(d, L):
    """The measured step (csrc/gcl.hip: fused forward, A-stationary input gradient, 128x128 weight gradient) against
    the same step built from the round-1 kernels (PM_GCL_FUSED=0, child process), message dropout p = 0.1 on: same losses,
    outputs and gradients up to fp32 accumulation order and the order of the atomic adds."""
    from util import run_ranks
    (lu, gu, cu, mu_u, info_u), = run_ranks(_unfused_gcl_worker, 1, (d, L, 31), timeout=180.0)
    lf, gf, cf, mu_f, info_f = _synthetic_step(d, L, 31)
    assert info_f["compact"] == 1 and info_f["planes"] == 1 and info_f["gcl_fused"] == 1 and info_u["gcl_fused"] == 0
    assert info_f["h2"] == 3 and info_u["h2"] == 0     # (the fp16 pair format exists for the kernels of gcl.hip only)
    assert {k: v for k, v in info_f.items() if k not in ("gcl_fused", "h2")} == {k: v for k, v in info_u.items() if k not in ("gcl_fused", "h2")}
    for k in lf:
        assert abs(lf[k] - lu[k]) <= 2e-6 * max(1.0, abs(lu[k])), k
    assert rel_err(torch.from_numpy(cf), torch.from_numpy(cu)) < 2e-5
    assert rel_err(torch.from_numpy(mu_f), torch.from_numpy(mu_u)) < 2e-5
    # Gradients.  The two kernel sets use different operand formats (fp16 pair against the exact bf16 triple), so their
    # activations differ at the 1e-6 level and a pre-activation that close to zero may take the other ReLU decision: ONE such
    # element moves the gradient by ~1 / sqrt(N d) = 1.6e-3 in relative L2 here (tests/test_fullsize_gpu.py imposes the
    # decisions and finds nothing else).  Hence: the decoder head's gradients, formed before the backward meets a ReLU, to
    # 2e-5; the whole gradient to what a handful of flipped decisions can move it.
    gf, gu = torch.from_numpy(gf).double(), torch.from_numpy(gu).double()
    lo = info_f["head_lo"]
    assert float((gf[lo:] - gu[lo:]).norm() / gu[lo:].norm()) < 2e-5
    assert float((gf - gu).norm() / gu.norm()) < 4e-3


def test_loss_trajectory_follows_the_fp64_oracle():
    """bench.py trains ONE fixed synthetic batch with training.json's optimizer (Adam, lr 1e-4 in the warm-up
    of the schedule the bench uses, beta = 0) and prints the last step's losses; the unweighted KLD goes through a
    transient of several orders of magnitude in the first steps.  Is that the model or a fault of the HIP path (which
    shares its BatchNorm, loss and Adam kernels between all kernel sets)?  Six steps of oracle/vae_cpu.py in fp64 with
    stock torch Adam on the same batch, weights, eps and schedule, message dropout off (so that both paths see the same
    forward): the trajectories agree — every loss of steps 1-3 to 1e-3, the reconstruction terms of steps 1-4 to 1e-3 and
    of steps 5-6 to 1e-2 (the trajectories separate slowly) — and the ORACLE's KLD grows by more than 10x within the six
    steps as well: the transient is the reference's model at beta = 0, not a HIP fault."""
    import math
    from oracle import vae_cpu
    from util import _as_dtype
    B, nb, d, L = 64, 2, 256, 8
    cfg = dict(dropout=0, batch_norm=True, gnn_n_layers=L, d=d, n_bars=nb, resolution=8)
    cpu = synthetic_batch(B, nb, p=0.25, seed=1234)
    torch.manual_seed(0)
    vae = VAE(**cfg, device=DEV).to(DEV)
    vae.train()
    vae.msg_dropout = 0.0
    sd = {k: v.detach().cpu().clone() for k, v in vae.state_dict().items()}
    names = [n for n, _ in vae.named_parameters()]
    tj = dict(peak_lr=1e-4, final_lr_scale=0.01, warmup_steps=8000, decay_steps=800000)        # as bench.py (training.json:19-24)
    tr = HipTrainer(vae, lr=5e-6, betas=(0.9, 0.98), eps=1e-9, lr_scheduler=tj)
    gen = torch.Generator().manual_seed(7)
    eps = [torch.randn(B, d, generator=gen) for _ in range(6)]
    gpu_batch = cpu.to(DEV)
    hip = [tr.losses_dict(tr.train_step(gpu_batch, e.to(DEV))) for e in eps]
    P, _ = vae_cpu.split_state({k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}, names)
    # (the reference builds Adam with training.json's lr = 5e-6 and its scheduler overwrites the learning rate AFTER every
    #  update, training.py:58-60,169: the first update runs at 5e-6, the following ones at the warm-up's 1e-4)
    opt = torch.optim.Adam([P[n] for n in names], lr=5e-6, betas=(0.9, 0.98), eps=1e-9)
    b64 = _as_dtype(cpu, torch.float64)
    ora = []
    nthr = torch.get_num_threads()
    try:
        import os
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import host_cores
        torch.set_num_threads(host_cores())
        for e in eps:
            _, parts, _ = vae_cpu.train_step(b64, P, names, cfg, opt, e.double(), msg_dropout=0.0)
            ora.append({k: float(v.detach()) for k, v in parts.items()})
            opt.param_groups[0]["lr"] = tj["peak_lr"]
    finally:
        torch.set_num_threads(nthr)
    for i in range(6):
        for k in ("pitch", "dur", "structure"):
            assert abs(hip[i][k] - ora[i][k]) <= (1e-3 if i < 4 else 1e-2) * max(1.0, abs(ora[i][k])), (i, k, hip[i][k], ora[i][k])
        assert math.isfinite(hip[i]["kld"]) and hip[i]["kld"] > 0
    for i in range(3):
        assert abs(hip[i]["kld"] - ora[i]["kld"]) <= 1e-3 * max(1.0, abs(ora[i]["kld"])), (i, hip[i]["kld"], ora[i]["kld"])
    print("kld trajectory  HIP:", [round(h["kld"], 3) for h in hip], " fp64 oracle:", [round(o["kld"], 3) for o in ora])
    assert max(o["kld"] for o in ora) > 10.0 * ora[0]["kld"], [o["kld"] for o in ora]         # the oracle's own transient
    # The transient multiplies the KLD ~6x per step at its end, and with it every rounding difference: the SIXTH step moves from
    # run to run of one binary with the order of the step's float atomics — 341 .. 399 in eleven runs of the round-5 and round-6
    # libraries on one box (oracle 343; profiles/LOG.md, round 6), the fifth 62.5 .. 64.8 (oracle 63.7).  So: the fifth step within
    # 4 % of the oracle's, the peak inside the band that spread allows (a 15 % band here failed one run in four, both rounds).
    assert abs(hip[4]["kld"] / ora[4]["kld"] - 1.0) < 0.04, (hip[4]["kld"], ora[4]["kld"])
    assert 0.75 < max(h["kld"] for h in hip) / max(o["kld"] for o in ora) < 1.35, ([h["kld"] for h in hip], [o["kld"] for o in ora])
    # (while the trajectories have not separated, the KLDs agree in order of magnitude at every step)
    for i in range(6):
        assert 0.2 < hip[i]["kld"] / ora[i]["kld"] < 5.0, (i, hip[i]["kld"], ora[i]["kld"])


@pytest.mark.parametrize("d,batch_norm,p_cfg", [(128, True, 0.2), (128, False, 0.0), (256, False, 0.25), (64, False, 0.1)])
def test_native_step_covers_the_constructor_switches(d, batch_norm, p_cfg):
    """`batch_norm=False` (model.py:176-188,218-238,278-292) and `dropout != 0` (the element dropout layers) through the
    C++ step — with the kernels of gcl.hip / linear.hip at d = 128 / 256 — against the Python orchestration of the same
    configuration (which tests/test_model_gpu.py pins to the oracle): same counter-hash masks, so everything agrees to
    accumulation order."""
    from polyphemus_amd import _lib
    B, nb, L = 12, 2, 2
    cfg = dict(dropout=p_cfg, batch_norm=batch_norm, gnn_n_layers=L, d=d, n_bars=nb, resolution=8)
    batch = synthetic_batch(B, nb, p=0.25, seed=21).to(DEV)
    eps = torch.randn(B, d, generator=torch.Generator().manual_seed(9)).to(DEV)
    res = []
    _lib.set_deterministic(True)              # (one fixed realisation of both steps: a ReLU decision flipped by atomics order
    try:                                      #  moves small-batch gradients by 1e-3)
        for native in (True, False):
            torch.manual_seed(0)
            vae = VAE(**cfg, device=DEV).to(DEV)
            vae.train()
            tr = HipTrainer(vae, lr=5e-6, native=native, structure_loss_on_logits=True)
            assert tr.native == native
            loss = tr.losses_dict(tr.train_step(batch, eps))
            res.append((loss, tr.grads.clone(), {k: v.detach().clone() for k, v in vae.state_dict().items()}))
    finally:
        _lib.set_deterministic(False)
    (la, ga, sa), (lb, gb, sb) = res
    for k in la:
        assert abs(la[k] - lb[k]) <= 1e-5 * max(1.0, abs(lb[k])), k
    assert rel_err(ga, gb) < 2e-3
    live = gb.abs() > 1e-3 * gb.abs().max()
    assert float(((ga - gb).abs()[live] / gb.abs()[live]).median()) < 5e-4
    for k in sa:
        if sa[k].dtype.is_floating_point:
            assert float((sa[k] - sb[k]).abs().max()) <= 1e-5 * max(1.0, float(sb[k].abs().max())) + 4e-5, k
        else:
            assert torch.equal(sa[k], sb[k]), k
