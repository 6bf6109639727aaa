"""Parity at BASELINE.json's full sizes: the native HIP training step — the variant bench.py measures:
compact GCL, bf16-planes GEMM operands, the kernels of gcl.hip / linear.hip (d = 256) resp. wide.hip (d = 512), row
classes, active slots, message dropout p = 0.1 replayed from the counter hash — against oracle/vae_cpu.py on the same
synthetic batch, weights and eps, at

    configs[1]     LMD2  2-bar  B = 256  d = 256  L = 8   (the bench workload)
    configs[2]     LMD16 16-bar B = 64   d = 256  L = 8
    training.json  LMD2  2-bar  B = 256  d = 512  L = 8   (the reference's own training configuration)
    configs[4]     dense stress, one GPU's shard of the 8-GPU job reduced to B = 8 (d = 512, 16,256 edges per bar)

and, without the oracle (its per-edge fp64 tensors do not fit), property checks of configs[4]'s shard at its real size
(B = 64, N = 16,384, 2.08 M edges).

The oracle runs twice, in fp32 (the reference's arithmetic) and in fp64.  At these sizes the reference's own fp32
arithmetic sits 1e-4 .. 5e-4 from exact on the model outputs and 3e-2 .. 1.5e-1 on single gradient tensors (16
BatchNorm'd layers amplify summation-order rounding), so "within 1e-4 of the reference" is anchored on the fp64 result:

    outputs  (s_logits, c_logits, mu, log_var):  |HIP - fp64| <= 1e-4 rel (measured 2e-6 .. 8e-6), and
             |HIP - fp32 oracle| <= |fp32 oracle - fp64| + 1e-4  (the HIP path is no further from the reference than the
             reference is from exact arithmetic);
    losses   1e-6;
    gradients: ABSOLUTE caps per configuration on the worst tensor (max|a-b| over max(max|ref|, 1 % of the largest
             gradient)) and on the whole-vector relative L2 error against fp64, at about 3x the measured values
             (GRAD_CAPS below; measured values and their attribution: profiles/r03_fullsize_parity.json), and the L2
             error at most 0.4x the fp32 oracle's own (0.8x on the dense shard, which is chaotic at the 1e-3 level).

What the worst-tensor figure is worth (tools/parity_attribution.py, profiles/r03_fullsize_parity.json): it is a property of
the REALISATION (batch, dropout mask), not of a kernel set.  At configs[1] it is 1.32e-2 on
`decoder.c_decoder.bars_decoder.weight` with every kernel set (the round-1 kernels and the fp32-MFMA GCL products
included), repetitions agree to 1e-3 of it (atomics order), and over 8 realisations (4 batches x dropout on / off) it
ranges 2.0e-3 .. 4.5e-2 while the fp32 oracle's ranges 2.7e-2 .. 1.5e-1; the relative L2 error ranges 1.7e-4 .. 8.6e-4
against 2.9e-3 .. 5.3e-3.  (Round 2 quoted 1.9e-3 for this configuration before its dropout stream was re-defined:
another mask, another realisation; with the dropout off the same batch gives 2.1e-3.)"""
import os
import sys

import pytest
import torch

from util import DENSE_SHARD_B64, FULLSIZE, REL_TOL, SMALLSIZE, hip_fullsize_step, hip_vs_oracle_fullsize

pytestmark = pytest.mark.gpu

# (worst tensor, relative L2) of the gradient against the fp64 oracle: about 3x the values measured on the fixed
# realisation of each configuration (seed 1234)
GRAD_CAPS = {                                            # measured (profiles/r03_fullsize_parity.json):
    "configs1_lmd2_b256_d256": (4e-2, 1.5e-3),           # 1.32e-2 / 5.1e-4   (fp32 oracle: 5.2e-2 / 3.6e-3)
    "configs2_lmd16_b64_d256": (8e-3, 7e-4),             # 2.6e-3  / 2.3e-4   (5.1e-2 / 2.8e-3)
    "training_json_b256_d512": (2e-2, 1e-3),             # 5.7e-3  / 3.4e-4   (4.3e-2 / 3.0e-3)
    "configs1_seed1235_258_tiles": (3e-2, 1.7e-3),       # 9.1e-3  / 5.5e-4   (4.0e-2 / 3.0e-3)   [profiles/r03_fullsize_parity_final.jsonl]
    # the dense shard is chaotic at the 1e-3 level: every cell of every bar is active, all bars have the same graph, the
    # aggregates of a bar's nodes are nearly equal and the BatchNorms run over near-constant channels.  Repetitions of
    # the SAME step with the round-1 kernel set differ from each other by up to 1.1e-3 in relative L2 (atomics order);
    # kernel sets and realisations range 6e-4 .. 2.2e-3 (fp32 oracle 3.0e-3 .. 4.5e-3)
    "configs4_dense_shard_b8_d512": (1.2e-1, 5.5e-3),    # 3.9e-2  / 1.8e-3   (3.8e-2 / 3.0e-3)
}


def _expect_dedicated_kernels(info, L=8):
    """the three GCL products and the chord products of the step ran on gcl.hip / linear.hip / wide.hip (the chord ENCODER
    runs as table algebra, chord.hip: the decoder's two products remain)"""
    n = info["launches"]
    assert info.get("h2_clamp_events", 0) == 0, info        # (no operand of the fp16 pair format saturated: no gradient was clipped)
    assert info["compact"] == 1 and info["planes"] == 1 and info["b_frag"] == 1 and info["n_slots"] < 15, info
    assert info["chord_tables"] == 1 and info["dagg_bn"] == 1, info
    assert n["gcl_fwd"] == 2 * L and n["gcl_dagg"] == 2 * L and n["gcl_dw"] == 2 * L and n["rows_w"] == 2, info
    assert n["planesB_nn"] == 0 and n["planesB_nt"] == 0 and n["planes_tn"] == 0, info


@pytest.mark.parametrize("name", list(FULLSIZE))
def test_native_step_matches_oracle_at_full_size(name):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import host_cores
    rep = hip_vs_oracle_fullsize(FULLSIZE[name], threads=host_cores())
    _expect_dedicated_kernels(rep["info"])
    for k, e in rep["outputs"].items():
        assert e["hip_vs_o64"] < REL_TOL, (k, e)
        assert e["hip_vs_o32"] <= e["o32_vs_o64"] + REL_TOL, (k, e)
    for k, e in rep["losses"].items():
        assert e["hip_vs_o64"] < 1e-6, (k, e)
    g = rep["grad"]
    cap_w, cap_l2 = GRAD_CAPS[name]
    assert g["hip_vs_o64"]["worst_tensor_err"] < cap_w and g["hip_vs_o64"]["rel_l2"] < cap_l2, g
    # ... and never as far from exact arithmetic as the reference's own fp32 arithmetic is (measured 0.08 - 0.14 of it at
    # the sparse configurations, 0.2 - 0.6 on the chaotic dense shard)
    assert g["hip_vs_o64"]["rel_l2"] <= (0.8 if FULLSIZE[name]["dense"] else 0.4) * g["o32_vs_o64"]["rel_l2"], g


def test_deterministic_step_matches_fp64_oracle_at_the_bench_workload():
    """configs[1] in DETERMINISTIC mode (bit-reproducible: what this test sees is what every run sees, unlike the default
    mode whose realisation depends on atomics order) against the fp64 oracle, same bounds as above."""
    from polyphemus_amd import _lib
    from util import grad_errors, hip_fullsize_step, oracle_fullsize, rel_err
    name = "configs1_lmd2_b256_d256"
    spec = FULLSIZE[name]
    _lib.set_deterministic(True)
    try:
        run = hip_fullsize_step(spec)
        again = hip_fullsize_step(spec)
    finally:
        _lib.set_deterministic(False)
    _expect_dedicated_kernels(run["info"])
    for n in run["names"]:                                    # a second fresh model + step: the same bits
        if run["grads"][n] is not None:
            assert torch.equal(run["grads"][n], again["grads"][n]), n
    for k, v in run["outputs"].items():
        assert torch.equal(v, again["outputs"][k]), k
    res, _ = oracle_fullsize(spec, run, dtypes=(("o64", torch.float64),))
    o64, l64, g64 = res["o64"]
    S = run["info"]["n_slots"]
    for k in ("s_logits", "c_logits", "mu", "log_var"):
        ref = o64[k][:, :S] if k == "c_logits" else o64[k]
        assert rel_err(run["outputs"][k], ref) < REL_TOL, k
    for k in ("pitch", "dur", "structure", "kld"):
        assert abs(run["losses"][k] - l64[k]) / max(1.0, abs(l64[k])) < 1e-6, k
    g = grad_errors(run["names"], run["grads"], g64)["hip_vs_o64"]
    cap_w, cap_l2 = GRAD_CAPS[name]
    assert g["worst_tensor_err"] < cap_w and g["rel_l2"] < cap_l2, g


@pytest.mark.parametrize("name", ["configs1_lmd2_b256_d256", "configs1_seed1235_258_tiles", "configs2_lmd16_b64_d256", "training_json_b256_d512",
                                  "configs4_dense_shard_b8_d512", "small_b24_d128_l3"])
def test_gradient_is_the_fp64_oracles_under_the_relu_decisions_the_step_took(name):
    """WHY the default-mode gradient sits 2e-4 .. 8e-4 (relative L2) from the fp64 oracle at full size while every output is
    2e-6 away: the loss is piecewise smooth, and an fp32 step whose activations are ~1e-6 from the exact ones takes the other
    ReLU decision at the few hundred (of 67 M) elements whose pre-activation is that close to zero; each moves the gradient by
    ~1 / sqrt(N d).  The step's own decisions are read back (the tensors its backward decides from, pm_vae_step_saved /
    pm_bn_relu_decisions) and imposed on the fp64 oracle (oracle/kinks.ReluProbe.forced): with the SAME decisions the gradient
    of the default-mode step — the kernels bench.py measures — is the oracle's to BASELINE's 1e-4 (relative L2, and every tensor
    against max(|ref|max, 1 % of the largest gradient)), nothing left over.  GRAD_CAPS above then only guard the unforced
    comparison against regressions."""
    from oracle import kinks
    from util import grad_errors, hip_relu_decisions, oracle_fullsize
    spec = FULLSIZE.get(name) or SMALLSIZE[name]
    live = {}
    run = hip_fullsize_step(spec, lr=0.0, keep=live)          # (lr = 0: the norms' gamma / beta the decisions depend on stay put)
    forced = hip_relu_decisions(live, run["cfg"])
    live.clear()
    torch.cuda.empty_cache()
    with kinks.ReluProbe(forced=forced, keep=False) as probe:
        res, _ = oracle_fullsize(spec, run, dtypes=(("o64", torch.float64),))
    _, l64, g64 = res["o64"]
    flips = sum(probe.disagree.values())
    total = sum(int(m.numel()) for m in forced.values())
    g = grad_errors(run["names"], run["grads"], g64)
    print(f"{name}: {flips} of {total} imposed ReLU decisions differ from the oracle's own; gradient under the step's decisions: "
          f"relative L2 {g['hip_vs_o64']['rel_l2']:.2e}, worst tensor {g['hip_vs_o64']['worst_tensor_err']:.2e} "
          f"({g['hip_vs_o64']['worst_tensor']})")
    for k in ("pitch", "dur", "structure", "kld"):
        assert abs(run["losses"][k] - l64[k]) / max(1.0, abs(l64[k])) < 1e-6, k
    assert set(probe.disagree) == set(forced)                  # every imposed site was reached, in the oracle's call order
    assert g["hip_vs_o64"]["rel_l2"] < 1e-4, g["hip_vs_o64"]
    assert g["hip_vs_o64"]["worst_tensor_err"] < 1e-4, g["hip_vs_o64"]


def _switch(**env):
    """set / clear step switches and make the library re-read them (pm_vae_step_reload_switches)"""
    from polyphemus_amd._lib import lib
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    assert lib().pm_vae_step_reload_switches() == 0


def _rel_l2(a, b, names):
    num = sum(float(((a[n].double() - b[n].double()) ** 2).sum()) for n in names)
    den = sum(float((b[n].double() ** 2).sum()) for n in names)
    return (num / den) ** 0.5


HEAD_TENSORS = ("decoder.c_decoder.chord_decoder.", "decoder.c_decoder.drums_pitch_emb.", "decoder.c_decoder.non_drums_pitch_emb.",
                "decoder.c_decoder.dur_emb.")


def _same_step_up_to_relu_kinks(spec, env):
    """The step under the default switches and under `env` (two kernel sets: the same arithmetic in another accumulation
    order).  What can be asserted of two fp32 runs of a ReLU network: the losses agree to 1e-6, the model outputs to
    1e-5, the gradients of the decoder head — formed before the backward pass meets a ReLU — to 1e-5, the whole
    gradient to 3e-3 in relative L2.  Not tighter, because of ReLU kinks: activations of two runs differ at the 1e-7
    level (another accumulation order; the head GEMMs add their K slices with fp32 atomics, so this holds for two runs of
    ONE kernel set as well), an element of relu(BN(h)) within that distance of zero flips its derivative, and ONE flipped
    element moves the gradient by |du| / ||dh|| ~ 1 / sqrt(N d) in relative L2 — 1.1e-3 at N d = 8e5.  Measured at
    B = 24, d = 512: repetitions of one kernel set land on 5.6e-7 or 7.8e-4, nothing in between; two kernel sets on
    2.8e-4 or 8.2e-4.  (The kernels themselves are compared bit for bit in tests/test_kernels_gpu.py.)"""
    a = hip_fullsize_step(spec, lr=0.0)
    try:
        _switch(**env)
        b = hip_fullsize_step(spec, lr=0.0)
    finally:
        _switch(**{k: None for k in env})
    for k, v in a["outputs"].items():
        assert float((v - b["outputs"][k]).abs().max()) <= 1e-5 * float(v.abs().max()), k
    live = [k for k in a["names"] if float(a["grads"][k].abs().max()) > 0]
    head = [k for k in live if k.startswith(HEAD_TENSORS)]
    assert len(head) >= 8
    assert _rel_l2(a["grads"], b["grads"], head) < 1e-5
    assert _rel_l2(a["grads"], b["grads"], live) < 3e-3
    return a, b


@pytest.mark.parametrize("fix_structure", [False, True])
def test_structure_branch_on_the_second_stream_changes_nothing(fix_structure):
    """The step issues the structure encoder / decoder chains on the library's second stream (vae_step.hip: fork / join
    events around them; forward of both, backward of the encoder's, and — when the structure loss reaches the logits —
    backward of the decoder's).  Against PM_SIDE_STREAM=0 (everything on the caller's stream), at configs[1] where the
    branches really run beside the content path: same losses, outputs and gradients (to what two fp32 runs can agree on).
    The structure branch's own parameters all receive their gradient."""
    spec = dict(FULLSIZE["configs1_lmd2_b256_d256"], fix_structure=fix_structure)
    a, b = _same_step_up_to_relu_kinks(spec, {"PM_SIDE_STREAM": 0})
    for k in ("pitch", "dur", "structure", "kld"):
        assert abs(a["losses"][k] - b["losses"][k]) <= 1e-6 * max(1.0, abs(b["losses"][k])), k
    branch = [k for k in a["names"] if k.startswith(("encoder.s_encoder.",) + (("decoder.s_decoder.",) if fix_structure else ()))]
    assert len(branch) == (26 if fix_structure else 14)
    for k in branch:
        assert float(a["grads"][k].abs().max()) > 0, k
    assert _rel_l2(a["grads"], b["grads"], branch) < 3e-3


@pytest.mark.parametrize("env", [{"PM_DAGG_BN": 0}, {"PM_CHORD_TABLES": 0}, {"PM_PLAN_SIDE": 0}, {"PM_DAGG_RES": 1},
                                 {"PM_PAD_SKIP": 0}, {"PM_UNEMBED_DW": 0}],
                         ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_round4_rearrangements_change_nothing(env):
    """The step's rearrangements of round 4, each against the formulation it replaced, at configs[1]: the norm backward
    inside the input-gradient kernel (PM_DAGG_BN=0: its own pass), the chord encoder as table algebra (PM_CHORD_TABLES=0:
    gather + products over X [N, S, d]), the plan beside the content encoder (PM_PLAN_SIDE=0), the residual gradient in dA's
    self block (PM_DAGG_RES=1; off by default), and round 6's decoder head over the row lists without PAD targets (PM_PAD_SKIP=0:
    every active-slot row) and its three un-embedding weight gradients in one launch (PM_UNEMBED_DW=0: three split-K tile products) —
    same losses, outputs and gradients, to what two fp32 runs of a ReLU network
    can agree on.  (Switches the step re-reads: pm_vae_step_reload_switches.)"""
    a, b = _same_step_up_to_relu_kinks(dict(FULLSIZE["configs1_lmd2_b256_d256"]), env)
    for k in ("pitch", "dur", "structure", "kld"):
        assert abs(a["losses"][k] - b["losses"][k]) <= 1e-6 * max(1.0, abs(b["losses"][k])), k
    key = {"PM_DAGG_BN": "dagg_bn", "PM_CHORD_TABLES": "chord_tables", "PM_PAD_SKIP": "pad_skip"}.get(next(iter(env)))
    if key:                                         # (the switch reached the library: the two steps ran different kernel sets)
        assert a["info"][key] == 1 and b["info"][key] == 0, (a["info"], b["info"])
    chord = [k for k in a["names"] if "chord_encoder" in k or "emb" in k.split(".")[-2:][0]]
    if chord:
        assert _rel_l2(a["grads"], b["grads"], [k for k in chord if float(a["grads"][k].abs().max()) > 0]) < 3e-3


def test_dense_shard_at_its_real_size_properties():
    """configs[4], one GPU's shard (B = 64, d = 512, every cell active: N = 16,384, E = 2.08 M), without the oracle:
    (1) the step takes the dense route of the d = 512 kernels — stand-alone segment-reduce, then the product from its A'
        planes (`pm_gcl_forward_from_planes`), the input / weight gradients and the chord products on wide.hip / gcl.hip;
    (2) everything finite, the losses of a default-init model where they must be (CE ~ log of the vocabulary);
    (3) the same step through the ROUND-1 kernel set (`PM_GCL_FUSED=0`: segment-reduce + grouped planes products, the set
        the B = 8 shard pins to the oracle as well) gives the same losses (1e-6) and gradient (relative L2 < 1e-2: this
        configuration is chaotic at the 1e-3 level, see GRAD_CAPS — repetitions of ONE kernel set differ that much);
    (4) the gradient IS the derivative of the loss the step reports: central differences along the normalised gradient
        of the encoder parameters and along that of the decoder parameters agree with <g, v> to 1 % / 0.1 % (fp32 loss;
        steps that move the loss by 3e-4 / 1e-2)."""
    keep = {}
    run = hip_fullsize_step(DENSE_SHARD_B64, lr=0.0, keep=keep)
    info, names = run["info"], run["names"]
    n = info["launches"]
    assert info["N"] == 16384 and info["E"] == 2080768 and info["compact"] == 1 and info["planes"] == 1, info
    assert info["h2"] == 3 and info["h2_clamp_events"] == 0, info     # (round 6: the dense route runs the pair format too; nothing saturated)
    assert n["gcl_fwd"] == 16 and n["segreduce_fwd"] == 16 and n["gcl_dagg"] == 16 and n["gcl_dw"] == 16 and n["rows_w"] == 2, info
    assert n["planesB_nn"] == 0 and n["planesB_nt"] == 0 and n["planes_tn"] == 0, info
    for k, v in run["outputs"].items():
        assert bool(torch.isfinite(v).all()), k
    live = [k for k in names if float(run["grads"][k].abs().max()) > 0]
    assert all(bool(torch.isfinite(run["grads"][k]).all()) for k in names) and len(live) >= 100
    assert 3.0 < run["losses"]["pitch"] < 6.5 and 3.0 < run["losses"]["dur"] < 6.0, run["losses"]
    # (4) directional derivatives, on the live trainer (lr = 0: Adam leaves the parameters alone; the dropout stream is
    # rewound before every step so that each evaluation sees the same masks)
    vae, tr, batch, eps, step0 = (keep[k] for k in ("vae", "trainer", "batch", "eps", "step0"))
    flat = vae.flat_params
    g = tr.grads.detach().clone()
    theta = flat.detach().clone()

    def loss_at(delta):
        flat.copy_(theta + delta)
        vae._step = step0
        l = tr.losses_dict(tr.train_step(batch, eps))
        return l["tot"]

    dec_lo = vae._offsets[vae._names("decoder.")[0]]                # flat order: encoder parameters, then the decoder's
    # (measured, tools: the decoder direction is smooth — 5e-6 at a loss change of 1e-2; along the encoder direction the
    #  central difference converges more slowly, -1.7e-2 / -1.9e-3 at loss changes of 1e-3 / 3e-4; repeated evaluations of the
    #  loss agree to 1e-9)
    for lo, hi, dl, tol in ((0, dec_lo, 3e-4, 1e-2), (dec_lo, g.numel(), 1e-2, 1e-3)):
        v = torch.zeros_like(g)
        v[lo:hi] = g[lo:hi] / g[lo:hi].norm()
        slope = float((g.double() * v.double()).sum())
        h = dl / max(abs(slope), 1e-3)
        fd = (loss_at(h * v) - loss_at(-h * v)) / (2 * h)
        assert abs(fd - slope) <= tol * abs(slope), (fd, slope, h)
    flat.copy_(theta)
    # (3) the round-1 kernel set on the same batch
    try:
        _switch(PM_GCL_FUSED=0)
        old = hip_fullsize_step(DENSE_SHARD_B64, lr=0.0)
    finally:
        _switch(PM_GCL_FUSED=None)
    m = old["info"]["launches"]
    assert m["gcl_fwd"] == 0 and m["gcl_dagg"] == 0 and m["planesB_nn"] == 16 and m["planes_tn"] == 16, old["info"]
    for k in ("pitch", "dur", "structure", "kld"):
        assert abs(old["losses"][k] - run["losses"][k]) <= 1e-6 * max(1.0, abs(run["losses"][k])), k
    assert _rel_l2(run["grads"], old["grads"], live) < 1e-2


@pytest.mark.parametrize("d", [256, 512])
def test_offset_limit_takes_the_round1_kernels_and_agrees(d):
    """The kernels of gcl.hip / linear.hip / wide.hip address their operands with 32-bit byte offsets; batches whose
    operands would not fit (N > ~349 k nodes at d = 256) take the round-1 kernels (`gcl_fits`, vae_step.hip).  The limit
    is lowered here (`PM_GCL_OFFSET_LIMIT`) so that a small batch triggers that fallback: the step then runs the
    segment-reduce + grouped-product kernels and returns the same losses, outputs and (up to ReLU kinks) gradients."""
    spec = dict(B=24, nb=2, d=d, L=2, p=0.25, dense=False, msg_p=0.1, seed=7)
    a, b = _same_step_up_to_relu_kinks(spec, dict(PM_GCL_OFFSET_LIMIT=1 << 20))
    assert a["info"]["launches"]["gcl_fwd"] == 4 and a["info"]["launches"]["rows_w"] == 2, a["info"]
    m = b["info"]["launches"]
    assert m["gcl_fwd"] == 0 and m["gcl_dagg"] == 0 and m["gcl_dw"] == 0 and m["rows_w"] == 0, b["info"]
    assert m["planesB_nn"] == 4 and m["planesB_nt"] == 4 and m["planes_tn"] == 4 and m["segreduce_fwd"] == 4, b["info"]
