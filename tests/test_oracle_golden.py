"""Pin the CPU restatement (oracle/vae_cpu.py) to the vectors captured from the
reference's own model.py / training.py (oracle/make_golden.py): eval outputs, train
outputs, the 7 losses, every gradient, BN running stats and Adam-updated parameters."""
import json

import numpy as np
import pytest
import torch

from oracle import vae_cpu
from util import REL_TOL, batch_from_golden, load_case, rel_err, state_dict_from_golden

CASES = ["lmd2_tiny", "nb3_tiny", "bnoff_tiny"]


@pytest.fixture(autouse=True)
def _single_thread():
    """The goldens were captured with one CPU thread; reductions are then bit-reproducible
    (multi-threaded sums differ in the last bits, which tiny-batch BatchNorm amplifies)."""
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


@pytest.mark.parametrize("case", CASES)
def test_eval_forward(case):
    z, cfg = load_case(case)
    g = batch_from_golden(z, cfg)
    P, _ = vae_cpu.split_state(state_dict_from_golden(z), [str(n) for n in z["param_names"]])
    with torch.no_grad():
        (s_logits, c_logits), mu, lv = vae_cpu.vae_forward(g, P, cfg, False, torch.from_numpy(z["in/eps"]))
    for name, got in (("s_logits", s_logits), ("c_logits", c_logits), ("mu", mu), ("log_var", lv)):
        assert rel_err(got, z[f"eval/{name}"]) < 1e-6, name
    # eval mode leaves every buffer untouched
    for k, v in state_dict_from_golden(z).items():
        assert torch.equal(P[k].detach(), v), k


@pytest.mark.parametrize("case", CASES)
def test_two_train_steps(case):
    z, cfg = load_case(case)
    g = batch_from_golden(z, cfg)
    names = [str(n) for n in z["param_names"]]
    P, names = vae_cpu.split_state(state_dict_from_golden(z), names)
    optcfg = json.loads(str(z["opt"]))
    opt = torch.optim.Adam([P[n] for n in names], **optcfg["optimizer"])
    eps = torch.from_numpy(z["in/eps"])
    none = set(str(n) for n in z["train1/grad_none"])
    # SURVEY B-1: the structure decoder gets no gradient in the reference
    assert none == {n for n in names if n.startswith("decoder.s_decoder.")}
    for step in (1, 2):
        assert abs(opt.param_groups[0]["lr"] - float(z[f"train{step}/lr"])) < 1e-15
        outs, parts, grads = vae_cpu.train_step(g, P, names, cfg, opt, eps, msg_dropout=0.0)
        want = json.loads(str(z[f"train{step}/losses"]))
        for k, v in want.items():
            assert abs(float(parts[k].detach()) - v) <= 1e-5 * max(1.0, abs(v)), (step, k)   # kld cancels: 1+lv-e^lv
        if step == 1:
            for name, got in zip(("s_logits", "c_logits", "mu", "log_var"), outs):
                assert rel_err(got.detach(), z[f"train1/{name}"]) < 1e-6, name
            for n in names:
                if n in none:
                    assert grads[n] is None, n
                else:
                    assert rel_err(grads[n], z[f"train1/grad/{n}"]) < 1e-5, n
        lr = vae_cpu.exp_decay_lr(step, **optcfg["lr_scheduler"])          # training.py:169-170
        for pg in opt.param_groups:
            pg["lr"] = lr
        after = state_dict_from_golden(z, f"train{step}/sd_after/")
        for k, v in after.items():
            got = P[k].detach()
            if v.dtype.is_floating_point:
                assert rel_err(got, v) < 2e-6, (step, k)
            else:
                assert torch.equal(got, v), (step, k)


def test_train_step_d128_slim_case():
    """The d = 128 case (fixture without the 5 MB state_dict: its sha256 pins the default init): one train step."""
    z, cfg = load_case("d128_l2")
    g = batch_from_golden(z, cfg)
    names = [str(n) for n in z["param_names"]]
    P, names = vae_cpu.split_state(state_dict_from_golden(z), names)
    optcfg = json.loads(str(z["opt"]))
    opt = torch.optim.Adam([P[n] for n in names], **optcfg["optimizer"])
    outs, parts, grads = vae_cpu.train_step(g, P, names, cfg, opt, torch.from_numpy(z["in/eps"]), msg_dropout=0.0)
    for k, v in json.loads(str(z["train1/losses"])).items():
        assert abs(float(parts[k].detach()) - v) <= 1e-5 * max(1.0, abs(v)), k
    for name, got in zip(("s_logits", "c_logits", "mu", "log_var"), outs):
        assert rel_err(got.detach(), z[f"train1/{name}"]) < 1e-6, name
    none = set(str(n) for n in z["train1/grad_none"])
    for n in names:
        if n not in none:
            assert rel_err(grads[n], z[f"train1/grad/{n}"]) < 1e-5, n
    for k, v in state_dict_from_golden(z, "train1/sd_after/").items():
        assert (rel_err(P[k].detach(), v) < 2e-6) if v.dtype.is_floating_point else torch.equal(P[k].detach(), v), k


def test_losses_quirk_structure_on_target():
    """training.py:307 replaces the logits by the target: the loss ignores s_logits."""
    z, cfg = load_case("lmd2_tiny")
    g = batch_from_golden(z, cfg)
    s_logits = torch.randn(8, 2, 4, 32, requires_grad=True)
    c_logits = torch.from_numpy(z["train1/c_logits"].copy())
    mu = torch.from_numpy(z["train1/mu"].copy())
    lv = torch.from_numpy(z["train1/log_var"].copy())
    tot, parts = vae_cpu.losses(g.s_tensor, s_logits, g.c_tensor, c_logits, mu, lv)
    want = json.loads(str(z["train1/losses"]))
    assert abs(float(parts["structure"]) - want["structure"]) < 1e-6
    assert not tot.requires_grad


def test_exp_decay_lr_schedule():
    kw = dict(peak_lr=1e-4, warmup_steps=8000, final_lr_scale=0.01, decay_steps=800000)
    assert vae_cpu.exp_decay_lr(1, **kw) == 1e-4 and vae_cpu.exp_decay_lr(8000, **kw) == 1e-4
    assert abs(vae_cpu.exp_decay_lr(808000, **kw) - 1e-6) < 1e-12


def test_binary_from_logits_forces_empty_bars():
    s = torch.full((2, 2, 4, 32), -3.0)
    s[0, 0, 1, 5] = 2.0
    b = vae_cpu.binary_from_logits(s)
    assert b[0, 0].sum() == 1 and b[0, 0, 1, 5]
    for i, j in ((0, 1), (1, 0), (1, 1)):
        assert b[i, j].sum() == 1 and b[i, j, 0, 0]


@pytest.mark.parametrize("case", ["lmd2_tiny", "nb3_tiny"])
def test_oracle_accuracies_match_reference(case):
    """oracle.accuracies == the reference's `_accuracies` (training.py:349-497) on the captured eval-mode outputs."""
    import json
    import os
    import numpy as np
    z, cfg = load_case(case)
    m = np.load(os.path.join(os.path.dirname(__file__), "golden", f"{case}_metrics.npz"), allow_pickle=True)
    want = json.loads(str(m["accs"]))
    g = batch_from_golden(z, cfg)
    got = vae_cpu.accuracies(g.s_tensor, torch.from_numpy(z["eval/s_logits"]), g.c_tensor,
                             torch.from_numpy(z["eval/c_logits"]), g.is_drum)
    assert set(got) == set(want)
    for k, v in want.items():
        assert abs(got[k] - v) < 1e-7, (k, got[k], v)


def _generate_golden(case):
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", f"{case}_generate.npz"), allow_pickle=False)


@pytest.mark.parametrize("case", ["lmd2_tiny", "nb3_tiny"])
def test_generation_helpers_match_reference(case):
    """oracle.binary_from_logits / mtp_from_logits / decoder_forward on a thresholded structure == the reference's
    `_binary_from_logits` (model.py:609-623), `mtp_from_logits` (utils.py:59-79) and `decoder(z, None)`
    (model.py:634-655), captured by `oracle/make_golden.py generate`."""
    import hashlib
    from polyphemus_amd.graphs import collate_samples, graph_from_structure
    z, cfg = load_case(case)
    gg = _generate_golden(case)
    # thresholding, including logits around 0, +-inf and an empty bar
    for k in ("gen", "corner"):
        got = vae_cpu.binary_from_logits(torch.from_numpy(gg[f"{k}/s_logits"]))
        assert np.array_equal(got.numpy().astype(np.uint8), gg[f"{k}/s_binary"]), k
    # the pianoroll laid out on the batch's own structure (generate.py:26-35 with s_tensor_cond): exact bytes
    B, nb = z["in/eps"].shape[0], cfg["n_bars"]
    s_cond = torch.from_numpy(z["in/s_tensor"]).view(B, nb, 4, 32)
    mtp = vae_cpu.mtp_from_logits(torch.from_numpy(z["eval/c_logits"]), s_cond)
    assert hashlib.sha256(mtp.numpy().tobytes()).hexdigest() == str(gg["cond/mtp_sha256"])
    assert np.array_equal(mtp.double().sum(dim=(-1, -2)).numpy(), gg["cond/mtp_cellsum"])
    with pytest.raises((RuntimeError, IndexError, ValueError)):
        vae_cpu.mtp_from_logits(torch.from_numpy(z["eval/c_logits"][:-1]), s_cond)
    # decoder(z, None): structure from the thresholded logits, then the content decoder on the graphs built from it
    s_bin = gg["gen/s_binary"].astype(bool)
    samples = []
    for i in range(B):
        g = graph_from_structure(s_bin[i])
        g["tokens"] = np.zeros((g["num_nodes"], 16, 2), np.int32)
        g["s_tensor"] = s_bin[i].astype(np.float32)
        samples.append(g)
    graph = collate_samples(samples, nb)
    assert graph.num_nodes == int(gg["gen/num_nodes"])
    P, _ = vae_cpu.split_state(state_dict_from_golden(z), [str(n) for n in z["param_names"]])
    with torch.no_grad():
        s_logits, c_logits = vae_cpu.decoder_forward(torch.from_numpy(gg["gen/z"]), graph, P, cfg, False)
    assert rel_err(s_logits, gg["gen/s_logits"]) < 1e-6
    assert rel_err(c_logits[:4], gg["gen/c_logits_head"]) < 1e-5
    assert rel_err(c_logits.double().sum(dim=(-1, -2)), gg["gen/c_logits_nodesum"]) < 1e-5
    mtp = vae_cpu.mtp_from_logits(c_logits, torch.from_numpy(s_bin))
    assert rel_err(mtp.double().sum(dim=(-1, -2)), gg["gen/mtp_cellsum"]) < 1e-5
