"""Generation helpers on the device (csrc/generate.hip, polyphemus_amd/generate.py) against the
reference's captured outputs (tests/golden/*_generate.npz) and the oracle: thresholding and the pianoroll layout are
byte / index work and must be bit-exact."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import vae_cpu
from polyphemus_amd import ops
from polyphemus_amd.generate import generate_music, generate_z
from polyphemus_amd.model import VAE
from util import GOLDEN, REL_TOL, batch_from_golden, load_case, rel_err, state_dict_from_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"
CASES = ["lmd2_tiny", "nb3_tiny"]


def _gen(case):
    return np.load(os.path.join(GOLDEN, f"{case}_generate.npz"), allow_pickle=False)


@pytest.mark.parametrize("case", CASES)
def test_binary_from_logits_matches_reference(case):
    gg = _gen(case)
    for k in ("gen", "corner"):                      # corner: logits around 0, +-inf, an empty bar
        got = ops.binary_from_logits(torch.from_numpy(gg[f"{k}/s_logits"]).to(DEV))
        assert got.dtype == torch.bool
        assert np.array_equal(got.cpu().numpy().astype(np.uint8), gg[f"{k}/s_binary"]), k


def test_binary_from_logits_threshold_nan_and_sizes():
    for G in (1, 3, 8, 129):
        x = torch.randn(G, 4, 32, generator=torch.Generator().manual_seed(G)) * 2 - 1.5
        x[0] = -9.0
        if G > 2:
            x[2, 1, 7] = float("nan")
        for th in (0.5, 0.3, 0.9):
            got = ops.binary_from_logits(x.to(DEV), th).cpu()
            assert torch.equal(got, vae_cpu.binary_from_logits(x.clone(), th)), (G, th)
        assert got[0].sum() == 1 and got[0, 0, 0]
    with pytest.raises(ValueError):
        ops.binary_from_logits(torch.zeros(2, 4, 31, device=DEV))


@pytest.mark.parametrize("case", CASES)
def test_mtp_from_logits_matches_reference_bytes(case):
    """The pianoroll laid out on the batch's own structure: the exact bytes of the reference's `mtp_from_logits`."""
    z, cfg = load_case(case)
    gg = _gen(case)
    B, nb = z["in/eps"].shape[0], cfg["n_bars"]
    s_cond = torch.from_numpy(z["in/s_tensor"]).view(B, nb, 4, 32).to(DEV)
    c_logits = torch.from_numpy(z["eval/c_logits"]).to(DEV)
    for s in (s_cond, s_cond.bool(), s_cond.float()):
        mtp = ops.mtp_from_logits(c_logits, s)
        assert mtp.shape == (B, nb, 4, 32, 15, 230)
        assert hashlib.sha256(mtp.cpu().numpy().tobytes()).hexdigest() == str(gg["cond/mtp_sha256"])
    with pytest.raises(ValueError):                   # the reference's masked assignment raises on a count mismatch
        ops.mtp_from_logits(c_logits[:-1].contiguous(), s_cond)
    with pytest.raises(ValueError):
        ops.mtp_from_logits(c_logits[:, :14].contiguous(), s_cond)


@pytest.mark.parametrize("G,p", [(1, 0.0), (1, 1.0), (5, 0.3), (64, 0.02), (33, 0.97)])
def test_mtp_from_logits_matches_oracle(G, p):
    gen = torch.Generator().manual_seed(G)
    s = (torch.rand(1, G, 4, 32, generator=gen) < p)
    N = int(s.sum())
    c = torch.randn(N, 15, 230, generator=gen)
    got = ops.mtp_from_logits(c.to(DEV), s.to(DEV)).cpu()
    assert torch.equal(got, vae_cpu.mtp_from_logits(c, s))


@pytest.mark.parametrize("case", CASES)
def test_generate_music_matches_reference(case):
    """generate.py:21-37 with s_cond = None: decoder -> thresholded structure -> device graph build -> content decoder
    -> pianoroll, against the reference's capture of the same call."""
    z, cfg = load_case(case)
    gg = _gen(case)
    vae = VAE(**cfg, device=DEV).to(DEV)
    vae.load_state_dict(state_dict_from_golden(z))
    vae.eval()
    zs = torch.from_numpy(gg["gen/z"]).to(DEV)
    with torch.no_grad():
        s_logits, c_logits = vae.decoder(zs, None)
        mtp, s_tensor = generate_music(vae, zs)
    assert rel_err(s_logits, gg["gen/s_logits"]) < REL_TOL
    want = torch.from_numpy(gg["gen/s_binary"]).bool()
    assert torch.equal(s_tensor.cpu(), want)          # (the smallest |logit| of the captures is 1e-5: no knife edges)
    assert c_logits.shape[0] == int(gg["gen/num_nodes"])
    assert rel_err(c_logits[:4], gg["gen/c_logits_head"]) < REL_TOL
    scale = float(np.abs(gg["gen/c_logits_head"]).max()) * 230 * 15
    assert float(np.abs(c_logits.double().sum(dim=(-1, -2)).cpu().numpy() - gg["gen/c_logits_nodesum"]).max()) < REL_TOL * scale
    tok = torch.stack([c_logits[..., :131].argmax(-1), c_logits[..., 131:].argmax(-1)], -1).cpu().numpy()
    assert (tok != gg["gen/c_argmax"]).mean() < 2e-3  # arg-max ties of an untrained model can flip at 1e-6
    assert float(np.abs(mtp.double().sum(dim=(-1, -2)).cpu().numpy() - gg["gen/mtp_cellsum"]).max()) < REL_TOL * scale
    assert torch.equal(mtp.cpu(), vae_cpu.mtp_from_logits(c_logits.cpu(), s_tensor.cpu()))
    assert mtp.shape == (*want.shape, 15, 230)


def test_generate_music_with_structure_conditioning():
    """generate.py:205-237: a given binary structure (repeated over the batch) conditions the content decoder."""
    z, cfg = load_case("lmd2_tiny")
    vae = VAE(**cfg, device=DEV).to(DEV)
    vae.load_state_dict(state_dict_from_golden(z))
    vae.eval()
    s_one = torch.zeros(cfg["n_bars"], 4, 32, dtype=torch.bool)
    s_one[0, 1, ::4] = True
    s_one[0, 0, 2] = True                              # bar 1 stays empty -> [0,0] switched on by _structure_from_binary
    s_tensor = s_one.unsqueeze(0).repeat(3, 1, 1, 1).to(DEV)
    zz = generate_z(3, cfg["d"], DEV)
    assert zz.shape == (3, cfg["d"]) and zz.is_cuda
    with torch.no_grad():
        graph = vae.decoder._structure_from_binary(s_tensor)
        mtp, s_out = generate_music(vae, zz, graph, s_tensor)
        _, c_logits = vae.decoder(zz, graph)
    assert s_out is s_tensor and bool(s_tensor[:, 1, 0, 0].all())
    assert torch.equal(mtp[s_tensor], c_logits)
    sil = mtp[~s_tensor]
    assert torch.equal(sil.argmax(-1)[:, 0], torch.full((sil.shape[0],), 129, device=DEV))
    assert torch.equal(sil.argmax(-1)[:, 1:], torch.full((sil.shape[0], 14), 130, device=DEV))
    assert float(sil.sum()) == 15.0 * sil.shape[0]


def test_mtp_from_logits_full_size_round_trip():
    """BASELINE configs[1] size (B=256, 2 bars, p=0.25): gather-back of the active cells returns the logits, the rest
    is the silence pattern — size-independent properties, no CPU copy of the 0.9 GB tensor."""
    gen = torch.Generator().manual_seed(3)
    s = (torch.rand(256, 2, 4, 32, generator=gen) < 0.25).to(DEV)
    N = int(s.sum())
    c = torch.randn(N, 15, 230, device=DEV)
    mtp = ops.mtp_from_logits(c, s)
    assert torch.equal(mtp[s], c)
    sil = mtp[~s]
    assert float(sil.sum()) == 15.0 * sil.shape[0]
    assert bool((sil[:, 0, 129] == 1).all()) and bool((sil[:, 1:, 130] == 1).all())
