"""Host-side checks of the drop-in nn.Module surface (no GPU): state_dict contract
(SURVEY App. C), default-initialisation parity with the reference under the same seed,
flat parameter storage, checkpoint round trip, and the loud failure without a GPU."""
import numpy as np
import pytest
import torch

from polyphemus_amd.model import VAE
from polyphemus_amd._lib import HipExtensionError
from util import batch_from_golden, load_case, state_dict_from_golden


@pytest.mark.parametrize("case", ["lmd2_tiny", "nb3_tiny", "bnoff_tiny"])
def test_state_dict_keys_shapes_and_init_match_reference(case):
    z, cfg = load_case(case)
    ref = state_dict_from_golden(z)
    torch.manual_seed(0)
    vae = VAE(**cfg, device=torch.device("cpu"))
    sd = vae.state_dict()
    assert list(sd.keys()) == list(ref.keys())                     # same keys in the same order
    for k in ref:
        assert sd[k].shape == ref[k].shape and sd[k].dtype == ref[k].dtype, k
        assert torch.equal(sd[k], ref[k]), f"default init differs from the reference at {k}"
    assert [n for n, _ in vae.named_parameters()] == [str(n) for n in z["param_names"]]


def test_parameter_counts_match_survey():
    vae = VAE(dropout=0, batch_norm=True, gnn_n_layers=8, d=256, n_bars=2, resolution=8, device="cpu")
    assert len(vae.state_dict()) == 255
    assert len(list(vae.parameters())) == 152
    assert sum(p.numel() for p in vae.parameters()) == 10_752_477


def test_parameters_are_views_of_one_flat_buffer_and_survive_load():
    z, cfg = load_case("lmd2_tiny")
    vae = VAE(**cfg, device="cpu")
    flat = vae.flat_params
    for n, p in vae.named_parameters():
        assert p.data_ptr() == flat.data_ptr() + 4 * vae._offsets[n]
    w = vae.encoder.c_encoder.graph_encoder.layers[1]
    assert w.root.data_ptr() == w.weight.data_ptr() + 4 * w.weight.numel()      # [W_0..W_5 | root] contiguous
    sd = state_dict_from_golden(z)
    vae.load_state_dict(sd)
    for n, p in vae.named_parameters():
        assert p.data_ptr() == flat.data_ptr() + 4 * vae._offsets[n]
        assert torch.equal(p.detach(), sd[n])
    # the shared edge network is ONE tensor under every layer key
    l = vae.decoder.c_decoder.graph_decoder.layers
    assert l[0].nn.weight is l[1].nn.weight


def test_forward_fails_loudly_without_gpu():
    z, cfg = load_case("lmd2_tiny")
    vae = VAE(**cfg, device="cpu")
    g = batch_from_golden(z, cfg)
    with pytest.raises(HipExtensionError):
        vae(g)


def test_binary_from_logits_and_structure_from_binary():
    z, cfg = load_case("lmd2_tiny")
    vae = VAE(**cfg, device="cpu")
    s = torch.full((2, 2, 4, 32), -3.0)
    s[0, 0, 1, 5] = 2.0
    b = vae.decoder._binary_from_logits(s)
    assert b[0, 0].sum() == 1 and b[0, 0, 1, 5] and b[1, 1, 0, 0] and b[1, 1].sum() == 1
    g = vae.decoder._structure_from_binary(b)
    assert g.num_nodes == 4 and g.edge_index.shape[1] == 4 and int(g.bars.max()) == 1


def test_dropout_seed_keeps_a_base_seed_and_salts_it_per_rank():
    """the model keeps its BASE seed (what a checkpoint stores); the rank only salts the seeds derived from it.
    Two ranks draw different streams, re-applying the salt is idempotent (a second trainer on the same model), and a
    base seed restored from a rank-0 checkpoint still gives every rank its own stream."""
    from polyphemus_amd.model import VAE
    cfg = dict(dropout=0, batch_norm=True, gnn_n_layers=2, d=32, n_bars=2, resolution=8)
    vae = VAE(**cfg, device=torch.device("cpu"))
    base = vae.seed
    salts = [(0x9E3779B9 * (r + 1)) & 0xFFFFFFFF for r in range(2)]
    seen = []
    for salt in salts + [salts[0]]:
        vae.rank_salt, vae._step = salt, 0
        seen.append([vae._next_seed() for _ in range(3)])
        assert vae.seed == base                                   # never rewritten
    assert seen[0] != seen[1] and seen[0] == seen[2]
    assert len(set(seen[0] + seen[1])) == 6
