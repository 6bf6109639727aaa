"""Host graph construction == the reference's `graph_from_tensor` (data.py:141-204),
bit-exact (integer work), on structure.json and seeded / corner-case structures."""
import os

import numpy as np
import pytest

from polyphemus_amd import constants as C
from polyphemus_amd.graphs import graph_from_structure, collate_samples
from polyphemus_amd.synthetic import disk_sample, sample_from_disk, synthetic_batch
from util import GOLDEN, load_case, batch_from_golden

Z = np.load(os.path.join(GOLDEN, "graphs.npz"))


@pytest.mark.parametrize("name", [str(n) for n in Z["names"]])
def test_graph_matches_reference(name):
    s = Z[f"{name}/s"].astype(bool).copy()
    g = graph_from_structure(s)
    ei = Z[f"{name}/edge_index"]
    assert g["num_nodes"] == int(Z[f"{name}/num_nodes"])
    np.testing.assert_array_equal(g["src"], ei[0])
    np.testing.assert_array_equal(g["dst"], ei[1])
    np.testing.assert_array_equal(g["etype"], Z[f"{name}/etype"])
    np.testing.assert_array_equal(g["edist"], Z[f"{name}/edist"])
    np.testing.assert_array_equal(g["bars"], Z[f"{name}/bars"])
    np.testing.assert_array_equal(g["is_drum"], Z[f"{name}/is_drum"].astype(bool))


def test_structure_json_known_answers():
    """N=30, E=105, per-type counts, nodes per bar, drum nodes, distances."""
    name = "structure_json"
    g = graph_from_structure(Z[f"{name}/s"].astype(bool).copy())
    assert g["num_nodes"] == 30 and g["src"].shape[0] == 105
    assert np.bincount(g["etype"], minlength=6).tolist() == [22, 8, 8, 6, 36, 25]
    assert np.bincount(g["bars"]).tolist() == [10, 20]
    assert int(g["is_drum"].sum()) == 13
    hist = dict(zip(*np.unique(g["edist"], return_counts=True)))
    assert hist == {0: 36, 2: 18, 4: 30, 8: 7, 10: 4, 12: 2, 16: 4, 20: 2, 28: 2}


def test_empty_bar_is_mutated_in_place():
    s = np.zeros((1, 4, 32), bool)
    g = graph_from_structure(s)
    assert s[0, 0, 0] and g["num_nodes"] == 1            # data.py:152-153
    assert (g["src"], g["dst"], g["etype"], g["edist"]) == (0, 0, 0, 0)   # data.py:173-176


@pytest.mark.parametrize("case", ["lmd2_tiny", "nb3_tiny"])
def test_collate_matches_reference_dataloader(case):
    """disk samples -> my dataset/collate == reference PolyphemusDataset + PyG collate."""
    z, cfg = load_case(case)
    ref = batch_from_golden(z, cfg)
    nb = cfg["n_bars"]
    n = len([k for k in z.files if k.startswith("disk/") and k.endswith("/s_tensor")])
    samples = [sample_from_disk(z[f"disk/{i}/c_tensor"], z[f"disk/{i}/s_tensor"], nb) for i in range(n)]
    mine = collate_samples(samples, nb)
    for k in ("edge_index", "edge_type", "edge_dist", "tokens", "s_tensor", "is_drum", "bars", "batch"):
        assert getattr(mine, k).shape == getattr(ref, k).shape, k
        assert bool((getattr(mine, k) == getattr(ref, k)).all()), k
    assert mine.num_nodes == ref.num_nodes


def test_synthetic_statistics():
    """App. D: p=0.25 gives ~31.5 nodes and ~113 edges per bar, in-degree <= 8."""
    b = synthetic_batch(32, 2, p=0.25, seed=3)
    G = b.num_graphs
    assert 28 < b.num_nodes / G < 35
    assert 95 < b.edge_index.shape[1] / G < 130
    deg = np.bincount(b.edge_index[1].numpy(), minlength=b.num_nodes)
    assert deg.max() <= 8
    tok = b.tokens.numpy()
    assert (tok[:, 0, 0] == C.PITCH_SOS).all() and (tok[:, 0, 1] == C.DUR_SOS).all()
    assert ((tok[..., 0] == C.PITCH_EOS).sum(1) == 1).all()


def test_dense_stress_graph():
    b = synthetic_batch(2, 2, dense=True)
    assert b.num_nodes == 2 * 2 * 128 and b.edge_index.shape[1] == 4 * 128 * 127
    assert int(b.edge_dist.max()) == 31


@pytest.mark.parametrize("case", ["lmd2_tiny", "nb3_tiny"])
def test_dataset_relayout_matches_reference_getitem(case, tmp_path):
    """`PolyphemusDataset.__getitem__` (token ids bar by bar, data.py:226-233): the active cells' rows, in cell order,
    are the reference batch's token rows; no GPU involved."""
    import os
    from polyphemus_amd.data import PolyphemusDataset
    z, cfg = load_case(case)
    nb = cfg["n_bars"]
    n = len([k for k in z.files if k.startswith("disk/") and k.endswith("/s_tensor")])
    for i in range(n):
        np.savez(os.path.join(tmp_path, f"{i:04d}.npz"), c_tensor=z[f"disk/{i}/c_tensor"], s_tensor=z[f"disk/{i}/s_tensor"])
    ds = PolyphemusDataset(str(tmp_path), nb)
    assert len(ds) == n
    rows = []
    for i in range(n):
        tok, s = ds[i]
        assert tok.shape == (nb, 4, 32, 16, 2) and tok.dtype == np.int16 and s.shape == (nb, 4, 32) and s.dtype == np.uint8
        s = s.astype(bool)
        for b in range(nb):                                   # data.py:152-153: an empty bar gets cell [0,0]
            if not s[b].any():
                s[b, 0, 0] = True
        rows.append(tok.reshape(-1, 16, 2)[s.reshape(-1)])
    assert np.array_equal(np.concatenate(rows), z["in/tokens"])
