"""Device-side bar-graph construction (csrc/graph.hip) against the host construction (polyphemus_amd/graphs.py),
which tests/test_graphs.py pins bit for bit to vectors captured from the reference's `graph_from_tensor`.
Integer work: everything must be identical, edge order included."""
import numpy as np
import pytest
import torch

from polyphemus_amd import constants as C
from polyphemus_amd.graphs import collate_samples, device_batch_from_structure, graph_from_structure
from util import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda"


def host_batch(s_all: np.ndarray, n_bars: int):
    """reference-order batch of B = G / n_bars samples built on the host"""
    samples = []
    for i in range(s_all.shape[0] // n_bars):
        s = s_all[i * n_bars:(i + 1) * n_bars].copy()
        rec = graph_from_structure(s)
        rec["s_tensor"] = s
        rec["tokens"] = np.zeros((rec["num_nodes"], 16, 2), np.int64)
        samples.append(rec)
    return collate_samples(samples, n_bars)


def assert_same(dev_b, host_b):
    assert dev_b.num_nodes == host_b.num_nodes
    assert torch.equal(dev_b.edge_index.cpu(), host_b.edge_index)
    assert torch.equal(dev_b.edge_type.cpu(), host_b.edge_type) and torch.equal(dev_b.edge_dist.cpu(), host_b.edge_dist)
    assert torch.equal(dev_b.bars.cpu(), host_b.bars) and torch.equal(dev_b.batch.cpu(), host_b.batch)
    assert torch.equal(dev_b.is_drum.cpu(), host_b.is_drum)
    assert torch.equal(dev_b.s_tensor.cpu(), host_b.s_tensor)          # incl. the in-place fix of empty bars


@pytest.mark.parametrize("p,n_bars,B", [(0.06, 2, 9), (0.25, 2, 16), (0.5, 3, 5), (0.9, 1, 7), (1.0, 2, 2)])
def test_device_graphs_equal_host_graphs(p, n_bars, B):
    rng = np.random.default_rng(int(p * 100) + n_bars)
    s = (rng.random((B * n_bars, 4, 32)) < p)
    s[0] = False                                                        # an empty bar -> cell [0,0]
    if B * n_bars > 3:
        s[1] = False; s[1, 2, 17] = True                                # a single cell -> self loop
        s[2] = False; s[2, :, 5] = True                                 # one chord column only
        s[3] = False; s[3, 1, :] = True                                 # one full track only
    host = host_batch(s.astype(np.float32), n_bars)
    dev = device_batch_from_structure(torch.from_numpy(s.astype(np.float32)).to(DEV), n_bars)
    assert_same(dev, host)


def test_device_graphs_on_the_reference_structures():
    """the structures whose graphs were captured from the reference itself (tests/golden/graphs.npz)"""
    import os
    Z = np.load(os.path.join(GOLDEN, "graphs.npz"))
    for name in (str(n) for n in Z["names"]):
        s = Z[f"{name}/s"].astype(np.float32)
        dev = device_batch_from_structure(torch.from_numpy(s).to(DEV), s.shape[0])
        assert dev.num_nodes == int(Z[f"{name}/num_nodes"]), name
        assert np.array_equal(dev.edge_index.cpu().numpy(), Z[f"{name}/edge_index"].astype(np.int64)), name
        assert np.array_equal(dev.edge_type.cpu().numpy(), Z[f"{name}/etype"].astype(np.int32)), name
        assert np.array_equal(dev.edge_dist.cpu().numpy(), Z[f"{name}/edist"].astype(np.int32)), name
        assert np.array_equal(dev.bars.cpu().numpy(), Z[f"{name}/bars"].astype(np.int64)), name
        assert np.array_equal(dev.is_drum.cpu().numpy(), Z[f"{name}/is_drum"].astype(bool)), name


def test_token_grid_gather():
    rng = np.random.default_rng(3)
    s = rng.random((4, 4, 32)) < 0.3
    grid = torch.from_numpy(rng.integers(0, 99, (4, 4, 32, 16, 2))).to(DEV)
    b = device_batch_from_structure(torch.from_numpy(s).to(DEV), 2, token_grid=grid)
    g, k, t = np.nonzero(s)                                            # (bar, track, timestep) order == node order
    assert torch.equal(b.tokens.cpu(), grid.cpu()[g, k, t].to(torch.int32))


def test_batch_flags_on_the_device_equal_the_torch_definition():
    """graphs.batch_flags — the active token slots, the compact-GCL premise, the id range check of a foreign batch — by
    csrc/plan.hip pm_batch_flags (two launches, one host read) against the torch op sequence on the CPU copy."""
    from polyphemus_amd.graphs import batch_flags
    from polyphemus_amd.synthetic import synthetic_batch
    for kw in (dict(seed=3), dict(seed=4, max_notes=9), dict(seed=5, max_notes=14)):
        b = synthetic_batch(6, 2, p=0.3, **kw)
        want = batch_flags(b.tokens, b.edge_index, b.edge_type, b.num_nodes)
        g = b.to(DEV)
        assert batch_flags(g.tokens, g.edge_index, g.edge_type, g.num_nodes) == want == (b.n_slots, b.track_unique)
    # a node that receives edges of two track relations: the compact GCL must be refused
    b = synthetic_batch(4, 2, p=0.3, seed=6)
    et = b.edge_type.clone()
    trk = torch.nonzero(et < 4).reshape(-1)
    dst = b.edge_index[1]
    victim = int(dst[trk[0]])
    other = [int(i) for i in trk if int(dst[i]) == victim]
    et[other[0]] = (int(et[other[0]]) + 1) % 4 if len(other) > 1 else et[other[0]]
    if len(other) > 1:
        assert batch_flags(b.tokens, b.edge_index, et, b.num_nodes)[1] is False
        assert batch_flags(b.tokens.to(DEV), b.edge_index.to(DEV), et.to(DEV), b.num_nodes)[1] is False
    bad = b.tokens.clone()
    bad[0, 3, 0] = 131
    with pytest.raises(ValueError):
        batch_flags(bad.to(DEV), b.edge_index.to(DEV), b.edge_type.to(DEV), b.num_nodes)
