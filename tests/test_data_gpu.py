"""Dataset + device collate (polyphemus_amd/data.py) against the batch the reference's own PolyphemusDataset + PyG
DataLoader built from the same on-disk samples (captured in tests/golden/<case>.npz as `disk/*` and `in/*`): integer
work, everything bit-exact.  Then a training step fed by the loader."""
import os

import numpy as np
import pytest
import torch

from polyphemus_amd.data import DeviceLoader, PolyphemusDataset, collate_on_device, relayout_sample
from polyphemus_amd.model import VAE
from polyphemus_amd.synthetic import disk_sample, sample_from_disk
from polyphemus_amd.graphs import collate_samples
from polyphemus_amd.trainer import HipTrainer
from util import batch_from_golden, load_case

pytestmark = pytest.mark.gpu
DEV = "cuda"
KEYS = ("edge_index", "edge_type", "edge_dist", "tokens", "s_tensor", "is_drum", "bars", "batch")


def write_disk(tmp_path, samples):
    for i, (c, s) in enumerate(samples):
        np.savez(os.path.join(tmp_path, f"{i:04d}.npz"), c_tensor=c, s_tensor=s)


def assert_same(dev, ref):
    assert dev.num_nodes == ref.num_nodes
    for k in KEYS:
        a, b = getattr(dev, k).cpu(), getattr(ref, k)
        assert a.shape == b.shape and bool((a == b).all()), k


@pytest.mark.parametrize("case", ["lmd2_tiny", "nb3_tiny"])
def test_device_collate_equals_reference_dataloader(case, tmp_path):
    z, cfg = load_case(case)
    nb = cfg["n_bars"]
    n = len([k for k in z.files if k.startswith("disk/") and k.endswith("/s_tensor")])
    write_disk(tmp_path, [(z[f"disk/{i}/c_tensor"], z[f"disk/{i}/s_tensor"]) for i in range(n)])
    ds = PolyphemusDataset(str(tmp_path), nb)
    assert len(ds) == n
    ref = batch_from_golden(z, cfg)                                   # built by the reference (oracle/make_golden.py)
    assert_same(collate_on_device([ds[i] for i in range(n)], nb, DEV), ref)
    batches = list(DeviceLoader(ds, batch_size=n, device=DEV))
    assert len(batches) == 1
    assert_same(batches[0], ref)


def test_loader_batches_shuffle_and_partial_batches(tmp_path):
    rng = np.random.default_rng(5)
    disk = [disk_sample(rng, 2, 0.2) for _ in range(11)]
    disk[3][1][:, :32] = False                                        # sample 3: an empty first bar
    write_disk(tmp_path, disk)
    ds = PolyphemusDataset(str(tmp_path), 2)
    host = lambda idx: collate_samples([sample_from_disk(disk[i][0], disk[i][1].copy(), 2) for i in idx], 2)
    for workers in (0, 3):
        loader = DeviceLoader(ds, batch_size=4, device=DEV, num_workers=workers)
        got = list(loader)
        assert len(loader) == 3 and [b.s_tensor.shape[0] for b in got] == [8, 8, 6]
        for j, b in enumerate(got):
            assert_same(b, host(range(4 * j, min(4 * j + 4, 11))))
    assert len(DeviceLoader(ds, batch_size=4, device=DEV, drop_last=True)) == 2
    sh = DeviceLoader(ds, batch_size=4, shuffle=True, seed=7, device=DEV)
    order = np.random.default_rng(7).permutation(11)
    for j, b in enumerate(sh):
        assert_same(b, host(order[4 * j:4 * j + 4]))
    order2 = np.random.default_rng(8).permutation(11)                 # next epoch, next permutation
    assert_same(next(iter(sh)), host(order2[:4]))


def test_relayout_rejects_wrong_shapes():
    c, s = disk_sample(np.random.default_rng(0), 2, 0.2)
    with pytest.raises(ValueError):
        relayout_sample(c, s, 3)
    with pytest.raises(ValueError):
        relayout_sample(c[:, :-1], s, 2)
    with pytest.raises(ValueError):
        collate_on_device([], 2, DEV)


def test_training_from_the_loader_equals_training_from_host_batches(tmp_path):
    rng = np.random.default_rng(9)
    disk = [disk_sample(rng, 2, 0.25) for _ in range(12)]
    write_disk(tmp_path, disk)
    cfg = dict(dropout=0, batch_norm=True, gnn_n_layers=2, d=32, n_bars=2, resolution=8)
    epss = [torch.randn(6, 32, generator=torch.Generator().manual_seed(i)).to(DEV) for i in range(2)]
    out = []
    for use_loader in (True, False):
        torch.manual_seed(0)
        vae = VAE(**cfg, device=DEV).to(DEV)
        vae.train()
        tr = HipTrainer(vae, lr=1e-4)
        if use_loader:
            batches = DeviceLoader(PolyphemusDataset(str(tmp_path), 2), batch_size=6, device=DEV)
        else:
            batches = [collate_samples([sample_from_disk(*disk[i], 2) for i in range(6 * j, 6 * j + 6)], 2).to(DEV)
                       for j in range(2)]
        losses = [tr.losses_dict(tr.train_step(b, e)) for b, e in zip(batches, epss)]
        out.append((losses, vae.flat_params.clone()))
    (la, pa), (lb, pb) = out
    for step, (x, y) in enumerate(zip(la, lb)):                       # (the second step sees Adam-amplified atomics noise)
        for k in x:
            assert abs(x[k] - y[k]) <= (1e-6, 1e-4)[step] * max(1.0, abs(y[k])), (step, k)
    # two Adam updates at lr = 1e-4: rounding-noise gradients may move an element by +-lr per update on either side
    assert float((pa - pb).abs().max()) <= 4.5e-4 and float((pa - pb).abs().mean()) < 4e-6
