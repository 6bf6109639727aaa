#!/usr/bin/env python3
"""Golden-vector generator — ORACLE / TEST INFRASTRUCTURE, runs in the BUILD
CONTAINER ONLY (it needs /root/reference, which does not exist on the GPU box).

Imports the reference's own `model.py`, `data.py`, `training.py` *unchanged*
from /root/reference over the pure-torch PyG stand-in in `oracle/pyg_shim/`
(the real torch_geometric / torch_scatter / torch_sparse are not installed and
not vendored; see oracle/pyg_shim/README.md) and writes the captured tensors to
`tests/golden/*.npz`.  Only tensors are committed — no reference source travels.

What is captured:
  graphs.npz      reference `graph_from_tensor` (data.py:141-204) on structure.json
                  and on seeded random / corner-case structures -> edges, types, distances
  <case>.npz      a tiny VAE (reference `VAE(**cfg)` under torch.manual_seed(0)):
                  on-disk samples, the reference-built batch, the reference-keyed
                  state_dict, injected eps; eval-mode outputs; train-mode outputs
                  with GCL message dropout forced to 0 (public attribute, model.py:48);
                  the 7 loss values (training.py:298-347); gradients of every
                  parameter; buffers + parameters after 1 and 2 Adam steps
                  (train.py:181, training.py:152-172, lr from ExpDecayLRScheduler).

  <case>_generate.npz   the generation helpers (`_binary_from_logits`, `decoder(z, None)`, `mtp_from_logits`)

Usage:  python oracle/make_golden.py [generate|d128]     (from the repo root; `generate` rewrites only <case>_generate.npz,
        `d128` only the slim d = 128 case)
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REPO, "oracle", "pyg_shim"))

_cwd = os.getcwd()
os.chdir(REF)            # generation_config.py:14 opens its yaml relative to cwd
import data as ref_data          # noqa: E402
import model as ref_model        # noqa: E402
import training as ref_training  # noqa: E402
from torch_geometric.data import Batch  # noqa: E402  (the shim)
os.chdir(_cwd)

from polyphemus_amd import constants as C          # noqa: E402
from polyphemus_amd.synthetic import disk_sample   # noqa: E402

OUT = os.environ.get("PM_GOLDEN_OUT", os.path.join(REPO, "tests", "golden"))
torch.set_num_threads(1)


def graph_record(s):
    """reference graph_from_tensor on a [nb,4,32] bool tensor -> int arrays."""
    g = ref_data.graph_from_tensor(s.clone())
    return dict(
        s=s.numpy().astype(np.uint8),
        edge_index=g.edge_index.numpy().astype(np.int32),
        etype=g.edge_attrs[:, 0].numpy().astype(np.int8),
        edist=g.edge_attrs[:, 1:].argmax(1).numpy().astype(np.int8),
        bars=g.bars.numpy().astype(np.int8),
        is_drum=g.is_drum.numpy().astype(np.uint8),
        num_nodes=np.int32(int(g.num_nodes)))


def make_graphs():
    out = {}
    cases = {}
    cases["structure_json"] = torch.tensor(json.load(open(os.path.join(REF, "structure.json")))).bool()
    rng = np.random.default_rng(7)
    for p in (0.05, 0.1, 0.25, 0.5, 1.0):
        cases[f"bern_p{p}"] = torch.from_numpy(rng.random((2, 4, 32)) < p)
    cases["bern3_p0.25"] = torch.from_numpy(rng.random((3, 4, 32)) < 0.25)
    s = torch.zeros(2, 4, 32, dtype=torch.bool)          # bar 0 empty -> forced [0,0]; bar 1 single node track 2
    s[1, 2, 17] = True
    cases["empty_and_single"] = s
    s = torch.zeros(1, 4, 32, dtype=torch.bool)          # one timestep only -> next-edges early exit (data.py:97-98)
    s[0, :, 5] = True
    cases["one_timestep"] = s
    s = torch.zeros(1, 4, 32, dtype=torch.bool)          # one track only
    s[0, 3, ::3] = True
    cases["one_track"] = s
    for name, s in cases.items():
        for k, v in graph_record(s).items():
            out[f"{name}/{k}"] = v
    out["names"] = np.array(sorted(cases.keys()))
    np.savez_compressed(os.path.join(OUT, "graphs.npz"), **out)
    print("graphs.npz:", len(cases), "cases")


def build_batch(tmpdir, disk, n_bars):
    """Write samples in the reference's .npz layout, read them back through the
    reference's own PolyphemusDataset (data.py:207-271) and collate them the way
    the reference's DataLoader does."""
    for i, (c, s) in enumerate(disk):
        np.savez(os.path.join(tmpdir, f"{i:04d}.npz"), c_tensor=c, s_tensor=s)
    ds = ref_data.PolyphemusDataset(tmpdir, n_bars=n_bars)
    ds.files = sorted(ds.files, key=lambda e: e.name)
    return Batch.from_data_list([ds[i] for i in range(len(ds))])


def capture_case(name, cfg, batch_size, p, seed, corner=False, slim=False):
    rng = np.random.default_rng(seed)
    nb = cfg["n_bars"]
    disk = [disk_sample(rng, nb, p) for _ in range(batch_size)]
    if corner:
        c, s = disk[0]                                   # sample 0: bar 0 silent, bar 1 a single (non-drum) cell
        s[:] = False
        s[2, C.N_TIMESTEPS + 9] = True
        # bar 0 stays empty here: data.py:152-153 switches cell [0,0] on at load time
    out = {}
    for i, (c, s) in enumerate(disk):
        out[f"disk/{i}/c_tensor"] = c
        out[f"disk/{i}/s_tensor"] = s
    with tempfile.TemporaryDirectory() as td:
        graph = build_batch(td, disk, nb)

    out["in/edge_index"] = graph.edge_index.numpy().astype(np.int32)
    out["in/etype"] = graph.edge_attrs[:, 0].numpy().astype(np.int8)
    out["in/edist"] = graph.edge_attrs[:, 1:].argmax(1).numpy().astype(np.int8)
    out["in/tokens"] = np.stack([graph.c_tensor[..., :C.N_PITCH_TOKENS].argmax(-1).numpy(),
                                 graph.c_tensor[..., C.N_PITCH_TOKENS:].argmax(-1).numpy()], -1).astype(np.int16)
    assert float(graph.c_tensor.sum()) == graph.c_tensor.shape[0] * 32      # strictly one-hot pairs
    out["in/s_tensor"] = graph.s_tensor.numpy().astype(np.uint8)
    out["in/is_drum"] = graph.is_drum.numpy().astype(np.uint8)
    out["in/bars"] = graph.bars.numpy().astype(np.int16)
    out["in/batch"] = graph.batch.numpy().astype(np.int16)
    out["in/num_nodes"] = np.int32(int(graph.num_nodes))
    out["cfg"] = np.array(json.dumps(cfg))

    torch.manual_seed(0)
    vae = ref_model.VAE(**cfg, device=torch.device("cpu"))
    sd0 = {k: v.clone() for k, v in vae.state_dict().items()}
    if slim:      # the state is the reference's default init under torch.manual_seed(0), which the product's module tree
        import hashlib      # reproduces bit for bit (tests/test_model_cpu.py): store its digest instead of its 5 MB
        h = hashlib.sha256()
        for k, v in sd0.items():
            h.update(k.encode()); h.update(v.numpy().tobytes())
        out["sd_sha256"] = np.array(h.hexdigest())
    else:
        for k, v in sd0.items():
            out[f"sd/{k}"] = v.numpy()
    out["param_names"] = np.array([n for n, _ in vae.named_parameters()])
    B = batch_size
    eps = torch.from_numpy(np.random.default_rng(seed + 1).standard_normal((B, cfg["d"])).astype(np.float32))
    out["in/eps"] = eps.numpy()

    def fwd(model, g):
        mu, lv = model.encoder(g)                        # == model.py:668-676 with eps injected
        z = torch.exp(0.5 * lv) * eps + mu
        s_logits, c_logits = model.decoder(z, g)
        return s_logits, c_logits, mu, lv

    # ---- eval mode -----------------------------------------------------------
    vae.eval()
    with torch.no_grad():
        s_logits, c_logits, mu, lv = fwd(vae, graph)
    out["eval/s_logits"] = s_logits.numpy()
    if not slim:
        out["eval/c_logits"] = c_logits.numpy()
    out["eval/mu"], out["eval/log_var"] = mu.numpy(), lv.numpy()

    # ---- train mode, message dropout off, two optimizer steps ----------------
    vae.train()
    for m in vae.modules():
        if isinstance(m, ref_model.GCL):
            m.dropout = 0.0
    tj = json.load(open(os.path.join(REF, "training.json")))
    opt = torch.optim.Adam(vae.parameters(), **tj["optimizer"])          # train.py:181
    sched = ref_training.ExpDecayLRScheduler(optimizer=opt, **tj["lr_scheduler"])
    trainer = ref_training.PolyphemusTrainer("unused", vae, opt, lr_scheduler=sched)
    trainer.beta = 0                                                     # training.py:116
    out["opt"] = np.array(json.dumps(dict(optimizer=tj["optimizer"], lr_scheduler=tj["lr_scheduler"])))
    opt.zero_grad()
    for step in ((1,) if slim else (1, 2)):
        s_logits, c_logits, mu, lv = fwd(vae, graph)
        tot, losses = trainer._losses(graph.s_tensor, s_logits, graph.c_tensor, c_logits, mu, lv)
        tot.backward()                                                   # training.py:155
        pre = f"train{step}"
        out[f"{pre}/lr"] = np.float64(opt.param_groups[0]["lr"])
        if step == 1:
            out[f"{pre}/s_logits"], out[f"{pre}/c_logits"] = s_logits.detach().numpy(), c_logits.detach().numpy()
            out[f"{pre}/mu"], out[f"{pre}/log_var"] = mu.detach().numpy(), lv.detach().numpy()
            none = []
            for n, q in vae.named_parameters():
                if q.grad is None:
                    none.append(n)
                else:
                    out[f"{pre}/grad/{n}"] = q.grad.numpy().copy()
            out[f"{pre}/grad_none"] = np.array(none)
        out[f"{pre}/losses"] = np.array(json.dumps(losses))
        opt.step()                                                       # training.py:164
        opt.zero_grad()
        sched.step()                                                     # training.py:170
        for k, v in vae.state_dict().items():
            if not slim or "running_" in k or k.endswith("num_batches_tracked"):
                out[f"{pre}/sd_after/{k}"] = v.numpy().copy()
    # ---- evaluation metrics of the reference (training.py:349-497 `_accuracies`, `_losses` in eval mode) on the
    #      eval-mode outputs; kept in a separate small file
    vae.load_state_dict(sd0)
    vae.eval()
    with torch.no_grad():
        s_logits, c_logits, mu, lv = fwd(vae, graph)
        _, ev_losses = trainer._losses(graph.s_tensor, s_logits, graph.c_tensor, c_logits, mu, lv)
        accs = trainer._accuracies(graph.s_tensor, s_logits, graph.c_tensor, c_logits, graph.is_drum)
    np.savez_compressed(os.path.join(OUT, f"{name}_metrics.npz"), accs=np.array(json.dumps(accs)),
                        losses=np.array(json.dumps(ev_losses)))
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(f"{name}_metrics.npz: {accs}")
    print(f"{name}.npz: N={int(graph.num_nodes)} E={graph.edge_index.shape[1]} "
          f"params={sum(q.numel() for q in vae.parameters())} losses={losses}")


def capture_generation(name):
    """Generation helpers of the reference on the committed case `name` (its state_dict and inputs are read back
    from tests/golden/<name>.npz, so the big case files need not be regenerated):
      cond/*  `mtp_from_logits(eval c_logits, batch s_tensor)` (utils.py:59-79; generate.py:26-35 with s_tensor_cond)
      gen/*   `vae.decoder(z, None)` in eval mode (model.py:634-655: threshold -> host graph build -> content decoder),
              `_binary_from_logits` (model.py:609-623) and the resulting mtp.
    The [.,4,32,15,230] pianorolls are stored as a sha256 of their bytes plus per-cell sums; c_logits of the gen path as
    per-node sums, arg-max tokens and the first nodes' rows."""
    import hashlib
    os.chdir(REF)
    import utils as ref_utils
    os.chdir(_cwd)
    z0 = np.load(os.path.join(REPO, "tests", "golden", f"{name}.npz"))
    cfg = json.loads(str(z0["cfg"]))
    vae = ref_model.VAE(**cfg, device=torch.device("cpu"))
    vae.load_state_dict({k[3:]: torch.from_numpy(z0[k]) for k in z0.files if k.startswith("sd/")})
    vae.eval()
    out = {}
    B, nb = z0["in/eps"].shape[0], cfg["n_bars"]
    s_cond = torch.from_numpy(z0["in/s_tensor"]).view(B, nb, 4, 32).float()
    mtp = ref_utils.mtp_from_logits(torch.from_numpy(z0["eval/c_logits"]), s_cond)
    out["cond/mtp_sha256"] = np.array(hashlib.sha256(mtp.numpy().tobytes()).hexdigest())
    out["cond/mtp_cellsum"] = mtp.double().sum(dim=(-1, -2)).numpy()
    zs = torch.from_numpy(z0["in/eps"]) * 3.0                       # a wider latent sample: more varied structures
    with torch.no_grad():
        s_logits, c_logits = vae.decoder(zs, None)
        s_bin = vae.decoder._binary_from_logits(s_logits)
        mtp = ref_utils.mtp_from_logits(c_logits, s_bin)
    out["gen/z"], out["gen/s_logits"] = zs.numpy(), s_logits.numpy()
    out["gen/s_binary"] = s_bin.numpy().astype(np.uint8)
    out["gen/num_nodes"] = np.int32(c_logits.shape[0])
    out["gen/c_logits_head"] = c_logits[:4].numpy()
    out["gen/c_logits_nodesum"] = c_logits.double().sum(dim=(-1, -2)).numpy()
    out["gen/c_argmax"] = np.stack([c_logits[..., :C.N_PITCH_TOKENS].argmax(-1).numpy(),
                                    c_logits[..., C.N_PITCH_TOKENS:].argmax(-1).numpy()], -1).astype(np.int16)
    out["gen/mtp_cellsum"] = mtp.double().sum(dim=(-1, -2)).numpy()
    # threshold corner cases: logits around 0 (sigmoid rounds to exactly 0.5 for tiny negative logits), +-inf, an empty bar
    corner = torch.zeros(1, 4, 4, 32)
    corner[0, 0] = torch.linspace(-2e-7, 2e-7, 128).view(4, 32)
    corner[0, 1] = -5.0                                              # empty bar -> [0,0] switched on
    corner[0, 2] = torch.randn(4, 32, generator=torch.Generator().manual_seed(5)) * 1e-3
    corner[0, 3, 0, :4] = torch.tensor([float("inf"), float("-inf"), 88.0, -104.0])
    out["corner/s_logits"] = corner.numpy()
    out["corner/s_binary"] = vae.decoder._binary_from_logits(corner).numpy().astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, f"{name}_generate.npz"), **out)
    print(f"{name}_generate.npz: gen nodes={int(c_logits.shape[0])} active={int(s_bin.sum())} "
          f"corner on={int(out['corner/s_binary'].sum())}")


def capture_bnoff():
    """A model built with batch_norm = False (model.py:176-188: no norm layers in the GCNs; :218-238,278-292: CNNs without
    BatchNorm2d, so the Sequential indices of the second convolutions shift): state_dict keys and arithmetic of that
    constructor switch, pinned to the reference."""
    capture_case("bnoff_tiny", dict(dropout=0, batch_norm=False, gnn_n_layers=2, d=16, n_bars=2, resolution=8),
                 batch_size=5, p=0.08, seed=14)


def capture_d128():
    """d = 128 (a multiple of 128: the GCL forward and input-gradient products of the native step take the B-direct
    planes GEMM, the weight gradient the planes TN kernel) — the variant bench.py measures, pinned to the reference."""
    capture_case("d128_l2", dict(dropout=0, batch_norm=True, gnn_n_layers=2, d=128, n_bars=2, resolution=8),
                 batch_size=4, p=0.2, seed=13, slim=True)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:] == ["d128"]:
        capture_d128()
        sys.exit(0)
    if sys.argv[1:] == ["bnoff"]:
        capture_bnoff()
        sys.exit(0)
    if sys.argv[1:] == ["generate"]:
        capture_generation("lmd2_tiny")
        capture_generation("nb3_tiny")
        sys.exit(0)
    make_graphs()
    capture_case("lmd2_tiny", dict(dropout=0, batch_norm=True, gnn_n_layers=2, d=32, n_bars=2, resolution=8),
                 batch_size=8, p=0.06, seed=11, corner=True)
    capture_case("nb3_tiny", dict(dropout=0, batch_norm=True, gnn_n_layers=1, d=16, n_bars=3, resolution=8),
                 batch_size=6, p=0.05, seed=12)
    capture_generation("lmd2_tiny")
    capture_generation("nb3_tiny")
    capture_d128()
    capture_bnoff()

