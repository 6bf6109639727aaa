#!/usr/bin/env python3
"""Data parallel vs single device (GPU box; two ranks share the GPU over gloo): gradient of a 2-rank step on 6 + 6 samples
against the single-device step on the 12 samples, with / without synchronised BatchNorm and the global token mean.
    python tools/dp_equivalence.py
Measured (round 2): sync_bn + global_token_mean 1.8e-7 of the largest gradient (rel. L2 7.8e-7); per-replica BatchNorm
2.2e-1 (8.0e-1); sync_bn without the token weighting 1.2e-2 (2.8e-2)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch, datetime
import test_zz_dp_gpu as T
from util import run_ranks

def worker(rank, world, sync, gtm):
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        from polyphemus_amd.graphs import collate_samples
        from polyphemus_amd.model import VAE
        from polyphemus_amd.trainer import HipTrainer
        torch.manual_seed(100)
        vae = VAE(**T.CFG, device="cuda").to("cuda"); vae.train(); vae.msg_dropout = 0.0
        tr = HipTrainer(vae, lr=1e-3, native=False, sync_bn=sync, global_token_mean=gtm)
        batch = collate_samples(T._sync_samples()[6 * rank:6 * rank + 6], 2).to("cuda")
        eps = torch.randn(12, T.CFG["d"], generator=torch.Generator().manual_seed(3))[6 * rank:6 * rank + 6].cuda()
        tr.train_step(batch, eps); torch.cuda.synchronize()
        return tr.grads.detach().cpu().numpy()
    finally:
        dist.destroy_process_group()

if __name__ == "__main__":
    from polyphemus_amd.graphs import collate_samples
    from polyphemus_amd.model import VAE
    from polyphemus_amd.trainer import HipTrainer
    torch.manual_seed(100)
    vae = VAE(**T.CFG, device="cuda").to("cuda"); vae.train(); vae.msg_dropout = 0.0
    tr = HipTrainer(vae, lr=1e-3, native=False)
    eps = torch.randn(12, T.CFG["d"], generator=torch.Generator().manual_seed(3)).cuda()
    tr.train_step(collate_samples(T._sync_samples(), 2).to("cuda"), eps)
    g1 = tr.grads.detach().cpu().double()
    for sync, gtm in ((True, True), (False, True), (True, False), (False, False)):
        g = torch.from_numpy(run_ranks(worker, 2, (sync, gtm), timeout=120)[0]).double() / 2
        print(f"sync_bn={sync} global_token_mean={gtm}: max err / gmax = {float((g - g1).abs().max() / g1.abs().max()):.2e}, rel L2 = {float(((g - g1) ** 2).sum().sqrt() / (g1 ** 2).sum().sqrt()):.2e}")
