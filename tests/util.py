"""Shared helpers for the parity tests: golden loading and the 1e-4 relative metric."""
import json
import os

import numpy as np
import torch

from polyphemus_amd.graphs import BarGraphBatch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REL_TOL = 1e-4          # BASELINE.json north_star: "outputs matching reference within 1e-4 rel"


def rel_err(a, b):
    """max|a-b| / max|b|  (SURVEY §8(d)); 0 when both are all-zero."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    den = float(b.abs().max()) if b.numel() else 0.0
    num = float((a - b).abs().max()) if b.numel() else 0.0
    return num / den if den > 0 else num


def load_case(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    cfg = json.loads(str(z["cfg"]))
    return z, cfg


def batch_from_golden(z, cfg) -> BarGraphBatch:
    """The reference-built batch of a golden case as a BarGraphBatch (CPU)."""
    from polyphemus_amd.graphs import batch_flags
    ei, et = torch.from_numpy(z["in/edge_index"].astype(np.int64)), torch.from_numpy(z["in/etype"].astype(np.int32))
    n_slots, unique = batch_flags(torch.from_numpy(z["in/tokens"].astype(np.int32)), ei, et, int(z["in/num_nodes"]))
    return BarGraphBatch(
        n_slots=n_slots, track_unique=unique,        # what `collate_samples` attaches: selects the measured native path
        edge_index=torch.from_numpy(z["in/edge_index"].astype(np.int64)),
        edge_type=torch.from_numpy(z["in/etype"].astype(np.int32)),
        edge_dist=torch.from_numpy(z["in/edist"].astype(np.int32)),
        tokens=torch.from_numpy(z["in/tokens"].astype(np.int32)),
        s_tensor=torch.from_numpy(z["in/s_tensor"].astype(np.float32)),
        is_drum=torch.from_numpy(z["in/is_drum"].astype(bool)),
        bars=torch.from_numpy(z["in/bars"].astype(np.int64)),
        batch=torch.from_numpy(z["in/batch"].astype(np.int64)),
        num_nodes=int(z["in/num_nodes"]), n_bars=cfg["n_bars"])


def state_dict_from_golden(z, prefix="sd/"):
    return {k[len(prefix):]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith(prefix)}


# ---- counter-based dropout stream of the HIP path, restated in numpy (csrc/common.h) ----
_M32 = np.uint64(0xFFFFFFFF)


def _mix32(x):
    x = x.astype(np.uint64) & _M32
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x


def dropout_keep_np(seed, layer_uid, eids, d, p):
    """keep[e, c] of pm_dropout_keep(seed, layer, eid, channel) for p in [0,1): float32 0/1."""
    eids = np.asarray(eids, dtype=np.uint64)
    k0 = _mix32(np.array([(seed & 0xFFFFFFFF) ^ (((layer_uid + 1) * 0x9E3779B9) & 0xFFFFFFFF)], np.uint64))[0]
    ek = _mix32(k0 ^ ((eids * np.uint64(0x85EBCA6B) + np.uint64(0x27D4EB2F)) & _M32))
    ch = (np.arange(d, dtype=np.uint64) * np.uint64(0xC2B2AE35)) & _M32
    h = _mix32((ek[:, None] + ch[None, :]) & _M32)
    thresh = np.uint64(int(np.float32(p) * np.float32(16777216.0)))
    return ((h >> np.uint64(8)) >= thresh).astype(np.float32)


def layer_uid_of(key: str) -> int:
    """GCL parameter prefix -> layer uid used by the HIP path (encoder GCN 0.., decoder GCN 1000..)."""
    base = 0 if key.startswith("encoder.") else 1000
    return base + int(key.rsplit(".", 1)[1])


# ---- multi-process harness of the data-parallel tests (CPU gloo and GPU) ----------------------------------
def _rank_main(fn, rank, world, port, q, dump_dir, hang_after, args):
    """Child entry: rendezvous on 127.0.0.1, run `fn(rank, world, *args)`, ship its result or its traceback."""
    import faulthandler
    import traceback
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), GLOO_SOCKET_IFNAME=os.environ.get("GLOO_SOCKET_IFNAME", "lo"))
    dump = open(os.path.join(dump_dir, f"rank{rank}.trace"), "w")
    faulthandler.enable(dump)
    faulthandler.dump_traceback_later(hang_after, exit=True, file=dump)     # a hung rank names its frame and dies
    try:
        q.put((rank, "ok", fn(rank, world, *args)))
    except BaseException:
        q.put((rank, "error", traceback.format_exc()))
    finally:
        faulthandler.cancel_dump_traceback_later()


class RanksHung(RuntimeError):
    pass


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(fn, world, args=(), timeout=90.0):
    """Run `fn(rank, world, *args)` in `world` fresh spawn processes (daemon, never re-exec'd) and return their results
    in rank order.  A rank that raises fails the caller with the rank's traceback; ranks that do not answer within
    `timeout` seconds are killed and RanksHung carries their faulthandler dumps.  Children never outlive the call."""
    import tempfile
    import time
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    dump_dir = tempfile.mkdtemp(prefix="pm_ranks_")
    procs = [ctx.Process(target=_rank_main, args=(fn, r, world, port, q, dump_dir, max(timeout - 10.0, 5.0), args),
                         daemon=True) for r in range(world)]
    res = {}
    try:
        for p in procs:
            p.start()
        deadline = time.time() + timeout
        while len(res) < world and time.time() < deadline:
            try:
                rank, status, payload = q.get(timeout=1.0)
            except Exception:
                if any(p.exitcode not in (None, 0) for p in procs) and q.empty():
                    time.sleep(0.5)
                    if q.empty():
                        break
                continue
            if status == "error":
                raise AssertionError(f"rank {rank} raised:\n{payload}")
            res[rank] = payload
        if len(res) < world:
            traces = []
            for r in range(world):
                try:
                    traces.append(f"--- rank {r} (exit code {procs[r].exitcode}) ---\n" +
                                  open(os.path.join(dump_dir, f"rank{r}.trace")).read()[-3000:])
                except OSError:
                    pass
            raise RanksHung(f"{world - len(res)} of {world} ranks gave no result within {timeout:.0f} s\n" + "\n".join(traces))
        for p in procs:
            p.join(timeout=20)
        bad = [(r, p.exitcode) for r, p in enumerate(procs) if p.exitcode not in (0, None)]
        assert not bad, f"ranks exited non-zero: {bad}"
        return [res[r] for r in range(world)]
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(timeout=5)
            if p.is_alive():
                p.kill()
                p.join(timeout=5)
