#!/usr/bin/env python3
"""Development helper: time `pm_unembed_dh` (input gradient of the three un-embeddings, csrc/unembed.hip) alone at the bench
batch, and — with a library built by `python tools/build_variants.py unembed.hip log=-DDH_LOG` and
PM_LIB_PATH=polyphemus_amd/variants/libpm_log.so — print workgroup 100's phase times (realtime ticks -> us):
consumer [tile start, products done, stores issued, barrier passed], producer groups [turn start, image written, loads issued].

    python tools/unembed_dh_bench.py [d ...]        (default 256 512 128)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphemus_amd.synthetic import synthetic_batch
from polyphemus_amd import ops
from polyphemus_amd._lib import lib, call, ptr, stream
DEV, S, B = "cuda:0", 5, 256

def main():
    cpu = synthetic_batch(B, 2, seed=1234)
    b = cpu.to(DEV)
    plan = ops.plan_build(b.edge_index, b.edge_type, b.edge_dist, b.bars, b.batch, b.is_drum, b.tokens, b.n_bars, b.s_tensor.shape[0], n_slots=S)
    N = cpu.num_nodes
    for d in [int(x) for x in sys.argv[1:]] or [256, 512, 128]:
        dh = d // 2
        dl = torch.randn(N, S, 230, device=DEV)
        Wd, Wn, Wu = torch.randn(131, dh, device=DEV) / 11, torch.randn(131, dh, device=DEV) / 11, torch.randn(99, dh, device=DEV) / 10
        scratch = torch.empty(int(lib().pm_unembed_dh_scratch_bytes(d)), dtype=torch.uint8, device=DEV)
        dH = torch.empty(N, S, d, device=DEV)
        def run(prep):
            call("pm_unembed_dh", ptr(dl), ptr(Wd), ptr(Wn), ptr(Wu), ptr(plan.buf), N, plan.E, plan.G, d, S, ptr(dH), ptr(scratch), prep, stream())
        run(1)
        for _ in range(5):
            run(0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run(0)
        e1.record(); torch.cuda.synchronize()
        print(f"d={d} rows={N * S}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per launch", flush=True)
        L = lib()
        if hasattr(L, "pm_debug_read_dhlog"):
            L.pm_debug_read_dhlog.argtypes, L.pm_debug_read_dhlog.restype = [ctypes.c_void_p], ctypes.c_int
            buf = (ctypes.c_longlong * (3 * 32 * 4))()
            assert L.pm_debug_read_dhlog(ctypes.cast(buf, ctypes.c_void_p)) == 0
            a = np.array(buf).reshape(3, 32, 4)
            t0 = a[a > 0].min()
            for who, name in enumerate(("consumer", "producer group 0", "producer group 1")):
                print(" ", name)
                for i in range(32):
                    if a[who, i, 0] > 0:
                        print("   ", i, [(int(x - t0) / 100.0 if x > 0 else None) for x in a[who, i]])

if __name__ == "__main__":
    main()
