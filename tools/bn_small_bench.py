"""Stand-alone timing of the one-launch small-batch norms (pm_bn_small_fwd / _bwd, B = 256 rows): 200 back-to-back calls.

    python tools/bn_small_bench.py         # on the GPU box; numbers in profiles/LOG.md, round 6
"""
import torch, sys
sys.path.insert(0, ".")
from polyphemus_amd._lib import call, ptr, stream
dev = "cuda"
for O, C in ((256, 512), (256, 256)):
    x = torch.randn(O, C, device=dev); dy = torch.randn(O, C, device=dev)
    mean = x.mean(0).contiguous(); var = x.var(0, unbiased=False).contiguous()
    ga = torch.ones(C, device=dev); be = torch.zeros(C, device=dev)
    dga = torch.zeros(C, device=dev); dbe = torch.zeros(C, device=dev); dx = torch.empty_like(x)
    y = torch.empty_like(x); rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev)
    def bwd():
        call("pm_bn_small_bwd", ptr(x), ptr(dy), O, C, ptr(mean), ptr(var), 1e-5, ptr(ga), ptr(be), 1, ptr(dga), ptr(dbe), None, ptr(dx), stream())
    def fwd():
        call("pm_bn_small_fwd", ptr(x), O, C, 1e-5, ptr(ga), ptr(be), None, 1, ptr(y), ptr(mean), ptr(var), ptr(rm), ptr(rv), 0.1, stream())
    for name, f in (("bwd", bwd), ("fwd", fwd)):
        for _ in range(10): f()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): f()
        e1.record(); torch.cuda.synchronize()
        print(f"O={O} C={C} {name}: {e0.elapsed_time(e1) / 200 * 1e3:.2f} us per call (back to back)")
