#!/usr/bin/env python3
"""Full-size parity report (GPU box): the native HIP step at BASELINE.json's configurations against oracle/vae_cpu.py in
fp32 and fp64.  Prints one JSON object per configuration; tests/test_fullsize_gpu.py asserts on the same numbers.

    python tools/fullsize_parity.py [--deterministic] [name ...] > gpurun_out/r2/fullsize_parity.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from util import FULLSIZE, hip_vs_oracle_fullsize   # noqa: E402

if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    from bench import host_cores
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    det = "--deterministic" in sys.argv                        # the HIP step in deterministic mode (bit-reproducible realisation)
    if det:
        from polyphemus_amd import _lib
        _lib.set_deterministic(True)
    for name in (args or list(FULLSIZE)):
        rep = hip_vs_oracle_fullsize(FULLSIZE[name], threads=host_cores())
        print(json.dumps({"config": name, "deterministic": det, **rep}), flush=True)
