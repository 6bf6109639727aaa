#!/usr/bin/env python3
"""Full-size parity report (GPU box): the native HIP step at BASELINE.json's configurations against oracle/vae_cpu.py in
fp32 and fp64.  Prints one JSON object per configuration; tests/test_fullsize_gpu.py asserts on the same numbers.

    python tools/fullsize_parity.py [name ...] > gpurun_out/r2/fullsize_parity.json
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
    for name in (sys.argv[1:] or list(FULLSIZE)):
        rep = hip_vs_oracle_fullsize(FULLSIZE[name], threads=host_cores())
        print(json.dumps({"config": name, **rep}), flush=True)
