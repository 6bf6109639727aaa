"""Build libpolyphemus_hip.so (gfx950) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container;
the resulting .so travels to the GPU box with the source snapshot (it is git-ignored).
Each .hip translation unit is compiled to an object in parallel, then linked.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "obj")
LIB = os.path.join(HERE, "libpolyphemus_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def _deps_mtime() -> float:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "polyphemus_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src: str, obj: str) -> str:
    cmd = [_hipcc(), *FLAGS, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
    return obj


def build_library(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdr_t = _deps_mtime()
    jobs, objs = [], []
    for f in srcs:
        src, obj = os.path.join(CSRC, f), os.path.join(OBJ, f[:-4] + ".o")
        objs.append(obj)
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t)
        if stale:
            jobs.append((src, obj))
    if jobs:
        if verbose:
            print(f"[build] hipcc: {len(jobs)} translation unit(s)", file=sys.stderr)
        with cf.ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(lambda a: _compile(*a), jobs))
    if jobs or not os.path.exists(LIB):
        r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
