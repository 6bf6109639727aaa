#!/usr/bin/env python3
"""Development helper: build extra copies of libpolyphemus_hip.so with compile-time switches for same-box A/B runs.

    python tools/build_variants.py gemm.hip abl1=-DPM_ABL=1 abl4=-DPM_ABL=4 ...
    PM_LIB_PATH=polyphemus_amd/variants/libpm_abl1.so python bench.py ...

Only the named translation unit is recompiled per variant; the other objects come from the regular build."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from polyphemus_amd import build as B

def main():
    B.build_library()
    tu = sys.argv[1]
    out_dir = os.path.join(B.HERE, "variants")
    os.makedirs(out_dir, exist_ok=True)
    from concurrent.futures import ThreadPoolExecutor
    def one(spec):
        tag, flags = spec.split("=", 1)
        obj = os.path.join(out_dir, f"{tu[:-4]}_{tag}.o")
        cmd = [B._hipcc(), *B.FLAGS, *flags.split(","), "-c", os.path.join(B.CSRC, tu), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise SystemExit(r.stderr)
        objs = [obj if f == tu[:-4] + ".o" else os.path.join(B.OBJ, f) for f in sorted(os.listdir(B.OBJ)) if f.endswith(".o")]
        lib = os.path.join(out_dir, f"libpm_{tag}.so")
        r = subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, "-ldl"], capture_output=True, text=True)
        if r.returncode:
            raise SystemExit(r.stderr)
        return lib
    with ThreadPoolExecutor(6) as ex:
        for lib in ex.map(one, sys.argv[2:]):
            print(lib)

if __name__ == "__main__":
    main()
