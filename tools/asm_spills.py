"""Where a kernel's scratch (spill) instructions sit relative to its MFMA loop (hipcc -S --cuda-device-only output).

    python tools/asm_spills.py file.s [name-substring]
"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
starts.append((len(lines), "end"))
for (a, name), (b, _) in zip(starts, starts[1:]):
    if pat not in name:
        continue
    body = lines[a:b]
    sc = [k for k, l in enumerate(body) if "scratch_" in l]
    mf = [k for k, l in enumerate(body) if "v_mfma" in l]
    if not mf:
        continue
    # loop body = between labels that enclose most MFMAs
    inl = [k for k in sc if mf[0] <= k <= mf[-1]]
    print(f"{name[:60]:60s} scratch {len(sc):4d}  inside mfma span {len(inl):4d}  mfma {len(mf):4d}  span {mf[0]}..{mf[-1]} of {len(body)}")
    if "-v" in sys.argv:
        for k in sc:
            print("   ", k, body[k].strip())
