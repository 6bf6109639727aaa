"""C-ABI checks that need no GPU: the shared library loads, exports every symbol that
include/polyphemus_hip.h declares, and the ctypes signature table matches the header."""
import ctypes
import os
import re

import pytest

from polyphemus_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "polyphemus_hip.h")


def header_prototypes():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"\b(int|int64_t|uint32_t|const char\*)\s+(pm_\w+)\s*\(([^;{]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        codes = ""
        for a in [x.strip() for x in args.split(",")]:
            if a in ("void", ""):
                continue
            if "*" in a:
                codes += "p"
            elif a.startswith("pm_stream_t"):
                codes += "s"
            elif a.startswith("int64_t"):
                codes += "l"
            elif a.startswith("uint32_t"):
                codes += "u"
            elif a.startswith("float"):
                codes += "f"
            elif a.startswith("double"):
                codes += "D"
            elif a.startswith("int32_t") or a.startswith("int "):
                codes += "i"
            else:
                raise AssertionError(f"unparsed argument {a!r} of {name}")
        protos[name] = codes
    return protos


def test_library_loads_and_exports_every_declared_symbol():
    L = _lib.lib()
    protos = header_prototypes()
    assert len(protos) >= 35
    for name in protos:
        assert hasattr(L, name), f"{name} declared in the header but not exported"
    assert L.pm_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define PM_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert b"gfx950" in L.pm_build_info()


def test_ctypes_signatures_match_header():
    protos = header_prototypes()
    for name, sig in _lib._SIGS.items():
        assert name in protos, name
        assert protos[name] == sig, f"{name}: header {protos[name]} != binding {sig}"
    for name in protos:
        assert name in _lib.EXPORTED, f"{name} not bound"


def test_plan_layout_host_only():
    off = _lib.plan_layout(100, 400, 8)
    assert len(off) == len(_lib.PLAN_FIELDS) + 1
    assert off[0] == 0 and all(b > a for a, b in zip(off, off[1:])) and all(o % 4 == 0 for o in off)
    assert off[1] - off[0] >= 100 * 6 + 1


def test_dropout_hash_is_host_callable_and_stable():
    L = _lib.lib()
    a = L.pm_dropout_hash(1234, 3, 17, 5)
    assert a == L.pm_dropout_hash(1234, 3, 17, 5) and 0 <= a < (1 << 24)
    assert a != L.pm_dropout_hash(1234, 3, 17, 6)


def test_dropout_stream_statistics_and_numpy_replica():
    """The counter-based dropout stream (common.h: one 32-bit mix per (edge, group of four channels), the four channels
    of a group = that word times four odd constants): the numpy restatement the oracle replays equals the library's
    host entry bit for bit; the keep rate is 1 - p; the four channels of a group, neighbouring groups and neighbouring
    edges drop independently (joint drop rate = p^2 within sampling error)."""
    import numpy as np
    from util import dropout_keep_np
    L = _lib.lib()
    p, d = 0.1, 64
    eids = np.arange(20000)
    keep = dropout_keep_np(77, 3, eids, d, p).astype(bool)                       # [E, d]
    thr = int(np.float32(p) * np.float32(16777216.0))
    for e in (0, 1, 4097, 19999):
        for c in (0, 1, 2, 3, 4, 7, 62, 63):
            assert (L.pm_dropout_hash(77, 3, int(e), c) >= thr) == bool(keep[e, c])
    drop = ~keep
    n = drop.size
    assert abs(drop.mean() - p) < 4 * (p * (1 - p) / n) ** 0.5
    sig2 = 4 * (p * p * (1 - p * p) / (drop.shape[0] * (d // 4))) ** 0.5          # 4 sigma of a joint rate
    for a in range(4):
        for b in range(a + 1, 4):                                                 # channels a, b of the same group
            assert abs((drop[:, a::4] & drop[:, b::4]).mean() - p * p) < sig2, (a, b)
    assert abs((drop[:, :-4] & drop[:, 4:]).mean() - p * p) < sig2                # same lane, next group
    assert abs((drop[:-1] & drop[1:]).mean() - p * p) < sig2                      # same channel, next edge
    other = dropout_keep_np(77, 4, eids, d, p).astype(bool)                       # next layer: another stream
    assert abs((drop & ~other).mean() - p * p) < sig2


def _tile_weight(tc, grp, t, use_classes):
    if not use_classes:
        return 4
    cb = tc[8 + 5 * grp: 13 + 5 * grp]
    m0 = t * 64
    return 2 + int(m0 < cb[3] and m0 + 64 > cb[1]) + int(m0 < cb[4] and m0 + 64 > cb[2])


@pytest.mark.parametrize("use_classes", [False, True])
def test_gcl_tile_schedule_covers_every_tile_once_heaviest_first(use_classes):
    """csrc/tile_order.h (host copy of the function the GCL kernels run): every (track group, tile) exactly once, every
    XCD (workgroup index mod 8) the same number of tiles within one, its heavy tiles (4 blocks of K) before its light
    ones — the workgroups beyond one per CU are the cheapest tiles."""
    import numpy as np
    rng = np.random.default_rng(3)
    for trial in range(300):
        tc = [0] * 32
        scale = (40, 300, 3000, 20000)[trial % 4]
        for g in range(4):
            c = [0 if rng.random() < 0.2 else int(rng.integers(0, scale)) for _ in range(4)]
            tc[g] = sum(c)
            tc[8 + 5 * g: 13 + 5 * g] = [0, c[0], c[0] + c[1], c[0] + c[1] + c[2], sum(c)]
        N = sum(tc[:4])
        order = _lib.gcl_tile_order(tc, use_classes, N)
        assert len(order) % 8 == 0
        live = [(b, g, t) for b, (g, t) in enumerate(order) if g >= 0]
        want = {(g, t) for g in range(4) for t in range((tc[g] + 63) // 64)}
        assert len(live) == len(want) and {(g, t) for _, g, t in live} == want
        per_xcd = [[_tile_weight(tc, g, t, use_classes) for b, g, t in live if b % 8 == x] for x in range(8)]
        assert max(map(len, per_xcd)) - min(map(len, per_xcd)) <= 1
        for w in per_xcd:
            assert w == sorted(w, reverse=True)
        # a workgroup that exits is never followed by a live one of the same XCD (the hardware deals them in order)
        for x in range(8):
            seq = [g >= 0 for g, _ in order[x::8]]
            assert seq == sorted(seq, reverse=True)
        # blocks of K per XCD: balanced to within one heavy tile
        tot = [sum(w) for w in per_xcd]
        assert max(tot) - min(tot) <= 4 + 2
