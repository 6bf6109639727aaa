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
