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


def _tile_weight(tc, grp, m0, rows, use_classes):
    if not use_classes:
        return 4
    cb = tc[8 + 5 * grp: 13 + 5 * grp]
    return 2 + int(m0 < cb[3] and m0 + rows > cb[1]) + int(m0 < cb[4] and m0 + rows > cb[2])


def _random_trk_cnt(rng, scale):
    tc = [0] * 32
    for g in range(4):
        c = [0 if rng.random() < 0.2 else int(rng.integers(0, scale)) for _ in range(4)]
        tc[g] = sum(c)
        tc[8 + 5 * g: 13 + 5 * g] = [0, c[0], c[0] + c[1], c[0] + c[1] + c[2], sum(c)]
    return tc


def _bench_like_trk_cnt(rng, n_nodes):
    """class sizes as the synthetic LMD2 batches have them (p = 0.25): per track ~8 % / 11 % / 47 % / 34 % of the nodes in
    the classes (0,0) (1,0) (1,1) (0,1)"""
    tc = [0] * 32
    per = rng.multinomial(n_nodes, [0.25] * 4)
    for g in range(4):
        c = rng.multinomial(int(per[g]), [0.08, 0.11, 0.47, 0.34])
        tc[g] = int(per[g])
        tc[8 + 5 * g: 13 + 5 * g] = [0, int(c[0]), int(c[0] + c[1]), int(c[0] + c[1] + c[2]), int(per[g])]
    return tc


def _check_cover(tc, order, use_classes):
    """every row of every track group in exactly one live workgroup; halves are 32 rows at a multiple of 32"""
    live = [(b, g, m0, rows) for b, (g, m0, rows) in enumerate(order) if g >= 0]
    seen = [[0] * tc[g] for g in range(4)]
    for _, g, m0, rows in live:
        assert rows in (32, 64) and m0 % 32 == 0 and (rows == 32 or m0 % 64 == 0) and m0 < tc[g], (g, m0, rows)
        for r in range(m0, min(m0 + rows, tc[g])):
            seen[g][r] += 1
    assert all(v == 1 for g in range(4) for v in seen[g])
    return live


@pytest.mark.parametrize("use_classes", [False, True])
def test_gcl_tile_schedule_covers_every_row_once_heaviest_first(use_classes):
    """csrc/tile_order.h (host copy of the function the GCL kernels run): every row of every track group exactly once,
    every XCD (workgroup index mod 8) the same number of tiles within one (a split tile counts once), its heavy tiles
    (4 blocks of K) before its light ones, whole tiles before halves, no live workgroup behind one that exits."""
    import numpy as np
    rng = np.random.default_rng(3)
    for trial in range(300):
        tc = _random_trk_cnt(rng, (40, 300, 3000, 20000)[trial % 4]) if trial % 3 else _bench_like_trk_cnt(rng, int(rng.integers(15800, 17100)))
        N = sum(tc[:4])
        order = _lib.gcl_tile_order(tc, use_classes, N)
        assert len(order) % 8 == 0
        live = _check_cover(tc, order, use_classes)
        per_xcd = [[(_tile_weight(tc, g, m0, rows, use_classes), rows) for b, g, m0, rows in live if b % 8 == x] for x in range(8)]
        tiles = [sum(1.0 if rows == 64 else 0.5 for _, rows in w) for w in per_xcd]
        assert max(tiles) - min(tiles) <= 1.0
        ntiles = sum((tc[g] + 63) // 64 for g in range(4))
        rounds = (ntiles - 1) // 256 if ntiles else 0
        for w in per_xcd:
            whole = [a for a, rows in w if rows == 64]
            if any(rows == 32 for _, rows in w):     # an XCD with two halves: its whole 2-block tiles at positions 28, 29, ..
                assert len(whole) == 32 and whole[:28] == sorted(whole[:28], reverse=True) and whole[28] == whole[29] == 2
                assert sorted(whole[28:]) == sorted(sorted(whole)[:4])
            elif whole != sorted(whole, reverse=True):
                # several rounds, an XCD with a tile more than `rounds` per CU: a chain of its lightest tiles (all 2-block)
                # at positions 28, 32, 64, .., 32 rounds; the other positions in longest-first order
                assert use_classes and rounds >= 2 and len(whole) == 32 * rounds + 1
                chain = [28] + [32 * j for j in range(1, rounds + 1)]
                assert all(whole[p] == 2 for p in chain)
                rest = [a for p, a in enumerate(whole) if p not in chain]
                assert rest == sorted(rest, reverse=True) and min(rest) >= 2
            assert [rows for _, rows in w] == sorted((rows for _, rows in w), reverse=True)      # halves last
        for x in range(8):
            seq = [g >= 0 for g, _, _ in order[x::8]]
            if not any(rows == 32 for _, _, rows in order[x::8]):        # (an empty second half may sit before a live one)
                assert seq == sorted(seq, reverse=True)


def test_gcl_tile_schedule_absorbs_a_few_tiles_more_than_cus():
    """Batches of the bench shape have 255..262 tiles for 256 CUs.  Model of the hardware as measured (per-workgroup clocks,
    profiles/LOG.md): workgroup b goes to XCD b % 8; an XCD deals its workgroups in order to its four shader engines in
    turn (8 CUs each) and one that finds its engine full holds back those behind it; a tile costs 14 + 13.5 per block
    of K in us, a 32-row half 14 + 6.75 per block (the shape of k_gcl_fwd).  Up to 6 tiles more than CUs the launch lasts no
    longer than 1.03 x its heaviest tile, where plain longest-first order needs 1.2 x."""
    import heapq
    import numpy as np
    rng = np.random.default_rng(11)
    seen_split = 0
    for trial in range(200):
        tc = _bench_like_trk_cnt(rng, int(rng.integers(16250, 16700)))
        N = sum(tc[:4])
        order = _lib.gcl_tile_order(tc, True, N)
        live = _check_cover(tc, order, True)
        ntiles = sum((tc[g] + 63) // 64 for g in range(4))
        cost = lambda g, m0, rows: 14.0 + (13.5 if rows == 64 else 6.75) * _tile_weight(tc, g, m0, rows, True)
        span = 0.0
        for x in range(8):
            engines = [[0.0] * 8 for _ in range(4)]
            for e in engines:
                heapq.heapify(e)
            issued = 0.0                                   # in-order: no workgroup starts before the one in front of it
            for k, (g, m0, rows) in enumerate(order[x::8]):
                e = engines[k % 4]
                start = max(heapq.heappop(e), issued)
                issued = start
                t = start + (cost(g, m0, rows) if g >= 0 else 0.0)
                span = max(span, t)
                heapq.heappush(e, t)
        if 256 < ntiles <= 262:
            seen_split += 1
            assert any(rows == 32 for _, _, _, rows in live), ntiles
            assert span <= 1.03 * 68.0, (ntiles, span)
        elif ntiles <= 256:
            assert not any(rows == 32 for _, _, _, rows in live)
            assert span <= 68.0 + 1e-9
    assert seen_split >= 40


def test_gcl_tile_schedule_absorbs_extra_tiles_of_a_two_round_batch():
    """LMD16-shaped batches (B = 64 x 16 bars) have 508..518 tiles: two rounds over the 256 CUs and a few tiles more.  Same
    hardware model as above: the launch lasts no longer than two of its heaviest tiles plus 3 %, where the plain
    longest-first order pays a third, short round (+11 % measured on k_gcl_fwd)."""
    import heapq
    import numpy as np
    rng = np.random.default_rng(12)
    seen = 0
    for trial in range(120):
        tc = _bench_like_trk_cnt(rng, int(rng.integers(32500, 33100)))
        N = sum(tc[:4])
        order = _lib.gcl_tile_order(tc, True, N)
        live = _check_cover(tc, order, True)
        ntiles = sum((tc[g] + 63) // 64 for g in range(4))
        cost = lambda g, m0, rows: 14.0 + (13.5 if rows == 64 else 6.75) * _tile_weight(tc, g, m0, rows, True)
        span = 0.0
        for x in range(8):
            engines = [[0.0] * 8 for _ in range(4)]
            issued = 0.0
            for k, (g, m0, rows) in enumerate(order[x::8]):
                e = engines[k % 4]
                heapq.heapify(e)
                start = max(heapq.heappop(e), issued)
                issued = start
                t = start + (cost(g, m0, rows) if g >= 0 else 0.0)
                span = max(span, t)
                heapq.heappush(e, t)
        if 512 < ntiles <= 518:
            seen += 1
            assert span <= 1.03 * 2 * 68.0, (ntiles, span)
        elif ntiles <= 512:
            assert span <= 2 * 68.0 + 1e-9, (ntiles, span)
    assert seen >= 25


def test_uniform_row_tiles_cover_every_row_once():
    """`pm_row_tile` (chord products: uniform 64-row tiles): every row in exactly one workgroup; with 257..264 tiles an XCD
    that has 33 runs its last tile as two halves at positions 32 and 33 of its list."""
    import ctypes
    L = _lib.lib()
    for M in (1, 63, 64, 65, 5000, 16271, 16384, 16385, 16417, 16550, 16896, 16897, 20000, 32768, 32869, 33100, 33300, 40000):
        grid = L.pm_row_tile_order(M, None, 0)
        out = (ctypes.c_int32 * (2 * grid))()
        assert L.pm_row_tile_order(M, ctypes.cast(out, ctypes.c_void_p), grid) == grid and grid % 8 == 0
        seen = [0] * M
        ntile = (M + 63) // 64
        halves = 0
        for b in range(grid):
            m0, rows = out[2 * b], out[2 * b + 1]
            if rows == 0:
                continue
            assert rows in (32, 64) and 0 <= m0 < M
            halves += rows == 32
            if rows == 32:
                assert (b // 8) % 32 in (0, 1)
            for r in range(m0, min(M, m0 + rows)):
                seen[r] += 1
        assert all(v == 1 for v in seen), M
        assert (halves > 0) == (0 < (ntile - 1) % 256 + 1 <= 8 and ntile > 256), (M, halves)


def test_plan_scratch_field_holds_what_the_plan_build_carves():
    """pm_plan_build carves cursors [7N], drum positions [N + 1], the tile sums of its four scans (+ 64), node classes
    [N] and class histograms [cdiv(N, 256)][16] out of the plan's last field; pm_plan_offsets must reserve at least that
    (round 3 grew the carve-out without the reservation: tiny N wrote past the plan buffer)."""
    cdiv = lambda a, b: (a + b - 1) // b
    for N, G in [(n, g) for n in list(range(1, 10)) + [255, 256, 257, 4097, 100000] for g in (1, 2, 7, 5000, 200000)]:
        off = _lib.plan_layout(N, max(N, 1), G)
        have = off[-1] - off[_lib.PLAN_FIELDS.index("scratch")]
        tiles = cdiv(6 * N + 1, 2048) + 2 * cdiv(N + 1, 2048) + cdiv(G + 1, 2048)
        need = 7 * N + (N + 1) + tiles + 64 + N + 16 * cdiv(N, 256)
        assert have >= need, (N, G, have, need)
