"""BASELINE.json's configurations at their FULL sizes (C2 10^6; C3, C4, C5 10^7 pairs, generated on the device), checked
through properties that do not need the oracle to run the whole batch:
  * window parity — the oracle on windows of the same seeded batch, spread over its whole range (the generator is random
    access: pair i depends on (seed, i) only), against the same slice of the full-size result;
  * shard invariance — the batch cut into 4 contiguous shards (the multi-GPU partition) gives the same penalties;
  * counters — the device accuracy counters equal a host recount of the downloaded arrays, and total = n;
  * metric facts for unit costs — |n-m| <= NW <= number of injected edit operations; Greedy >= NW wherever the reference
    defines Greedy on the whole string; penalties of an error-free batch are all zero;
  * determinism — a second run reproduces the first bit for bit."""
import numpy as np
import pytest

from tests.util import greedy_defined, leap_defined

pytestmark = pytest.mark.gpu

FULL = [("C2", 1_000_000), ("C3", 10_000_000), ("C4", 10_000_000), ("C5", 10_000_000)]
WINDOW = 2000


def _lengths(asm, cfg, first, n):
    ro = np.zeros(n + 1, np.uint32)
    fo = np.zeros(n + 1, np.uint32)
    import ctypes
    rc = asm.load_library().asm_generate_pairs(ctypes.byref(cfg), first, n, ro.ctypes.data, fo.ctypes.data, None, 0, None, 0)
    assert rc == 0
    return np.diff(ro).astype(np.int64), np.diff(fo).astype(np.int64)


def _run(asm, engine, cfg, first, n, params):
    batch = engine.generate(cfg, first, n, asm.GREEDY_CLEAN)
    d = [engine.malloc(4 * n) for _ in range(3)]
    d_cnt = engine.malloc(32)
    engine.memset_async(d_cnt, 0, 32)
    engine.run_benchmark_async(batch, params, d[0], d[1], d[2], d_cnt, repack=True)
    out = [engine.to_host(x, n) for x in d]
    cnt = engine.to_host(d_cnt, 8).view(np.uint64)[:4].copy()
    for x in d + [d_cnt]:
        engine.free(x)
    batch.free()
    return out, cnt


@pytest.mark.parametrize("wl,n", FULL)
def test_full_size_configuration(asm, engine, oracle, wl, n):
    cfg, n_cfg, params = asm.workload(wl)
    assert n_cfg == n, "BASELINE.json size"
    (nw, leap, greedy), cnt = _run(asm, engine, cfg, 0, n, params)

    # window parity against the oracle, 6 windows over the whole range
    for first in np.linspace(0, n - WINDOW, 6).astype(np.int64):
        first = int(first)
        hb = asm.generate_pairs(cfg, first, WINDOW)
        sl = slice(first, first + WINDOW)
        gd, ld = greedy_defined(hb, params.k), leap_defined(hb)
        assert np.array_equal(nw[sl], oracle.nw(hb)), (wl, first, "nw")
        assert np.array_equal(leap[sl][ld], oracle.leap(hb, k=params.k)[ld]), (wl, first, "leap")
        assert np.array_equal(greedy[sl][gd], oracle.greedy(hb, k=params.k, mode=1)[gd]), (wl, first, "greedy")

    # counters = host recount (benchmark_utils.h:249-255)
    assert cnt.tolist() == [n, n, int((leap == nw).sum()), int((greedy == nw).sum())], (wl, cnt)

    # shard invariance: 4 contiguous shards, generated and run on their own
    q = n // 4
    for r in range(4):
        lo, hi = r * q, (n if r == 3 else (r + 1) * q)
        (s_nw, s_leap, s_greedy), s_cnt = _run(asm, engine, cfg, lo, hi - lo, params)
        assert np.array_equal(s_nw, nw[lo:hi]) and np.array_equal(s_leap, leap[lo:hi]) and np.array_equal(s_greedy, greedy[lo:hi]), (wl, r)
        assert int(s_cnt[0]) == hi - lo

    # metric facts (x = o = e = 1)
    m, nn = _lengths(asm, cfg, 0, n)
    assert (nw >= np.abs(nn - m)).all() and (nw <= np.maximum(m, nn)).all()
    if cfg.kind == asm.GEN_EXACT_ERRORS:
        # the generator applies ceil(L * err) edit operations to a length-L pattern, L <= max(m, n); +1 for float rounding
        ops = np.ceil(np.maximum(m, nn).astype(np.float64) * float(cfg.err)).astype(np.int64) + 1
        assert (nw <= ops).all(), (wl, int((nw > ops).sum()))
    whole = (np.maximum(m, nn) <= 128) & (np.abs(nn - m) <= params.k)  # Greedy sees the whole pair and its destination lane is in the band
    assert (greedy[whole] >= nw[whole]).all(), (wl, int((greedy[whole] < nw[whole]).sum()))
    assert ((leap >= 0) | (leap == -1)).all()

    # determinism
    (nw2, leap2, greedy2), cnt2 = _run(asm, engine, cfg, 0, n, params)
    assert np.array_equal(nw, nw2) and np.array_equal(leap, leap2) and np.array_equal(greedy, greedy2) and np.array_equal(cnt, cnt2)


def test_error_free_batch_at_full_size(asm, engine):
    """10^7 identical pairs: every penalty is 0, every aligner agrees with NW, the filter passes everything with ED 0."""
    n = 10_000_000
    cfg = asm.GenConfig.exact(77, 100, 0.0)
    (nw, leap, greedy), cnt = _run(asm, engine, cfg, 0, n, asm.Params.default(k=3))
    assert not nw.any() and not leap.any() and not greedy.any()
    assert cnt.tolist() == [n, n, n, n]
    batch = engine.generate(cfg, 0, n, asm.GREEDY_CLEAN)
    assert not engine.simd_ed(batch, 3, True, asm.FILTER_CLEAN).any()
    assert engine.shd_filter(batch, 3).all()
    batch.free()
