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


@pytest.mark.parametrize("err", [0.05, 0.10, 0.15, 0.20])
def test_readme_accuracy_lines_from_the_device_counters(asm, engine, oracle, err):
    """The reference's README.md:16-20,32-36,47-51,63-67 — 10^6 simulated 100 bp pairs per error rate — re-run at that size
    through `_run_benchmark` on the device (Greedy in sequential mode: the reference as run), percentages from the device
    counters.  These four lines are all the reference holds about NW's results (parasail is absent): LEAP and Greedy accuracy
    are "penalty == NW penalty", so a wrong NW distance moves both.  The pairs are drawn the reference's way
    (oracle.reference_dataset = Dataset over glibc's rand(), byte-identical to the compiled reference generator): its
    dependent pattern characters shift these percentages by ~0.2 points at err >= 0.15 against independent ones.
    Tolerance: 4 binomial standard errors of the difference of two independent 10^6-pair samples.  Coverage (README's fourth
    line) is compared too, at 0.25 points: its NW traceback tie-break is this library's, parasail's is not pinned."""
    from tests.test_oracle_golden import README_ACCURACY, README_COVERAGE, readme_tolerance

    n = 1_000_000
    hb = asm.HostBatch(*oracle.reference_dataset(n, 100, err, seed=2000 + int(round(err * 100))))
    params = asm.Params.default(k=3)
    batch = engine.upload(hb, asm.GREEDY_SEQUENTIAL)
    d = [engine.malloc(4 * n) for _ in range(3)]
    d_cnt = engine.malloc(32)
    engine.memset_async(d_cnt, 0, 32)
    engine.run_benchmark_async(batch, params, d[0], d[1], d[2], d_cnt, repack=True)
    cnt = engine.to_host(d_cnt, 8).view(np.uint64)[:4].astype(np.float64)
    nw = engine.to_host(d[0], n)
    for x in d + [d_cnt]:
        engine.free(x)
    assert int(cnt[0]) == n and int(cnt[1]) == n
    edits = {0.05: 5, 0.10: 10, 0.15: 16, 0.20: 20}[err]   # benchmark_dataset.h:154: ceil(100 * 0.15f) = 16
    assert int(nw.max()) <= edits
    leap_pct, greedy_pct = 100.0 * cnt[2] / n, 100.0 * cnt[3] / n
    want_leap, want_greedy = README_ACCURACY[err]
    assert abs(leap_pct - want_leap) < readme_tolerance(want_leap, n), (err, leap_pct, want_leap)
    assert abs(greedy_pct - want_greedy) < readme_tolerance(want_greedy, n), (err, greedy_pct, want_greedy)
    cov = engine.coverage(batch, params, window=64)
    assert cov["undetermined"] == 0
    cov_pct = 100.0 * cov["covered"] / n
    assert abs(cov_pct - README_COVERAGE[err]) < 0.25, (err, cov_pct, README_COVERAGE[err])
    batch.free()
