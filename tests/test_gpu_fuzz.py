"""Seeded sweep over the parameter space the dispatcher cuts into kernel families: band half-width 1..63 (LEAP) / 1..50
(Greedy), ragged lengths 0..300 (several width classes in one batch), low and high error rates, unit and general
penalties, both Greedy tail modes and alignment types.  Every cell: HIP path through the C ABI == oracle, bit for bit."""
import numpy as np
import pytest

from tests.util import greedy_defined, leap_defined, random_ragged_batch

pytestmark = pytest.mark.gpu

KS = [1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 15, 16, 17, 21, 27, 31, 32, 33, 41, 50, 57, 63]


@pytest.mark.parametrize("k", KS)
def test_band_width_sweep(asm, engine, oracle, k):
    rng = np.random.default_rng(1000 + k)
    err = float(rng.choice([0.02, 0.08, 0.15, 0.3]))
    lo, hi = (0, 300) if k % 2 else (90, 160)
    hb = random_ragged_batch(asm, 500 + k, 1200, lo, hi, err=err)
    batch = engine.upload(hb, asm.GREEDY_CLEAN)
    ld = leap_defined(hb)
    for pen in ((1, 1, 1), (2, 3, 1)):
        x, o, e = pen
        p = asm.Params.default(k=k, x=x, o=o, e=e)
        got, want = engine.align(batch, asm.LEAP, p), oracle.leap(hb, k, x, o, e)
        assert np.array_equal(got[ld], want[ld]), ("leap", k, pen, err, int((got[ld] != want[ld]).sum()))
        if k <= 50:
            gd = greedy_defined(hb, k)
            for semi in (False, True):
                pg = asm.Params.default(k=k, x=x, o=o, e=e, alignment_type=asm.ALIGN_SEMI_GLOBAL if semi else asm.ALIGN_GLOBAL)
                got, want = engine.align(batch, asm.GREEDY, pg), oracle.greedy(hb, k, x, o, e, mode=1, semi=semi)
                assert np.array_equal(got[gd], want[gd]), ("greedy", k, pen, semi, err, int((got[gd] != want[gd]).sum()))
    assert np.array_equal(engine.align(batch, asm.NW, asm.Params.default(k=k)), oracle.nw(hb))
