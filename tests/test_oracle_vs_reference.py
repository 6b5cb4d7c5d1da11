"""Pins the oracle against the REAL reference compiled in place (oracle/_ref/libasm_ref.so).  That library exists only
where /root/reference does (this container); elsewhere these tests skip and the committed goldens carry the pin."""
import os

import numpy as np
import pytest

from tests import oracle_binding
from tests.util import greedy_defined, leap_defined

pytestmark = pytest.mark.skipif(not oracle_binding.have_reference(), reason="oracle/_ref not built (no /root/reference)")


@pytest.fixture(scope="module")
def ref():
    return oracle_binding.load_reference()


def test_conversion_bit_order_and_permutation(ref):
    """bit p of a plane <-> character p; the buffer is left permuted by SRC (GASMA/bit_convert.cpp:248-369)."""
    P = [0, 2, 1, 3, 4, 6, 5, 7]
    src = np.array([8 * (q % 16) + P[q // 16] for q in range(128)], np.uint8)
    after, _, _ = ref.convert2bit1(np.arange(128, dtype=np.uint8))
    assert np.array_equal(after, src)
    for p in (0, 1, 7, 8, 31, 32, 63, 64, 100, 127):
        for ch, (e0, e1) in {"A": (0, 0), "C": (1, 0), "G": (0, 1), "T": (1, 1), "N": (0, 0), "a": (0, 0)}.items():
            buf = np.zeros(128, np.uint8)
            buf[p] = ord(ch)
            _, b0, b1 = ref.convert2bit1(buf)
            assert int.from_bytes(b0.tobytes(), "little") == e0 << p and int.from_bytes(b1.tobytes(), "little") == e1 << p


@pytest.mark.parametrize("wl,n,k,pen", [("C1", 20000, 3, (1, 1, 1)), ("C2", 50000, 3, (1, 1, 1)), ("C3", 8000, 30, (1, 1, 1)),
                                        ("C4", 30000, 3, (1, 1, 1)), ("C5", 15000, 3, (1, 1, 1)), ("C2", 15000, 3, (2, 3, 1)),
                                        ("C2", 15000, 5, (4, 6, 2)), ("C5", 8000, 10, (1, 2, 1)), ("C2", 8000, 1, (1, 1, 1)),
                                        ("C2", 5000, 50, (1, 1, 1))])
def test_oracle_equals_reference(asm, oracle, ref, wl, n, k, pen):
    cfg, _, _ = asm.workload(wl)
    hb = asm.generate_pairs(cfg, 31337, n)
    x, o, e = pen
    gd, ld = greedy_defined(hb, k), leap_defined(hb)
    for mode in (0, 1):
        oc, ocig = oracle.greedy(hb, k, x, o, e, mode=mode, cigars=True)
        rc, rcig = ref.greedy(hb, k, x, o, e, mode=mode, cigars=True)
        assert np.array_equal(oc[gd], rc[gd]), (wl, k, pen, mode)
        assert all(a == b for a, b, d in zip(ocig, rcig, gd) if d), (wl, k, pen, mode, "CIGAR")
    assert np.array_equal(oracle.leap(hb, k, x, o, e)[ld], ref.leap(hb, k, x, o, e)[ld]), (wl, k, pen)


def test_stale_tail_model(asm, oracle, ref):
    """The oracle's model of the reference's persistent buffers reproduces them byte for byte."""
    cfg, _, _ = asm.workload("C5")
    hb = asm.generate_pairs(cfg, 5, 5000)
    _, views = ref.greedy(hb, 3, mode=0, views=True)
    assert np.array_equal(views, oracle.greedy_views(hb, 0))
    # and the order dependence is real: reversing the batch changes some sequential-mode costs
    cfg2, _, _ = asm.workload("C2")
    hb2 = asm.generate_pairs(cfg2, 0, 30000)
    pairs = [hb2.pair(i) for i in range(hb2.n)]
    rev = asm.HostBatch.from_strings(pairs[::-1])
    fwd_cost, rev_cost = ref.greedy(hb2, 3, mode=0), ref.greedy(rev, 3, mode=0)[::-1]
    assert (fwd_cost != rev_cost).sum() > 0
    assert np.array_equal(ref.greedy(hb2, 3, mode=1), ref.greedy(rev, 3, mode=1)[::-1])  # clean mode is order free


# ---- filtering stage: the real SIMD_ED / SHD sources (oracle/_ref/libasm_ref_simd.so) ----
@pytest.mark.skipif(not oracle_binding.have_reference_simd(), reason="oracle/_ref/libasm_ref_simd.so not built")
@pytest.mark.parametrize("wl,n", [("C1", 8000), ("C2", 20000), ("C4", 8000), ("C5", 10000)])
def test_filter_oracle_equals_reference(asm, oracle, wl, n):
    ref = oracle_binding.load_reference_simd()
    cfg, _, _ = asm.workload(wl)
    hb = asm.generate_pairs(cfg, 271828, n)
    for t in (1, 2, 3, 4, 6, 9, 13, 16, 21, 32):
        for shd in ((False, True) if t <= 16 else (False,)):
            r_ed, r_ps = ref.simd_ed(hb, t, shd)
            _, o_raw, o_ps = oracle.simd_ed(hb, t, shd, 0, oracle_binding.SIMD_WARM_STATE)
            assert np.array_equal(r_ps, o_ps) and np.array_equal(r_ed, o_raw), (wl, t, shd)
    for me in (0, 1, 2, 4, 7, 11, 16):
        assert np.array_equal(ref.shd(hb, me), oracle.shd(hb, me)), (wl, me)


@pytest.mark.skipif(not oracle_binding.have_reference_simd(), reason="oracle/_ref/libasm_ref_simd.so not built")
def test_filter_oracle_equals_reference_on_ragged_and_dirty_input(asm, oracle):
    from tests.util import random_ragged_batch
    ref = oracle_binding.load_reference_simd()
    hb = random_ragged_batch(asm, 61, 6000, 0, 300)
    rng = np.random.default_rng(9)
    reads = hb.reads.copy()
    hits = rng.random(reads.size) < 0.01
    reads[hits] = rng.choice(np.frombuffer(b"NnacgtRY-*", np.uint8), int(hits.sum()))
    hb = asm.HostBatch(reads, hb.read_off, hb.refs, hb.ref_off)
    for t, shd in ((3, True), (3, False), (8, True), (16, True), (25, False)):
        r_ed, r_ps = ref.simd_ed(hb, t, shd)
        _, o_raw, o_ps = oracle.simd_ed(hb, t, shd, 0, oracle_binding.SIMD_WARM_STATE)
        assert np.array_equal(r_ps, o_ps) and np.array_equal(r_ed, o_raw), (t, shd)
    for me in (1, 5, 16):
        assert np.array_equal(ref.shd(hb, me), oracle.shd(hb, me)), me


@pytest.mark.parametrize("wl,n,k,pen", [("C2", 30000, 3, (1, 1, 1)), ("C5", 10000, 3, (2, 3, 1)), ("C3", 4000, 30, (1, 1, 1)),
                                        ("C2", 6000, 10, (4, 6, 2)), ("C4", 10000, 5, (1, 2, 1))])
def test_semi_global_oracle_equals_reference(asm, oracle, ref, wl, n, k, pen):
    """hurdle_matrix constructed with SEMI_GLOBAL (hurdle_matrix.h:553): cost and CIGAR, both buffer-tail modes."""
    cfg, _, _ = asm.workload(wl)
    hb = asm.generate_pairs(cfg, 4242, n)
    gd = greedy_defined(hb, k)
    differs = 0
    for mode in (0, 1):
        oc, ocig = oracle.greedy(hb, k, *pen, mode=mode, cigars=True, semi=True)
        rc, rcig = ref.greedy(hb, k, *pen, mode=mode, cigars=True, semi=True)
        assert np.array_equal(oc[gd], rc[gd]), (wl, k, pen, mode)
        assert all(a == b for a, b, d in zip(ocig, rcig, gd) if d), (wl, k, pen, mode, "CIGAR")
        differs += int((oc != oracle.greedy(hb, k, *pen, mode=mode)).sum())
    assert differs > 0, "SEMI_GLOBAL must change some costs"


def test_input_distribution_restatement_matches_the_reference_generator(asm, oracle, tmp_path):
    """oracle/asm_oracle_dataset.c (Dataset's procedure over an emulated glibc rand()) against the reference's own `Dataset`
    compiled in place (oracle/_ref/ref_dataset: benchmark_dataset.h #included, time() supplied so that the seed is known):
    the .seq files are byte-for-byte equal, for every README error rate incl. the 100 * 0.15f -> 16 edits quirk (:154)."""
    import subprocess

    from tests import oracle_binding as ob

    if not os.path.exists(ob.REF_DATASET):
        pytest.skip("oracle/_ref/ref_dataset is built only where /root/reference exists")
    for seed, n, length, err in ((777, 3000, 100, 0.15), (5, 2000, 100, 0.05), (123456, 1500, 150, 0.20), (99, 1000, 64, 0.10)):
        for exact in (True, False):  # False: the "lt_eq" files (benchmark_dataset.h:153-156), 0 .. ceil(L*err) - 1 edits per pair
            path = tmp_path / f"ref_{seed}_{int(exact)}.seq"
            subprocess.check_call([ob.REF_DATASET, str(seed), str(n), str(length), str(err), str(path), "1" if exact else "0"])
            want = asm.HostBatch.read_seq_file(str(path))
            reads, ro, refs, fo = oracle.reference_dataset(n, length, err, seed, exact=exact)
            assert np.array_equal(ro, want.read_off) and np.array_equal(fo, want.ref_off)
            assert np.array_equal(reads, want.reads) and np.array_equal(refs, want.refs)


@pytest.mark.skipif(not oracle_binding.have_reference_simd(), reason="oracle/_ref/libasm_ref_simd.so not built")
@pytest.mark.parametrize("setting", [(3, 60, 2, 3, 1), (6, 30, 1, 1, 1), (12, 120, 4, 6, 2), (2, 25, 3, 5, 2), (20, 40, 1, 2, 1)])
def test_simd_ed_affine_clean_matches_the_compiled_reference(asm, oracle, setting):
    """SIMD_ED affine mode: the oracle's clean form against the real run_affine with init_affine before every pair
    (ragged lengths, reads longer than 256, all four workload shapes)."""
    from tests.util import random_ragged_batch
    ref = oracle_binding.load_reference_simd()
    g, af, x, o, e = setting
    batches = [asm.generate_pairs(asm.workload(wl)[0], 23, n) for wl, n in (("C2", 2500), ("C3", 1200), ("C4", 2000), ("C5", 2000))]
    batches.append(random_ragged_batch(asm, 5, 1500, 0, 300, err=0.15))
    for hb in batches:
        o_ed, o_ps = oracle.simd_ed_affine(hb, g, af, x, o, e)
        r_ed, r_ps = ref.simd_ed_affine(hb, g, af, x, o, e)
        assert np.array_equal(o_ps, r_ps), setting
        assert np.array_equal(o_ed, np.where(r_ps == 1, r_ed, -1)), setting


@pytest.mark.skipif(not oracle_binding.have_reference_simd(), reason="oracle/_ref/libasm_ref_simd.so not built")
@pytest.mark.parametrize("setting", [(3, 60, 2, 3, 1, 3), (6, 30, 1, 1, 1, 6), (6, 30, 1, 1, 1, 2), (12, 120, 4, 6, 2, 5), (16, 90, 2, 3, 1, 16),
                                     (20, 40, 1, 2, 1, 0), (5, 50, 2, 3, 1, 4)])
def test_simd_ed_affine_with_shd_matches_the_compiled_reference(asm, oracle, setting):
    """init_affine(..., SHD_enable = true, SHD_threshold): run_affine starts with the mask-array SHD over the FIRST
    2*SHD_threshold+1 lane masks (SIMD_ED.cpp:489-492, SHD.cpp:334-372) — centred on the main lane only when SHD_threshold equals
    the gap threshold.  Oracle against the compiled reference, centred and off-centre, thresholds that reject most and none."""
    from tests.util import random_ragged_batch
    ref = oracle_binding.load_reference_simd()
    g, af, x, o, e, shd_t = setting
    batches = [asm.generate_pairs(asm.workload(wl)[0], 29, n) for wl, n in (("C2", 2500), ("C3", 1200), ("C4", 2000), ("C5", 2000))]
    batches.append(random_ragged_batch(asm, 6, 1500, 0, 300, err=0.15))
    rejected = 0
    for hb in batches:
        o_ed, o_ps = oracle.simd_ed_affine(hb, g, af, x, o, e, shd_t=shd_t)
        r_ed, r_ps = ref.simd_ed_affine(hb, g, af, x, o, e, shd_t=shd_t)
        assert np.array_equal(o_ps, r_ps), setting
        assert np.array_equal(o_ed, np.where(r_ps == 1, r_ed, -1)), setting
        plain, _ = oracle.simd_ed_affine(hb, g, af, x, o, e)
        assert ((o_ed == plain) | (o_ed == -1)).all()  # the filter only ever rejects
        rejected += int(((o_ed == -1) & (plain != -1)).sum())
    if shd_t <= 4:
        assert rejected > 0, setting  # a tight threshold really filters


@pytest.mark.skipif(not oracle_binding.have_reference(), reason="oracle/_ref/libasm_ref.so not built")
@pytest.mark.parametrize("mode", [1, 2, 3])
def test_leap_ed_modes_match_the_compiled_reference(asm, oracle, mode):
    """LV::init's other ED_modes (LOCAL, SEMI_FREE_BEGIN, SEMI_FREE_END; LV_BAG.h:38, LV_BAG.cpp:102-104,220-238): the oracle
    against the compiled reference, driven both ways — init() before every pair and one object with reset() as the harness does
    (the tables reset() leaves behind change nothing on these inputs either) — narrow and wide bands, three penalty sets."""
    from tests.util import random_ragged_batch
    ref = oracle_binding.load_reference()
    batches = [asm.generate_pairs(asm.workload(wl)[0], 37, n) for wl, n in (("C2", 2500), ("C3", 600), ("C4", 2000), ("C5", 1500))]
    batches.append(random_ragged_batch(asm, 7, 1200, 0, 250, err=0.15))
    differs_from_global = 0
    for hb in batches:
        ok = np.maximum(*hb.lengths()) <= 256  # LEAP beyond 256 bases is undefined in the reference
        for k, x, o, e in ((3, 1, 1, 1), (5, 2, 3, 1), (10, 1, 1, 1), (8, 4, 6, 2), (30, 1, 1, 1)):
            got = oracle.leap(hb, k, x, o, e, mode)
            assert np.array_equal(got[ok], ref.leap_mode(hb, k, x, o, e, mode, clean=True)[ok]), (mode, k, x, o, e)
            assert np.array_equal(got[ok], ref.leap_mode(hb, k, x, o, e, mode, clean=False)[ok]), (mode, k, x, o, e, "as run")
            differs_from_global += int((got[ok] != oracle.leap(hb, k, x, o, e)[ok]).sum())
    if mode in (1, 2):
        assert differs_from_global > 0  # free begin gaps really change results


@pytest.mark.skipif(not oracle_binding.have_reference_simd(), reason="oracle/_ref/libasm_ref_simd.so not built")
@pytest.mark.parametrize("mode", [1, 2, 3])
def test_simd_ed_affine_ed_modes_match_the_compiled_reference(asm, oracle, mode):
    """init_affine's ED_modes argument (SIMD_ED.cpp:476-478,497-516,589-610,748-753): every lane starts at generation 0 in LOCAL
    and SEMI_FREE_BEGIN; LOCAL and SEMI_FREE_END accept any lane that reaches the end and report final_ED — a pair exact at
    generation 0 reads 0 there and reset_affine's 1000000 in the other two.  Clean form (init_affine before every pair), with and
    without the SHD pre-filter."""
    from tests.util import random_ragged_batch
    ref = oracle_binding.load_reference_simd()
    batches = [asm.generate_pairs(asm.workload(wl)[0], 41, n) for wl, n in (("C2", 2000), ("C3", 800), ("C4", 2000), ("C5", 1500))]
    batches.append(random_ragged_batch(asm, 8, 1200, 0, 300, err=0.15))
    saw_exact = False
    for hb in batches:
        for g, af, x, o, e, st in ((3, 60, 2, 3, 1, None), (6, 30, 1, 1, 1, 2), (12, 120, 4, 6, 2, None), (20, 40, 1, 2, 1, 5)):
            o_ed, o_ps = oracle.simd_ed_affine(hb, g, af, x, o, e, shd_t=st, mode=mode)
            r_ed, r_ps = ref.simd_ed_affine(hb, g, af, x, o, e, shd_t=st, mode=mode)
            assert np.array_equal(o_ps, r_ps), (mode, g, af, x, o, e, st)
            assert np.array_equal(o_ed, np.where(r_ps == 1, r_ed, -1)), (mode, g, af, x, o, e, st)
            saw_exact |= bool(((o_ed == 0) | (o_ed == 1000000)).any())
    assert saw_exact


@pytest.mark.skipif(not oracle_binding.have_reference_simd(), reason="oracle/_ref/libasm_ref_simd.so not built")
@pytest.mark.parametrize("ed_mode", [1, 2, 3])
def test_simd_ed_levenshtein_ed_modes_match_the_compiled_reference(asm, oracle, ed_mode):
    """init_levenshtein's ED_modes argument (SIMD_ED.cpp:246-266,277-351,748-753), AS RUN: one object, a warm-up pair that
    reaches the end at generation 1 (so final_ED, final_lane_idx and converge_ED are written), then the batch.  LOCAL and
    SEMI_FREE_END read no state of earlier pairs; SEMI_FREE_BEGIN carries GLOBAL's stale-state rule (S2) — the oracle gets the
    warm-up pair prepended and a zero state."""
    ref = oracle_binding.load_reference_simd()
    warm_read = "ACGTTGCAAGCTTAGGCATCGATCCGATTAGCATGCATGC"
    warm_ref = warm_read[:20] + ("A" if warm_read[20] != "A" else "C") + warm_read[21:]
    warm = (warm_read, warm_ref)

    def with_warm(hb):
        reads = np.concatenate([np.frombuffer(warm[0].encode(), np.uint8), hb.reads])
        refs = np.concatenate([np.frombuffer(warm[1].encode(), np.uint8), hb.refs])
        ro = np.concatenate([[0], hb.read_off.astype(np.int64) + len(warm[0])]).astype(np.uint32)
        fo = np.concatenate([[0], hb.ref_off.astype(np.int64) + len(warm[1])]).astype(np.uint32)
        return asm.HostBatch(reads, ro, refs, fo)

    for wl, n in (("C2", 2500), ("C4", 2000), ("C5", 2000), ("C3", 800)):
        hb = asm.generate_pairs(asm.workload(wl)[0], 43, n)
        hw = with_warm(hb)
        for t, shd in ((3, True), (5, False), (12, True), (20, False)):
            o_ed, o_raw, o_ps = oracle.simd_ed(hw, t, shd, 0, (0, 0, 0), ed_mode=ed_mode)
            assert o_ps[0] == 1 and o_raw[0] == 1, "the warm-up pair must reach the end at generation 1 on the main lane"
            r_ed, r_ps = ref.simd_ed_edmode(hb, t, shd, ed_mode, warm)
            assert np.array_equal(o_ps[1:], r_ps), (wl, t, shd, ed_mode)
            assert np.array_equal(o_raw[1:][r_ps == 1], r_ed[r_ps == 1]), (wl, t, shd, ed_mode)
            if ed_mode in (1, 3):  # order-independent: the clean form gives the same verdicts
                c_ed, _, c_ps = oracle.simd_ed(hb, t, shd, 1, (0, 0, 0), ed_mode=ed_mode)
                assert np.array_equal(c_ps, r_ps) and np.array_equal(c_ed, np.where(r_ps == 1, r_ed, -1))
