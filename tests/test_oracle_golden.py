"""The oracle against the committed golden vectors (outputs of the compiled reference, tests/golden/make_golden.py)
and the reference's literal known-answer pairs.  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

from tests.util import greedy_defined, leap_defined

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLD, "index.json")) as fh:
    INDEX = json.load(fh)


def _digest(strings):
    return np.array([int.from_bytes(hashlib.blake2b(s.encode(), digest_size=8).digest(), "little") for s in strings],
                    np.uint64)


def _inputs_sha(hb):
    h = hashlib.sha256()
    for a in (hb.read_off, hb.reads, hb.ref_off, hb.refs):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def load_case(asm, name):
    meta = INDEX["cases"][name]
    cfg, _, _ = asm.workload(meta["workload"])
    hb = asm.generate_pairs(cfg, meta["first"], meta["n"])
    assert _inputs_sha(hb) == meta["inputs_sha256"], "generator output changed: regenerate the goldens deliberately"
    return meta, hb, np.load(os.path.join(GOLD, name + ".npz"))


@pytest.mark.parametrize("name", sorted(INDEX["cases"]))
def test_oracle_matches_reference_goldens(asm, oracle, name):
    meta, hb, gold = load_case(asm, name)
    k, x, o, e = meta["k"], meta["x"], meta["o"], meta["e"]
    gd = greedy_defined(hb, k)  # |n-m| > k is undefined in the reference
    ld = leap_defined(hb)       # LEAP reads out of bounds beyond 256
    for mode, tag in ((0, "seq"), (1, "clean")):
        cost, cig = oracle.greedy(hb, k, x, o, e, mode=mode, cigars=True)
        assert np.array_equal(cost[gd], gold[f"greedy_{tag}_cost"][gd]), (name, tag)
        assert np.array_equal(_digest(cig)[gd], gold[f"greedy_{tag}_cigar"][gd]), (name, tag, "CIGAR")
    assert np.array_equal(oracle.leap(hb, k, x, o, e)[ld], gold["leap_ed"][ld]), name
    nw = oracle.nw(hb.slice(0, meta["nw_first"]), x, o, e)
    assert np.array_equal(nw, gold["nw_first"]), (name, "NW vs independent pure-Python Gotoh")
    assert gd.mean() > 0.99 and (ld.mean() > 0.99 or meta["workload"] == "C5")


def test_known_answer_vectors(asm, oracle):
    """Literal pairs from the reference tree (GASMA/main.cpp:7-8,14-15; pymatch/algorithms/LEAP.py:188)."""
    for key, ka in INDEX["known_answers"].items():
        hb = asm.HostBatch.from_strings([(ka["read"], ka["ref"])])
        assert int(oracle.nw(hb)[0]) == ka["nw"] == int(oracle.levenshtein(hb)[0]), key
        for k in (2, 3):
            if key == "KA-1":
                continue  # |n-m| = 4 > k: characterisation only
            cost, cig = oracle.greedy(hb, k=k, mode=1, cigars=True)
            assert (int(cost[0]), cig[0]) == (ka[f"greedy_k{k}"], ka[f"greedy_cigar_k{k}"]), (key, k)
            assert int(oracle.leap(hb, k=k)[0]) == ka[f"leap_k{k}"], (key, k)
    ka0 = INDEX["known_answers"]["KA-0"]
    assert ka0["greedy_cigar_k3"] == "22M1D50M1D28M" and ka0["greedy_k3"] == 6  # the demo of GASMA/main.cpp
    ka5 = INDEX["known_answers"]["KA-5"]
    assert ka5["leap_k3"] == 2 and ka5["nw"] == 3  # LEAP's final_ED omits the last lane switch


def test_nw_is_levenshtein_for_unit_costs(asm, oracle):
    cfg, _, _ = asm.workload("C5")
    hb = asm.generate_pairs(cfg, 1000, 300)
    assert np.array_equal(oracle.nw(hb, 1, 1, 1), oracle.levenshtein(hb))


# The reference's published accuracy lines (README.md:16-20, 32-36, 47-51, 63-67: 10^6 simulated 100 bp pairs per error
# rate, k = 3, x = o = e = 1).  "Accuracy" = penalty == NW penalty, so these four lines are the only reference-held evidence
# about NW (parasail is absent from the reference tree): a wrong NW distance moves both percentages.  err 0.15 is 16 edits,
# not 15: the generator evaluates ceil(100 * 0.15f) on a float product (benchmark_dataset.h:154), reproduced by asm_gen.h.
README_ACCURACY = {0.05: (99.757, 92.975), 0.10: (98.066, 78.020), 0.15: (93.424, 57.939), 0.20: (88.579, 46.023)}
README_COVERAGE = {0.05: 97.512, 0.10: 94.213, 0.15: 90.418, 0.20: 88.289}
README_PAIRS = 1_000_000


def readme_tolerance(pct, n, sigmas=4.0):
    """Binomial standard error of the difference between two independent samples (ours: n pairs, README: 10^6) of a
    proportion near pct, times `sigmas`."""
    p = pct / 100.0
    return 100.0 * sigmas * (p * (1 - p) * (1.0 / n + 1.0 / README_PAIRS)) ** 0.5


def test_glibc_rand_emulation_is_the_running_libc(oracle):
    """The reference draws its inputs from libc rand(); the oracle's emulation (public TYPE_3 algorithm of glibc's random_r.c)
    must be the generator this machine's libc runs, output for output."""
    import ctypes

    try:
        libc = ctypes.CDLL("libc.so.6")
    except OSError:
        pytest.skip("no glibc")
    for seed in (1, 42, 20211231):
        libc.srand(ctypes.c_uint(seed))
        want = np.array([libc.rand() for _ in range(3000)], np.int32)
        assert np.array_equal(oracle.glibc_rand_stream(seed, 3000), want)


def _accuracy(asm, oracle, hb, k=3):
    oracle.set_threads(min(8, os.cpu_count() or 1))
    try:
        nw = oracle.nw(hb)
        return nw, float((oracle.leap(hb, k) == nw).mean()) * 100, float((oracle.greedy(hb, k, mode=0) == nw).mean()) * 100
    finally:
        oracle.set_threads(1)


@pytest.mark.parametrize("err", sorted(README_ACCURACY))
def test_readme_accuracy_statistics(asm, oracle, err):
    """Statistical pin of NW + LEAP + Greedy (sequential mode: the reference as run) against every simulated-data line of the
    reference's README, within 4 binomial standard errors at this sample size — on pairs drawn THE REFERENCE'S WAY
    (oracle.reference_dataset: Dataset over glibc's rand(), byte-identical to the compiled reference generator,
    tests/test_oracle_vs_reference.py).  That matters: the lagged-Fibonacci rand() makes pattern characters dependent
    (c[i] = c[i-3] + c[i-31] + carry mod 4), and at err >= 0.15 the aligners agree with NW ~0.2 points more often on such
    patterns than on independent ones."""
    n = 300_000
    hb = asm.HostBatch(*oracle.reference_dataset(n, 100, err, seed=1000 + int(round(err * 100))))
    m, nn = hb.lengths()
    edits = {0.05: 5, 0.10: 10, 0.15: 16, 0.20: 20}[err]
    assert (np.abs(nn - m) <= edits).all() and (m == 100).all()
    nw, leap_acc, greedy_acc = _accuracy(asm, oracle, hb)
    assert int(nw.max()) <= edits                       # an NW distance above the number of edits made would be a wrong NW
    want_leap, want_greedy = README_ACCURACY[err]
    assert abs(leap_acc - want_leap) < readme_tolerance(want_leap, n), (err, leap_acc, want_leap)
    assert abs(greedy_acc - want_greedy) < readme_tolerance(want_greedy, n), (err, greedy_acc, want_greedy)


@pytest.mark.parametrize("err", sorted(README_ACCURACY))
def test_product_generator_statistics_stay_close_to_the_readme(asm, oracle, err):
    """The product's own generator (csrc/asm_gen.h: same procedure, independent draws from a counter-based RNG so that any
    slice of the stream can be made on any GPU) is NOT the reference's stream; its statistics sit within 0.35 points of the
    README's on all eight numbers (measured at 10^6 pairs: +0.02..+0.08 at err <= 0.10, -0.13..-0.22 at err >= 0.15)."""
    n = 100_000
    hb = asm.generate_pairs(asm.GenConfig.exact(1000 + int(round(err * 100)), 100, err), 0, n)
    m, nn = hb.lengths()
    edits = {0.05: 5, 0.10: 10, 0.15: 16, 0.20: 20}[err]
    assert (np.abs(nn - m) <= edits).all() and (m == 100).all()
    nw, leap_acc, greedy_acc = _accuracy(asm, oracle, hb)
    assert int(nw.max()) <= edits
    want_leap, want_greedy = README_ACCURACY[err]
    assert abs(leap_acc - want_leap) < 0.35 + readme_tolerance(want_leap, n, 3.0), (err, leap_acc, want_leap)
    assert abs(greedy_acc - want_greedy) < 0.35 + readme_tolerance(want_greedy, n, 3.0), (err, greedy_acc, want_greedy)


# GASMA/benchmark/README.md:21-32,44-55,67-78,90-101 (and again :121-...): result blocks on "simulated_5000000_100_<err>_lt_eq.seq",
# i.e. Dataset(..., exact_error_rate = false): LEAP / Greedy accuracy per error rate
README_LT_EQ = {0.05: (99.461, 99.741), 0.10: (97.642, 98.142), 0.15: (94.712, 94.004), 0.20: (92.481, 90.190)}
# what the aligners AS COMMITTED give on that input stream (oracle = compiled reference, k = 3, x = o = e = 1; 3 x 10^5 pairs)
LT_EQ_AS_COMMITTED = {0.05: (99.96, 97.8), 0.10: (99.58, 92.7), 0.15: (98.42, 83.9), 0.20: (97.12, 77.8)}


@pytest.mark.parametrize("err", sorted(README_LT_EQ))
def test_lt_eq_readme_blocks_are_characterised_not_reproduced(asm, oracle, err):
    """The second set of published numbers in the reference tree (GASMA/benchmark/README.md, runs on the "lt_eq" datasets:
    0 .. ceil(L*err) - 1 edits per pair).  The input stream IS reproduced exactly — oracle.reference_dataset(exact=False) is byte
    for byte the compiled reference's generator (tests/test_oracle_vs_reference.py) — but the result blocks are not results of the
    committed aligners: there Greedy beats LEAP at err 0.05 (99.741 vs 99.461), which the committed Greedy (pinned to the compiled
    reference, 97.8 % on this stream) does not do at any band width or substitution rate tried (k = 1..3, mismatch rate 0.5..0.96),
    and LEAP's 99.461 / 97.642 / 94.712 / 92.481 lie 30-80 standard errors below what the vendored LV_BAG gives (99.96 / 99.6 /
    98.4 / 97.1).  Those runs predate the code in the tree.  Recorded as a characterisation: the values of the committed code on
    that stream are pinned here, and the distance from the README's is asserted so that a change in either direction is noticed."""
    n = 100_000
    hb = asm.HostBatch(*oracle.reference_dataset(n, 100, err, seed=2000 + int(round(err * 100)), exact=False))
    m, nn = hb.lengths()
    edits = {0.05: 5, 0.10: 10, 0.15: 16, 0.20: 20}[err] - 1
    assert (np.abs(nn - m) <= edits).all() and (m == 100).all()
    nw, leap_acc, greedy_acc = _accuracy(asm, oracle, hb)
    assert int(nw.max()) <= edits
    have_leap, have_greedy = LT_EQ_AS_COMMITTED[err]
    assert abs(leap_acc - have_leap) < 0.05 + readme_tolerance(have_leap, n), (err, leap_acc)
    assert abs(greedy_acc - have_greedy) < 0.15 + readme_tolerance(have_greedy, n), (err, greedy_acc)
    want_leap, want_greedy = README_LT_EQ[err]
    assert leap_acc - want_leap > 10 * readme_tolerance(want_leap, n, 1.0), "the published LEAP line would now be reproduced: pin it"
    assert want_greedy - greedy_acc > 10 * readme_tolerance(want_greedy, n, 1.0)


def test_product_generator_up_to_mode(asm, oracle):
    """ASM_GEN_UP_TO_ERRORS (Dataset exact = false): 0 .. ceil(L*err) - 1 edits, uniformly; same edit mix as the exact mode."""
    n = 60_000
    hb = asm.generate_pairs(asm.GenConfig.up_to(9, 100, 0.10), 0, n)
    ex = asm.generate_pairs(asm.GenConfig.exact(9, 100, 0.10), 0, n)
    m, nn = hb.lengths()
    assert (m == 100).all() and (np.abs(nn - m) <= 9).all()
    oracle.set_threads(min(8, os.cpu_count() or 1))
    try:
        nw, nw_ex = oracle.nw(hb), oracle.nw(ex)
    finally:
        oracle.set_threads(1)
    assert int(nw.max()) <= 9 and int(nw_ex.max()) <= 10
    # a tenth of the pairs has no edit at all (and a substitution writes the old base one time in four: 0.1 / (1 - 0.24) are at
    # distance 0), and the mean distance is about half the exact mode's
    assert abs(float((nw == 0).mean()) - 0.1 / (1 - 0.96 * 0.25)) < 0.01
    assert 0.40 < nw.mean() / nw_ex.mean() < 0.50
    # same stream on the reference's generator: distribution of the distances agrees within sampling noise
    rb = asm.HostBatch(*oracle.reference_dataset(n, 100, 0.10, seed=77, exact=False))
    oracle.set_threads(min(8, os.cpu_count() or 1))
    try:
        nw_ref = oracle.nw(rb)
    finally:
        oracle.set_threads(1)
    h1, h2 = np.bincount(nw, minlength=11) / n, np.bincount(nw_ref, minlength=11) / n
    assert np.abs(h1 - h2).max() < 0.01


def test_srr_shaped_line_is_a_model_not_a_pin(asm, oracle):
    """README.md:69-90 (real reads SRR611076: LEAP 89.5 %, Greedy 92.7 %) cannot be reproduced from the three per-base rates it
    quotes: the file is not in the reference tree, and independent per-base events at those rates make pairs that k = 3 LEAP
    almost always solves (the real reads' LEAP < Greedy ordering says they carry structure the rates do not describe).  What
    config C4 generates is therefore characterised, not pinned: values of this repo's seeded model."""
    cfg, _, params = asm.workload("C4")
    n = 100_000
    hb = asm.generate_pairs(cfg, 0, n)
    oracle.set_threads(min(8, os.cpu_count() or 1))
    try:
        nw = oracle.nw(hb)
        leap_acc = float((oracle.leap(hb, params.k) == nw).mean()) * 100
        greedy_acc = float((oracle.greedy(hb, params.k, mode=0) == nw).mean()) * 100
    finally:
        oracle.set_threads(1)
    m, nn = hb.lengths()
    sub = float(nw.mean()) / 100.0                      # ~ substitutions + indels per base
    assert 0.020 < sub < 0.030                          # README.md:74-76: 2.45 % + 0.047 % + 0.055 %
    assert 0.085 < float((nn != m).mean()) < 0.105      # ~ 1 - (1 - 0.00102)^100
    assert leap_acc > 99.5 and 95.5 < greedy_acc < 97.2, (leap_acc, greedy_acc)


def test_coverage_metric_and_nw_cigar(asm, oracle):
    """benchmark_coverage.h semantics + consistency of the oracle's own NW traceback."""
    cfg, _, _ = asm.workload("C2")
    hb = asm.generate_pairs(cfg, 0, 2000)
    pen, ncig = oracle.nw_cigar(hb)
    assert np.array_equal(pen, oracle.nw(hb))
    # a CIGAR's cost recomputed from its ops equals the penalty, and it consumes both strings fully
    import re

    for i in range(0, 2000, 97):
        ops = re.findall(r"(\d+)([=XID])", ncig[i])
        a, b = hb.pair(i)
        assert sum(int(c) for c, t in ops if t in "=XI") == len(a) and sum(int(c) for c, t in ops if t in "=XD") == len(b)
        assert sum(int(c) if t == "X" else (1 + (int(c) - 1)) if t in "ID" else 0 for c, t in ops) == pen[i]
    gcost, gcig = oracle.greedy(hb, 3, mode=1, cigars=True)
    cov = oracle.coverage(hb, gcig, 1, ncig, 3)
    assert 0.85 < cov.mean() <= 1.0  # README.md:36 reports 94.2 % with parasail's traceback (unpinned tie-break)
    assert oracle.coverage(hb, ncig, 1, ncig, 3).all()  # an alignment covers itself


# ---- filtering stage: bit-parallel LEAP (SIMD_ED) and SHD, goldens of tests/golden/make_golden_filter.py ----
with open(os.path.join(GOLD, "filter_index.json")) as fh:
    FILTER_INDEX = json.load(fh)


@pytest.mark.parametrize("name", sorted(FILTER_INDEX["cases"]))
def test_filter_oracle_matches_reference_goldens(asm, oracle, name):
    meta = FILTER_INDEX["cases"][name]
    cfg, _, _ = asm.workload(meta["workload"])
    hb = asm.generate_pairs(cfg, meta["first"], meta["n"])
    assert _inputs_sha(hb) == meta["inputs_sha256"], "generator output changed: regenerate the goldens deliberately"
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    state = tuple(FILTER_INDEX["warm_state"])
    for t, shd in FILTER_INDEX["simd_settings"]:
        ed, raw, ps = oracle.simd_ed(hb, t, bool(shd), 0, state)
        assert np.array_equal(ps, gold[f"pass_t{t}_shd{shd}"]), (name, t, shd, "check_pass")
        assert np.array_equal(raw, gold[f"ed_t{t}_shd{shd}"]), (name, t, shd, "get_ED")
        assert np.array_equal(ed, np.where(ps == 1, raw, -1))
    for me in FILTER_INDEX["shd_errors"]:
        assert np.array_equal(oracle.shd(hb, me), gold[f"shd_e{me}"]), (name, me)
    for g, af, x, o, e in FILTER_INDEX["affine_settings"]:  # SIMD_ED affine mode, clean: init_affine before every pair
        ed, ps = oracle.simd_ed_affine(hb, g, af, x, o, e)
        key = f"g{g}_a{af}_x{x}o{o}e{e}"
        assert np.array_equal(ps, gold["af_pass_" + key]), (name, key, "check_pass")
        assert np.array_equal(ed, np.where(ps == 1, gold["af_ed_" + key], -1)), (name, key, "get_ED")
    for g, af, x, o, e, st in FILTER_INDEX["affine_shd_settings"]:  # ... with init_affine's SHD_enable / SHD_threshold
        ed, ps = oracle.simd_ed_affine(hb, g, af, x, o, e, shd_t=st)
        key = f"g{g}_a{af}_x{x}o{o}e{e}_s{st}"
        assert np.array_equal(ps, gold["afs_pass_" + key]), (name, key, "check_pass")
        assert np.array_equal(ed, np.where(ps == 1, gold["afs_ed_" + key], -1)), (name, key, "get_ED")


def test_filter_affine_hand_cases(asm, oracle):
    """SIMD_ED affine mode, clean: an exact pair passes with get_ED() = 1000000 (run_affine returns before converge_ED is
    written, SIMD_ED.cpp:509-514); one substitution costs x; a deleted character costs the gap on the neighbouring lane, counted
    twice — once as the generation in which the lane is entered, once in converge_ED's lane term (:591-594); a tight affine
    threshold rejects what a loose one accepts."""
    base = "ACGTTGCAAGCTTAGCCATGGATCCTAGGTACCGATATCGGCATGCAAGT"
    sub = base[:20] + ("A" if base[20] != "A" else "C") + base[21:]
    hb = asm.HostBatch.from_strings([(base, base), (sub, base), ("T" * 50, base)])
    ed, ps = oracle.simd_ed_affine(hb, 3, 60, 2, 3, 1)
    assert ed[0] == 1000000 and ps[0] == 1 and ed[1] == 2 and ps[1] == 1
    ed2, ps2 = oracle.simd_ed_affine(hb, 3, 1, 2, 3, 1)
    assert ps2.tolist()[:2] == [1, 0] and ed2[1] == -1
    cfg, _, _ = asm.workload("C2")
    big = asm.generate_pairs(cfg, 7, 3000)
    loose, _ = oracle.simd_ed_affine(big, 5, 60, 2, 3, 1)
    tight, _ = oracle.simd_ed_affine(big, 5, 18, 2, 3, 1)
    assert (loose >= 0).mean() > 0.99 and 0.05 < (tight >= 0).mean() < 0.95
    assert (tight[tight >= 0] <= 18).all()  # (not comparable with `loose` pair by pair: a lane refused for its lane term lets a later generation win)


def test_filter_clean_mode_hand_cases(asm, oracle):
    """Judged alone (clean mode): exact pair 0; one substitution inside 1; a substitution in the LAST character 2, because
    lane mid-1 is swept before the main lane and reaches the end through its diagonal neighbour (SIMD_ED.cpp:303-340);
    unrelated strings fail; a passing verdict is always <= T."""
    base = "ACGTTGCAAGCTTAGCCATGGATCCTAGGTACCGATATCGGCATGCAAGT"
    sub = base[:20] + ("A" if base[20] != "A" else "C") + base[21:]
    last = base[:-1] + ("A" if base[-1] != "A" else "C")
    other = "T" * len(base)
    hb = asm.HostBatch.from_strings([(base, base), (sub, base), (last, base), (other, base), ("", "")])
    ed, _, ps = oracle.simd_ed(hb, 3, False, 1, (0, 0, 0))
    assert ed.tolist() == [0, 1, 2, -1, 0] and ps.tolist() == [1, 1, 1, 0, 1]
    cfg, _, _ = asm.workload("C2")
    big = asm.generate_pairs(cfg, 7, 3000)
    ed, _, ps = oracle.simd_ed(big, 6, False, 1, (0, 0, 0))
    assert ((ed >= 0) == (ps == 1)).all() and (ed <= 6).all() and 0.02 < ps.mean() < 0.9
