"""Host logic of the fast Greedy kernel (csrc/asm_greedy3.h): the very pass the kernel runs per thread — g3_setup, g3_pass with
the integer rank keys, the host-built rank table — compiled for the CPU (host/g3_host_check.cpp) and diffed against the oracle,
which is pinned to the compiled reference.  No GPU needed; the GPU parity of the kernel around it is in test_gpu_parity.py."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from tests import oracle_binding, util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "approximate-string-matching_amd")
SO = os.environ.get("ASM_G3_HOSTCHECK_LIB") or os.path.join(PKG, "libg3_hostcheck.so")  # override: the sanitizer build
DEFAULT = np.array(oracle_binding.DEFAULT_PROBS, np.float64)


@pytest.fixture(scope="module")
def g3():
    subprocess.check_call(["make", "-s", "-C", PKG, "hostcheck"])
    lib = ctypes.CDLL(SO)
    lib.g3_host_batch.restype = ctypes.c_int
    lib.g3_host_table_ok.restype = ctypes.c_int
    return lib


def run(g3, orc, hb, k, mode=1, probs=DEFAULT):
    views = np.ascontiguousarray(orc.greedy_views(hb, mode=mode)).reshape(-1)
    m, n = hb.lengths()
    lens = (m.astype(np.uint32) | (n.astype(np.uint32) << 16)).astype(np.uint32)
    costs, passes = np.zeros(hb.n, np.int32), np.zeros(hb.n, np.int32)
    slow = ctypes.c_int64(0)
    probs = np.ascontiguousarray(probs, np.float64)
    rc = g3.g3_host_batch(ctypes.c_long(hb.n), views.ctypes.data_as(ctypes.c_void_p), lens.ctypes.data_as(ctypes.c_void_p), k,
                          probs.ctypes.data_as(ctypes.c_void_p), costs.ctypes.data_as(ctypes.c_void_p),
                          passes.ctypes.data_as(ctypes.c_void_p), ctypes.byref(slow))
    assert rc == 0, rc
    return costs, passes, slow.value


@pytest.mark.parametrize("wl,n", [("C1", 10_000), ("C2", 60_000), ("C4", 40_000), ("C5", 40_000), ("C3", 20_000)])
@pytest.mark.parametrize("k", [1, 2, 3])
def test_pass_matches_the_oracle_on_the_workloads(g3, asm, oracle, wl, n, k):
    """Every BASELINE workload's inputs (C3 and C5 exercise the 128-base cut and destinations outside the band) at k = 1..3."""
    cfg, _, _ = asm.workload(wl)
    hb = asm.generate_pairs(cfg, 11, n if k == 3 else n // 4)
    got, passes, _ = run(g3, oracle, hb, k)
    want = oracle.greedy(hb, k=k, mode=1)
    assert int((got != want).sum()) == 0
    assert passes.min() >= 1


def test_sequential_mode_views(g3, asm, oracle):
    """The stale-tail buffers of the reference as run reach the pass through the planes like any other bits."""
    cfg, _, _ = asm.workload("C2")
    hb = asm.generate_pairs(cfg, 5, 30_000)
    got, _, _ = run(g3, oracle, hb, 3, mode=0)
    assert np.array_equal(got, oracle.greedy(hb, k=3, mode=0))


def test_adversarial_shapes_and_the_slow_path(g3, asm, oracle):
    """Solid blocks of mismatches push hurdles + switches past the table's domain (the FP64 slow path must take over and
    agree); 128/128 pairs hit the lane-0 destination corner; empty, one-base and ragged strings; dirty alphabet."""
    rng = np.random.default_rng(3)
    acgt = "ACGT"
    rnd = lambda L: "".join(acgt[i] for i in rng.integers(0, 4, L))
    pairs = [("A" * 128, "T" * 128), ("A" * 100, "T" * 100), ("A" * 128, "A" * 128), ("ACGT" * 32, "ACGT" * 32),
             ("", ""), ("A", ""), ("", "A"), ("A", "A"), ("A", "C"), ("AC", "CA"), ("A" * 128, "A" * 125), ("A" * 125, "A" * 128),
             ("A" * 70 + "C" * 58, "C" * 58 + "A" * 70), ("AC" * 64, "CA" * 64), ("A" * 64 + "T" * 64, "T" * 64 + "A" * 64),
             ("ACGTN" * 20, "ACGTN" * 20), ("acgt" * 25, "ACGT" * 25)]
    for L in (128, 127, 100, 65, 64, 63, 33, 10, 3):
        a = rnd(L)
        pairs += [(a, a), (a, a[1:]), (a[1:], a), (a, a[:-1]), (a, a[2:] + "GG"), (a, "T" * 40 + a[40:]), ("C" * 70 + a[70:], a),
                  (a, rnd(L)), (a[: L // 2] + "T" * 66, a)]
    for _ in range(2000):  # long solid blocks at random places, random lengths up to 200 (cut at 128)
        L = int(rng.integers(1, 200))
        a = list(rnd(L))
        b = list(a)
        for _ in range(int(rng.integers(0, 4))):
            p, w = int(rng.integers(0, L)), int(rng.integers(1, 90))
            b[p:p + w] = list("ACGT"[(acgt.index(c) + 1) % 4] for c in b[p:p + w])
        if rng.random() < 0.3:
            q = int(rng.integers(0, len(b) + 1))
            b[q:q] = list(rnd(int(rng.integers(1, 4))))
        if rng.random() < 0.3 and len(b) > 4:
            q = int(rng.integers(0, len(b) - 3))
            del b[q:q + int(rng.integers(1, 4))]
        pairs.append(("".join(a), "".join(b)))
    hb = asm.HostBatch.from_strings(pairs)
    slow_seen = 0
    for k in (1, 2, 3):
        got, _, slow = run(g3, oracle, hb, k)
        want = oracle.greedy(hb, k=k, mode=1)
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, [(pairs[i], int(got[i]), int(want[i])) for i in bad[:3]]
        slow_seen += slow
    assert slow_seen > 0, "the inputs were meant to reach the slow path"


def test_ragged_random_batch(g3, asm, oracle):
    hb = util.random_ragged_batch(asm, 99, 6000, lo=0, hi=300, err=0.15)
    for k in (2, 3):
        got, _, _ = run(g3, oracle, hb, k)
        assert np.array_equal(got, oracle.greedy(hb, k=k, mode=1))


def test_rank_table_only_for_scores_with_the_class_structure(g3):
    """The integer keys need mismatch_sig == indel_sig (the reference's defaults: 0.20/3/0.25 == 0.40/3/2/0.25 bit for bit);
    other probabilities keep the FP64 kernel."""
    ok = lambda p, k=3: g3.g3_host_table_ok(np.ascontiguousarray(p, np.float64).ctypes.data_as(ctypes.c_void_p), k)
    assert ok(DEFAULT) == 1
    assert ok([0.80, 0.20 / 3, 0.40 / 3], 1) == 1 and ok([0.80, 0.20 / 3, 0.40 / 3], 2) == 1
    assert ok([0.90, 0.05, 0.10]) == 1          # still mismatch == indel / 2
    assert ok([0.80, 0.10, 0.05]) == 0          # different significances: no classes
    assert ok(DEFAULT, 4) == 0                  # k > 3 is not this kernel's


def test_other_probabilities_with_equal_significances(g3, asm, oracle):
    """p_mismatch == p_indel / 2 makes the two significances equal for any p_match: the table is rebuilt for them."""
    cfg, _, _ = asm.workload("C2")
    hb = asm.generate_pairs(cfg, 21, 20_000)
    probs = (0.90, 0.05, 0.10)
    got, _, _ = run(g3, oracle, hb, 3, probs=probs)
    assert np.array_equal(got, oracle.greedy(hb, k=3, mode=1, probs=probs))
