"""GPU parity tests proper: the HIP path (through the C ABI) against the oracle on the same seeded inputs.
Bar: bit-exact int32 penalties.  Sizes are what the oracle finishes in seconds."""
import os

import numpy as np
import pytest

ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))

from tests.util import KNOWN_PAIRS, greedy_defined, random_ragged_batch

pytestmark = pytest.mark.gpu


def _check(name, got, want, hb):
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (f"{name}: {bad.size}/{hb.n} differ; first {bad[:5]} got {got[bad[:5]]} want {want[bad[:5]]} "
                           f"pair {hb.pair(int(bad[0]))}")


@pytest.mark.parametrize("cfgname,n", [("C1", 10000), ("C2", 20000), ("C4", 20000)])
def test_three_aligners_narrow_band(asm, engine, oracle, cfgname, n):
    """BASELINE configs with k=3, x=o=e=1: thread-per-pair kernels."""
    cfg, _, params = asm.workload(cfgname)
    hb = asm.generate_pairs(cfg, 0, n)
    batch = engine.upload(hb, asm.GREEDY_CLEAN)
    _check("nw", engine.align(batch, asm.NW, params), oracle.nw(hb), hb)
    _check("leap", engine.align(batch, asm.LEAP, params), oracle.leap(hb, k=params.k), hb)
    _check("greedy", engine.align(batch, asm.GREEDY, params), oracle.greedy(hb, k=params.k, mode=1), hb)


def test_sequential_mode_matches_reference_order_dependence(asm, engine, oracle):
    """Greedy 'sequential' mode = the reference as run (stale buffer tails)."""
    cfg, _, params = asm.workload("C2")
    hb = asm.generate_pairs(cfg, 0, 20000)
    batch = engine.upload(hb, asm.GREEDY_SEQUENTIAL)
    want = oracle.greedy(hb, k=3, mode=0)
    _check("greedy-seq", engine.align(batch, asm.GREEDY, params), want, hb)
    assert (want != oracle.greedy(hb, k=3, mode=1)).sum() > 0  # the two modes really differ on this input
    # LEAP / NW ignore the tail bits
    _check("leap", engine.align(batch, asm.LEAP, params), oracle.leap(hb, k=3), hb)
    _check("nw", engine.align(batch, asm.NW, params), oracle.nw(hb), hb)


@pytest.mark.parametrize("wl,n,k", [("C5", 12000, 3), ("C1", 10000, 3), ("C4", 8000, 2)])
def test_sequential_mode_device_resolver(asm, engine, oracle, wl, n, k):
    """The stale-tail chain is resolved on the GPU (csrc/asm_tails.h: three passes over 40-pair chunks), both for
    uploaded batches and for batches generated on the device; mixed lengths make long carry chains."""
    cfg, _, _ = asm.workload(wl)
    hb = asm.generate_pairs(cfg, 17, n)
    want = oracle.greedy(hb, k=k, mode=0)
    params = asm.Params.default(k=k)
    _check("upload", engine.align(engine.upload(hb, asm.GREEDY_SEQUENTIAL), asm.GREEDY, params), want, hb)
    _check("generate", engine.align(engine.generate(cfg, 17, n, asm.GREEDY_SEQUENTIAL), asm.GREEDY, params), want, hb)


def test_sequential_mode_ragged_lengths(asm, engine, oracle):
    hb = random_ragged_batch(asm, 21, 6000, 0, 200)
    want = oracle.greedy(hb, k=3, mode=0)
    got = engine.align(engine.upload(hb, asm.GREEDY_SEQUENTIAL), asm.GREEDY, asm.Params.default(k=3))
    _check("ragged-seq", got, want, hb)


@pytest.mark.parametrize("sizes", [(7001, 5000, 2999), (2560, 2560, 10), (13, 0, 9000, 1), (25000,),
                                   (5120, 40, 5121, 39, 1), (41, 5119, 10, 10247, 3)])
def test_sequential_mode_chains_across_batches(asm, engine, oracle, sizes):
    """One file cut into shards / chunks (sizes with every phase mod 10, an empty one, sizes at and around the resolver's
    40-pair chunks and 5120-pair workgroups, one spanning several workgroups):
    summary per shard on the device -> fold on the host -> resolve each shard from the folded state.  Equals the reference as
    run over the WHOLE file (oracle mode 0, and the compiled reference where it travelled), which shards that each start from
    empty buffers do not."""
    from tests import oracle_binding as ob

    long_cfg = asm.GenConfig.exact(31, 100, 0.10, length_hi=128)
    mixed_cfg = asm.GenConfig.exact(32, 30, 0.10, length_hi=128)
    total = sum(sizes)
    a, b = asm.generate_pairs(long_cfg, 0, total // 3), asm.generate_pairs(mixed_cfg, 0, total - total // 3)
    hb = asm.HostBatch(np.concatenate([a.reads, b.reads]), np.concatenate([a.read_off, b.read_off[1:] + a.read_off[-1]]),
                       np.concatenate([a.refs, b.refs]), np.concatenate([a.ref_off, b.ref_off[1:] + a.ref_off[-1]]))
    params = asm.Params.default(k=3)
    want, want_cigars = oracle.greedy(hb, k=3, mode=0, cigars=True)
    if ob.have_reference():
        ok = greedy_defined(hb, 3)
        assert np.array_equal(ob.load_reference().greedy(hb, k=3, mode=0)[ok], want[ok])
    state = np.zeros(256, np.uint8)
    lo = 0
    got, got_cigars = [], []
    for n in sizes:
        part = hb.slice(lo, lo + n)
        batch = engine.upload(part, asm.GREEDY_CLEAN)       # uploaded without any knowledge of the pairs before it
        summary = engine.tail_summary(batch)
        assert np.array_equal(summary, oracle.tail_summary(part)), f"summary of [{lo}, {lo + n})"
        engine.resolve_tails(batch, state)
        cost, cigars, _ = engine.greedy_with_cigar(batch, params, cap=64)
        got.append(cost)
        got_cigars += cigars
        state = asm.tail_state_advance(state, summary, n)
        lo += n
    _check("chained", np.concatenate(got), want, hb)
    assert got_cigars == want_cigars
    oracle.greedy_views(hb, 0)
    assert np.array_equal(state, np.array([ob.base_code(c) for c in oracle.final_buffers()], np.uint8))


def test_c3_wide_band_150bp(asm, engine, oracle):
    """C3: 150 bp, err .20, k=30 — workgroup-per-pair kernels; Greedy sees the first 128 bases (F6)."""
    cfg, _, params = asm.workload("C3")
    hb = asm.generate_pairs(cfg, 0, 4000)
    for mode in (asm.GREEDY_CLEAN, asm.GREEDY_SEQUENTIAL):
        batch = engine.upload(hb, mode)
        _check("greedy", engine.align(batch, asm.GREEDY, params), oracle.greedy(hb, k=30, mode=mode), hb)
    _check("leap", engine.align(batch, asm.LEAP, params), oracle.leap(hb, k=30), hb)
    _check("nw", engine.align(batch, asm.NW, params), oracle.nw(hb), hb)


def test_c5_mixed_lengths(asm, engine, oracle):
    """C5: 64-300 bp in one batch (three 128-bit granules per plane)."""
    cfg, _, params = asm.workload("C5")
    hb = asm.generate_pairs(cfg, 0, 6000)
    batch = engine.upload(hb, asm.GREEDY_CLEAN)
    _check("nw", engine.align(batch, asm.NW, params), oracle.nw(hb), hb)
    _check("leap", engine.align(batch, asm.LEAP, params), oracle.leap(hb, k=3), hb)
    _check("greedy", engine.align(batch, asm.GREEDY, params), oracle.greedy(hb, k=3, mode=1), hb)


@pytest.mark.parametrize("k", [0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 50])
def test_band_widths(asm, engine, oracle, k):
    """Every dispatch boundary of the band: LEAP thread per pair up to k = 10 for strings of one granule (5 beyond), four threads
    per pair from there; Greedy thread per pair up to 16, then wave(s) per pair."""
    cfg, _, _ = asm.workload("C2")
    hb = asm.generate_pairs(cfg, 7, 3000)
    params = asm.Params.default(k=k)
    batch = engine.upload(hb, asm.GREEDY_CLEAN)
    _check(f"greedy k={k}", engine.align(batch, asm.GREEDY, params), oracle.greedy(hb, k=k, mode=1), hb)
    _check(f"leap k={k}", engine.align(batch, asm.LEAP, params), oracle.leap(hb, k=k), hb)


@pytest.mark.parametrize("k,x,o,e", [(3, 2, 3, 1), (5, 4, 6, 2), (10, 1, 2, 1), (3, 1, 1, 1), (3, 3, 5, 5), (2, 15, 15, 1), (3, 9, 12, 3),
                                     (5, 15, 15, 15), (4, 1, 7, 2), (1, 2, 2, 2)])
def test_general_penalties(asm, engine, oracle, k, x, o, e):
    """Arbitrary (x, o, e): affine NW, generic LEAP, Greedy costs."""
    cfg, _, _ = asm.workload("C2")
    hb = asm.generate_pairs(cfg, 11, 3000)
    params = asm.Params.default(k=k, x=x, o=o, e=e)
    batch = engine.upload(hb, asm.GREEDY_CLEAN)
    _check("nw", engine.align(batch, asm.NW, params), oracle.nw(hb, x, o, e), hb)
    _check("leap", engine.align(batch, asm.LEAP, params), oracle.leap(hb, k, x, o, e), hb)
    _check("greedy", engine.align(batch, asm.GREEDY, params), oracle.greedy(hb, k, x, o, e, mode=1), hb)


@pytest.mark.parametrize("x,o,e", [(2, 3, 1), (4, 6, 2), (5, 2, 2), (1, 4, 1)])
def test_affine_nw_wavefront_pass_and_full_matrix_fallback(asm, engine, oracle, x, o, e):
    """Affine NW = banded wavefront pass + full-matrix pass over the pairs the band cannot settle: noisy pairs (most fall
    back), clean pairs (none do), ragged lengths with |n-m| beyond the band, and the mixed-length workload."""
    for wl, err, n in (("C2", 0.25, 3000), ("C2", 0.01, 3000), ("C5", None, 3000)):
        cfg, _, _ = asm.workload(wl)
        if err is not None:
            cfg.err = err
        hb = asm.generate_pairs(cfg, 23, n)
        batch = engine.upload(hb, asm.GREEDY_CLEAN)
        _check(f"nw {wl} err={err}", engine.align(batch, asm.NW, asm.Params.default(x=x, o=o, e=e)), oracle.nw(hb, x, o, e), hb)
    hb = random_ragged_batch(asm, 31, 1500, 0, 250)
    batch = engine.upload(hb, asm.GREEDY_CLEAN)
    _check("nw ragged", engine.align(batch, asm.NW, asm.Params.default(x=x, o=o, e=e)), oracle.nw(hb, x, o, e), hb)


@pytest.mark.parametrize("x,o,e", [(1, 1, 0), (0, 2, 1), (1, 0, 1), (0, 0, 0), (3, 0, 0), (40, 50, 20)])
def test_nw_zero_and_large_penalties(asm, engine, oracle, x, o, e):
    """A zero among (x, o, e) must not go through the wavefront kernel (its ring would read the generation being written):
    plain Gotoh handles it.  Penalties that would saturate the 16-bit boundary cells are refused, not truncated."""
    cfg, _, _ = asm.workload("C2")
    hb = asm.generate_pairs(cfg, 29, 2000)
    batch = engine.upload(hb, asm.GREEDY_CLEAN)
    _check(f"nw ({x},{o},{e})", engine.align(batch, asm.NW, asm.Params.default(x=x, o=o, e=e)), oracle.nw(hb, x, o, e), hb)
    hb = random_ragged_batch(asm, 37, 800, 0, 250)
    batch = engine.upload(hb, asm.GREEDY_CLEAN)
    _check(f"nw ragged ({x},{o},{e})", engine.align(batch, asm.NW, asm.Params.default(x=x, o=o, e=e)), oracle.nw(hb, x, o, e), hb)


def test_nw_penalties_beyond_the_cell_width_are_refused(asm, engine):
    hb = asm.HostBatch.from_strings([("ACGT" * 100, "ACGA" * 100)])
    batch = engine.upload(hb, asm.GREEDY_CLEAN)
    for x, o, e in ((1, 1, 40), (1, 20000, 1), (20000, 1, 1)):
        with pytest.raises(asm.AsmError) as ei:
            engine.align(batch, asm.NW, asm.Params.default(x=x, o=o, e=e))
        assert ei.value.code == -4
    # the same extension penalty is fine on a short batch
    short = engine.upload(asm.HostBatch.from_strings([("ACGTACGT", "ACGACGT")]), asm.GREEDY_CLEAN)
    assert engine.align(short, asm.NW, asm.Params.default(x=1, o=1, e=40)).tolist() == [1]


@pytest.mark.parametrize("x,o,e", [(1, 1, 1), (2, 3, 1)])
def test_ragged_and_edge_lengths(asm, engine, oracle, x, o, e):
    """Empty strings, 1, 63/64/65, 127/128/129, 255/256/257, 300 and random lengths in one batch."""
    hb = random_ragged_batch(asm, 5, 1500, 0, 300)
    params = asm.Params.default(k=3, x=x, o=o, e=e)
    batch = engine.upload(hb, asm.GREEDY_CLEAN)
    _check("nw", engine.align(batch, asm.NW, params), oracle.nw(hb, x, o, e), hb)
    _check("leap", engine.align(batch, asm.LEAP, params), oracle.leap(hb, 3, x, o, e), hb)
    _check("greedy", engine.align(batch, asm.GREEDY, params), oracle.greedy(hb, 3, x, o, e, mode=1), hb)
    _check("greedy k=12", engine.align(batch, asm.GREEDY, asm.Params.default(k=12, x=x, o=o, e=e)),
           oracle.greedy(hb, 12, x, o, e, mode=1), hb)
    _check("leap k=12", engine.align(batch, asm.LEAP, asm.Params.default(k=12, x=x, o=o, e=e)),
           oracle.leap(hb, 12, x, o, e), hb)


def test_long_sequences_512(asm, engine, oracle):
    hb = random_ragged_batch(asm, 9, 400, 300, 490, err=0.04)
    params = asm.Params.default(k=3)
    batch = engine.upload(hb, asm.GREEDY_CLEAN)
    _check("nw", engine.align(batch, asm.NW, params), oracle.nw(hb), hb)
    _check("nw affine", engine.align(batch, asm.NW, asm.Params.default(x=2, o=3, e=1)), oracle.nw(hb, 2, 3, 1), hb)
    _check("leap", engine.align(batch, asm.LEAP, params), oracle.leap(hb, k=3), hb)
    _check("greedy", engine.align(batch, asm.GREEDY, params), oracle.greedy(hb, k=3, mode=1), hb)


def test_known_answer_vectors(asm, engine):
    """The reference's literal pairs, values produced by the compiled reference."""
    hb = asm.HostBatch.from_strings([KNOWN_PAIRS[k] for k in ("KA-0", "KA-5", "KA-6")])
    p = asm.Params.default(k=3)
    assert engine.align_host(hb, asm.GREEDY, p, asm.GREEDY_CLEAN).tolist() == [6, 3, 0]
    assert engine.align_host(hb, asm.LEAP, p).tolist() == [5, 2, 0]      # KA-5: LEAP 2 < NW 3 (F5)
    assert engine.align_host(hb, asm.NW, p).tolist() == [5, 3, 0]


def test_empty_batch_and_errors(asm, engine):
    hb = asm.HostBatch.from_strings([])
    assert engine.align_host(hb, asm.NW, asm.Params.default()).shape == (0,)
    with pytest.raises(asm.AsmError):
        engine.align_host(asm.HostBatch.from_strings([("ACGT", "ACGT")]), asm.GREEDY, asm.Params.default(k=51))
    with pytest.raises(asm.AsmError):
        engine.align_host(asm.HostBatch.from_strings([("ACGT", "ACGT")]), asm.LEAP, asm.Params.default(o=1, e=2))
    with pytest.raises(asm.AsmError):
        engine.align_host(asm.HostBatch.from_strings([("A" * 513, "ACGT")]), asm.NW, asm.Params.default())


def test_device_generator_matches_host_generator(asm, engine):
    cases = [(name, asm.workload(name)[0], n) for name, n in (("C2", 5000), ("C4", 5000), ("C5", 3000), ("C3", 2000))]
    cases.append(("lt_eq", asm.GenConfig.up_to(31, 100, 0.15), 4000))   # Dataset exact = false (benchmark_dataset.h:153-156)
    cases.append(("lt_eq mixed", asm.GenConfig.up_to(32, 40, 0.20, length_hi=180), 3000))
    for name, cfg, n in cases:
        hb = asm.generate_pairs(cfg, 123, n)
        db = engine.generate(cfg, 123, n).download()
        assert np.array_equal(hb.read_off, db.read_off) and np.array_equal(hb.ref_off, db.ref_off), name
        assert np.array_equal(hb.reads, db.reads) and np.array_equal(hb.refs, db.refs), name


def test_count_equal(asm, engine, oracle):
    cfg, _, params = asm.workload("C2")
    hb = asm.generate_pairs(cfg, 0, 10000)
    batch = engine.upload(hb)
    d_nw, d_leap, d_cnt = engine.malloc(4 * hb.n), engine.malloc(4 * hb.n), engine.malloc(8)
    engine.align_async(batch, asm.NW, params, d_nw)
    engine.align_async(batch, asm.LEAP, params, d_leap)
    engine.memset_async(d_cnt, 0, 8)
    engine.count_equal_async(d_nw, d_leap, hb.n, d_cnt)
    cnt = int(engine.to_host(d_cnt, 1, np.uint64)[0])
    assert cnt == int((oracle.nw(hb) == oracle.leap(hb, k=3)).sum())
    for p in (d_nw, d_leap, d_cnt):
        engine.free(p)


def test_asm_bench_harness_prints_reference_block(asm, oracle, tmp_path):
    """The C++ host side (host/asm_compat.hpp + asm-bench, counterpart of GASMA/benchmark/benchmark.cpp): reads a
    '>read\\n<ref\\n' file, runs the batch on the GPU, prints the reference's results block (benchmark_utils.h:390-402)."""
    import os
    import subprocess

    exe = os.path.join(os.path.dirname(asm.LIB_PATH), "asm-bench")
    cfg, _, _ = asm.workload("C2")
    hb = asm.generate_pairs(cfg, 0, 20000)
    path = str(tmp_path / "pairs.seq")
    hb.write_seq_file(path)
    nw = oracle.nw(hb)
    (tmp_path / "answers.txt").write_text("\n".join(str(int(v)) for v in nw) + "\n")
    for extra, mode in (([], 0), (["--answers", str(tmp_path / "answers.txt")], 0), (["--mode", "clean"], 1)):
        out = subprocess.check_output([exe, "--file", path, "--n", "20000", "--k", "3"] + extra, text=True)
        lines = out.splitlines()
        assert "===================== Benchmark Results =====================" in lines
        assert "Total number of alignments: 20000" in lines
        assert "[Accuracy] (percentage of alignments matching optimal penalty)" in lines
        want = {"Needleman-Wunsch": 100.0, "LEAP": 100.0 * float((oracle.leap(hb, 3) == nw).mean()),
                "Greedy": 100.0 * float((oracle.greedy(hb, 3, mode=mode) == nw).mean())}
        acc = lines[lines.index("[Accuracy] (percentage of alignments matching optimal penalty)") + 1:][:3]
        for line, (name, val) in zip(acc, want.items()):
            assert line == "=> %-16s | %.3f %%" % (name, val), (line, name, val)
        # [Coverage]: Greedy CIGAR vs the NW traceback (this library's documented tie-break = the oracle's)
        gc = oracle.greedy(hb, 3, mode=mode, cigars=True)[1]
        cov = oracle.coverage(hb, gc, 1, oracle.nw_cigar(hb)[1], 3)
        i = lines.index("[Coverage] (percentage of alignments covering all long consecutive matches)")
        assert lines[i + 1] == "=> %-16s | %.3f %%" % ("Greedy", 100.0 * float(cov.mean())), lines[i + 1]


@pytest.mark.parametrize("wl,n,k,mode", [("C2", 20000, 3, 1), ("C2", 8000, 3, 0), ("C1", 5000, 2, 1), ("C3", 3000, 30, 1),
                                         ("C5", 8000, 3, 1), ("C2", 3000, 10, 1), ("C2", 2000, 40, 1)])
def test_greedy_cigar(asm, engine, oracle, wl, n, k, mode):
    """hurdle_matrix::get_CIGAR on the device (narrow, wave-per-pair and workgroup-per-pair kernels), string for
    string against the oracle — whose CIGARs are pinned to the compiled reference by tests/golden."""
    cfg, _, _ = asm.workload(wl)
    hb = asm.generate_pairs(cfg, 3, n)
    want_cost, want_cig = oracle.greedy(hb, k=k, mode=mode, cigars=True)
    batch = engine.upload(hb, mode)
    cost, cig, nops = engine.greedy_with_cigar(batch, asm.Params.default(k=k), cap=64)
    _check("cost", cost, want_cost, hb)
    assert int(nops.max()) <= 64
    bad = [i for i in range(n) if cig[i] != want_cig[i]]
    assert not bad, (len(bad), bad[:3], cig[bad[0]], want_cig[bad[0]])


def test_accuracy_counters(asm, engine, oracle):
    """`_run_benchmark`'s counters (benchmark_utils.h:238,249-255) incl. an answers array; n not a multiple of 4."""
    cfg, _, params = asm.workload("C2")
    n = 9999
    hb = asm.generate_pairs(cfg, 5, n)
    batch = engine.upload(hb)
    d = [engine.malloc(4 * n) for _ in range(3)]
    d_cnt = engine.malloc(32)
    engine.memset_async(d_cnt, 0, 32)
    engine.run_benchmark_async(batch, params, d[0], d[1], d[2], d_cnt, repack=True)
    nw, leap, greedy = oracle.nw(hb), oracle.leap(hb, 3), oracle.greedy(hb, 3, mode=1)
    assert engine.to_host(d_cnt, 4, np.uint64).tolist() == [n, n, int((leap == nw).sum()), int((greedy == nw).sum())]
    # answers file semantics (benchmark_utils.h:249-252,358-368): INT32_MIN means "use the NW penalty"
    answers = nw.copy()
    answers[::3] = np.iinfo(np.int32).min
    answers[1::7] += 1
    d_ans = engine.malloc(4 * n)
    engine._chk(engine.lib.asm_memcpy_h2d(engine.h, d_ans, answers.ctypes.data, 4 * n))
    engine.memset_async(d_cnt, 0, 32)
    engine.accuracy_async(d[0], d[1], d[2], n, d_cnt, d_answers=d_ans)
    want = np.where(answers == np.iinfo(np.int32).min, nw, answers)
    assert engine.to_host(d_cnt, 4, np.uint64).tolist() == [n, int((nw == want).sum()), int((leap == want).sum()),
                                                             int((greedy == want).sum())]
    for p in d + [d_cnt, d_ans]:
        engine.free(p)


@pytest.mark.parametrize("wl,n,window,err", [("C2", 20000, 32, None), ("C2", 6000, 64, None), ("C1", 6000, 32, None),
                                             ("C4", 10000, 32, None), ("C5", 8000, 64, None), ("C3", 3000, 64, None),
                                             ("C2", 4000, 32, 0.30), ("C5", 3000, 32, 0.25)])
def test_coverage_counter_and_nw_traceback(asm, engine, oracle, wl, n, window, err):
    """The [Coverage] line of the harness (benchmark_utils.h:214-225,256-258) on the device: NW traceback with the
    oracle's documented tie-break, LCM strings, covers().  Compared pair by pair with the oracle — EVERY pair: what the
    banded pass cannot answer (distance above window/2 - 3: most pairs of the noisy cases) goes through the full matrix."""
    cfg, _, _ = asm.workload(wl)
    if err is not None:
        cfg.err = err
    hb = asm.generate_pairs(cfg, 9, n)
    params = asm.Params.default(k=3)
    batch = engine.upload(hb, asm.GREEDY_CLEAN)
    got = engine.coverage(batch, params, window=window, cap=192, want_nw_cigars=True)
    gcost, gcig = oracle.greedy(hb, k=3, mode=1, cigars=True)
    pen, ncig = oracle.nw_cigar(hb)
    want = oracle.coverage(hb, gcig, 1, ncig, 3)
    assert got["undetermined"] == 0 and not (got["cover"] == 2).any()
    assert got["covered"] == int((got["cover"] == 1).sum())
    if err is not None:
        assert (pen > window // 2 - 3).mean() > 0.3          # the fallback really ran on a large share
    bad = np.nonzero(got["cover"] != want)[0]
    assert bad.size == 0, (bad[:5], got["cover"][bad[:5]], want[bad[:5]], pen[bad[:5]])
    wrong = [i for i in range(n) if got["nw_cigars"][i] != ncig[i]]
    assert not wrong, (len(wrong), pen[wrong[0]], got["nw_cigars"][wrong[0]], ncig[wrong[0]])
    if wl == "C2" and err is None:
        assert 0.90 < want.mean() < 0.99  # README.md:36 reports 94.2 % with parasail's traceback


@pytest.mark.parametrize("x,o,e", [(2, 3, 1), (4, 6, 2), (1, 2, 1), (3, 1, 1), (1, 1, 0), (0, 2, 1)])
def test_coverage_with_general_penalties(asm, engine, oracle, x, o, e):
    """benchmark_utils.h:214-225 computes coverage for whatever (x, o, e) the harness was built with: affine traceback on the
    device (full Gotoh matrix with stored directions), CIGAR for CIGAR and verdict for verdict against the oracle; 100 bp,
    mixed 64-300 bp and ragged 0-250 bp batches."""
    from tests.util import random_ragged_batch as ragged

    k = 3
    for name, hb in (("C2", asm.generate_pairs(asm.workload("C2")[0], 41, 3000)), ("C5", asm.generate_pairs(asm.workload("C5")[0], 43, 1500)),
                     ("ragged", ragged(asm, 47, 800, 0, 250))):
        params = asm.Params.default(k=k, x=x, o=o, e=e)
        batch = engine.upload(hb, asm.GREEDY_CLEAN)
        got = engine.coverage(batch, params, window=64, cap=255, want_nw_cigars=True)
        gcost, gcig = oracle.greedy(hb, k, x, o, e, mode=1, cigars=True)
        pen, ncig = oracle.nw_cigar(hb, x, o, e)
        want = oracle.coverage(hb, gcig, 1, ncig, 3)
        assert got["undetermined"] == 0
        wrong = [i for i in range(hb.n) if got["nw_cigars"][i] != ncig[i]]
        assert not wrong, (name, len(wrong), hb.pair(wrong[0]), got["nw_cigars"][wrong[0]], ncig[wrong[0]])
        bad = np.nonzero(got["cover"] != want)[0]
        assert bad.size == 0, (name, bad[:5], got["cover"][bad[:5]], want[bad[:5]])
        assert np.array_equal(got["greedy_cost"], gcost)


def test_leap_work_hint_changes_schedule_not_results(asm, engine, oracle):
    """LEAP scheduled by a per-pair work estimate (the NW penalties; also garbage and mixed-length batches)."""
    for wl, n in (("C2", 30000), ("C5", 9000)):
        cfg, _, params = asm.workload(wl)
        hb = asm.generate_pairs(cfg, 2, n)
        batch = engine.upload(hb)
        want = oracle.leap(hb, k=3)
        d_nw, d_leap = engine.malloc(4 * n), engine.malloc(4 * n)
        engine.align_async(batch, asm.NW, params, d_nw)
        engine.align_hinted_async(batch, asm.LEAP, params, d_nw, d_leap)
        _check("hinted by NW", engine.to_host(d_leap, n), want, hb)
        junk = (np.arange(n, dtype=np.int32) * 7919) % 200 - 50
        engine._chk(engine.lib.asm_memcpy_h2d(engine.h, d_nw, junk.ctypes.data, 4 * n))
        engine.align_hinted_async(batch, asm.LEAP, params, d_nw, d_leap)
        _check("hinted by junk", engine.to_host(d_leap, n), want, hb)
        engine.free(d_nw), engine.free(d_leap)


@pytest.mark.parametrize("wl,n,k,pen", [("C3", 6000, 30, (1, 1, 1)), ("C2", 9000, 8, (1, 1, 1)), ("C2", 9000, 12, (2, 3, 1)),
                                         ("C5", 9000, 9, (1, 1, 1)), ("C5", 6000, 20, (4, 6, 2))])
def test_wide_band_leap_work_sorted(asm, engine, oracle, wl, n, k, pen):
    """Wide-band LEAP (four threads per pair) scheduled by a work estimate: sorted inside workgroups (short strings, unit
    penalties) or over the whole bucket (one radix pass: long strings, general penalties, each class of a mixed-length batch);
    hinted by NW, by junk, and — asm_run_benchmark_async without NW, the C3 case — by the Greedy penalties of the same call."""
    cfg, _, _ = asm.workload(wl)
    hb = asm.generate_pairs(cfg, 5, n)
    x, o, e = pen
    params = asm.Params.default(k=k, x=x, o=o, e=e)
    batch = engine.upload(hb, asm.GREEDY_CLEAN)
    want = oracle.leap(hb, k, x, o, e)
    d_hint, d_leap, d_greedy = engine.malloc(4 * n), engine.malloc(4 * n), engine.malloc(4 * n)
    engine.align_async(batch, asm.NW, params, d_hint)
    engine.align_hinted_async(batch, asm.LEAP, params, d_hint, d_leap)
    _check("hinted by NW", engine.to_host(d_leap, n), want, hb)
    junk = (np.arange(n, dtype=np.int32) * 7919) % 300 - 50
    engine._chk(engine.lib.asm_memcpy_h2d(engine.h, d_hint, junk.ctypes.data, 4 * n))
    engine.align_hinted_async(batch, asm.LEAP, params, d_hint, d_leap)
    _check("hinted by junk", engine.to_host(d_leap, n), want, hb)
    engine.memset_async(d_leap, 0xff, 4 * n)
    engine.run_benchmark_async(batch, params, None, d_leap, d_greedy, None, repack=True)  # no NW: Greedy first, LEAP sorted by it
    _check("run_benchmark leap", engine.to_host(d_leap, n), want, hb)
    _check("run_benchmark greedy", engine.to_host(d_greedy, n), oracle.greedy(hb, k, x, o, e, mode=1), hb)
    for d in (d_hint, d_leap, d_greedy):
        engine.free(d)


def test_seed_hit_batches_mapper_shape(asm, engine, oracle):
    """The reference mapper's call shape (GASMA/mapper/main.cpp:67-96): resident reference text, one Greedy alignment per
    seed hit against reference[start, start + len + 1), start = pos ? pos - 1 : 0; MAPQ = 60 + cost."""
    rng = np.random.default_rng(11)
    genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 200_000)]
    n, L = 5000, 100
    pos = rng.integers(0, genome.size - L, n).astype(np.uint64)
    pos[:4] = [0, 1, genome.size - L, genome.size - L - 1]  # both ends of the reference
    reads = np.empty((n, L), np.uint8)
    for i in range(n):
        r = genome[int(pos[i]):int(pos[i]) + L].copy()
        for _ in range(int(rng.integers(0, 5))):
            r[int(rng.integers(0, L))] = b"ACGT"[int(rng.integers(0, 4))]
        reads[i] = r
    read_off = (np.arange(n + 1) * L).astype(np.uint32)
    ref = engine.upload_reference(genome.tobytes())
    batch = engine.batch_from_hits(ref, reads.reshape(-1), read_off, pos)
    # the same pairs built explicitly on the host
    pairs = []
    for i in range(n):
        start = int(pos[i]) - 1 if pos[i] else 0
        pairs.append((reads[i].tobytes().decode(), genome[start:start + L + 1].tobytes().decode()))
    hb = asm.HostBatch.from_strings(pairs)
    got = batch.download()
    assert np.array_equal(got.refs, hb.refs) and np.array_equal(got.ref_off, hb.ref_off)
    params = asm.Params.default(k=3)
    cost = engine.align(batch, asm.GREEDY, params)
    _check("mapper greedy", cost, oracle.greedy(hb, k=3, mode=1), hb)
    mapq = 60 + cost
    assert mapq.min() >= 60


def test_pack_maps_every_non_base_byte_to_code_00(asm, engine, oracle):
    """Only exact 'C','G','T' set plane bits; 'N', lower case, punctuation, 0x01..0xff all collapse onto 'A'
    (bit_convert.cpp:340-355).  Checked through Greedy, whose oracle applies the same rule byte by byte."""
    rng = np.random.default_rng(3)
    pool = np.array(list(b"ACGTACGTACGTNnacgt-*BDEFHU@") + [1, 2, 0x42, 0x44, 0x53, 0x55, 0x7f, 0xc3, 0xd4, 0xff], np.uint8)
    n = 4000
    lens_a = rng.integers(1, 200, n)
    lens_b = np.clip(lens_a + rng.integers(-3, 4, n), 1, None)
    reads = pool[rng.integers(0, pool.size, int(lens_a.sum()))]
    refs = pool[rng.integers(0, pool.size, int(lens_b.sum()))]
    ro = np.zeros(n + 1, np.uint32); ro[1:] = np.cumsum(lens_a)
    fo = np.zeros(n + 1, np.uint32); fo[1:] = np.cumsum(lens_b)
    # make the pairs related so that Greedy takes real steps
    for i in range(n):
        k = min(lens_a[i], lens_b[i])
        refs[fo[i]:fo[i] + k] = reads[ro[i]:ro[i] + k]
        for _ in range(3):
            refs[fo[i] + int(rng.integers(0, k))] = pool[int(rng.integers(0, pool.size))]
    hb = asm.HostBatch(reads, ro, refs, fo)
    for mode in (asm.GREEDY_CLEAN, asm.GREEDY_SEQUENTIAL):
        got = engine.align(engine.upload(hb, mode), asm.GREEDY, asm.Params.default(k=3))
        _check("greedy on dirty alphabet", got, oracle.greedy(hb, k=3, mode=mode), hb)


def test_gpu_against_the_real_reference(asm, engine):
    """Where the compiled reference travelled with the repo (oracle/_ref), compare the HIP path with IT directly —
    Greedy cost + CIGAR in both buffer modes and LEAP's get_ED — on the pairs the reference defines."""
    from tests import oracle_binding
    from tests.util import greedy_defined, leap_defined

    if not oracle_binding.have_reference():
        pytest.skip("oracle/_ref/libasm_ref.so not present")
    ref = oracle_binding.load_reference()
    for wl, n, k in (("C2", 20000, 3), ("C5", 6000, 3), ("C3", 1500, 30)):
        cfg, _, _ = asm.workload(wl)
        hb = asm.generate_pairs(cfg, 1234, n)
        params = asm.Params.default(k=k)
        gd, ld = greedy_defined(hb, k), leap_defined(hb)
        for mode in (asm.GREEDY_SEQUENTIAL, asm.GREEDY_CLEAN):
            batch = engine.upload(hb, mode)
            cost, cig, _ = engine.greedy_with_cigar(batch, params, cap=96)
            rcost, rcig = ref.greedy(hb, k=k, mode=mode, cigars=True)
            assert np.array_equal(cost[gd], rcost[gd]), (wl, mode)
            assert all(a == b for a, b, d in zip(cig, rcig, gd) if d), (wl, mode, "CIGAR")
        leap = engine.align(batch, asm.LEAP, params)
        assert np.array_equal(leap[ld], ref.leap(hb, k=k)[ld]), wl


def test_pymatch_named_classes(asm, oracle):
    """pymatch-style call shape (two strings -> editDistance()) with the C++ harness's numbers."""
    from approximate_string_matching_amd import pymatch_like as pm
    from tests.util import KNOWN_PAIRS

    a, b = KNOWN_PAIRS["KA-0"]
    assert pm.NeedlemanWunsch(a, b).editDistance() == 5
    assert pm.LEAP(a, b, 3, 200).editDistance() == 5
    assert pm.GASMA(a, b, 3).editDistance() == 6
    a5, b5 = KNOWN_PAIRS["KA-5"]
    assert (pm.NeedlemanWunsch(a5, b5).editDistance(), pm.LEAP(a5, b5, 2, 10).editDistance()) == (3, 2)
    cfg, _, _ = asm.workload("C1")
    hb = asm.generate_pairs(cfg, 0, 500)
    pairs = [hb.pair(i) for i in range(hb.n)]
    assert np.array_equal(pm.batch_edit_distances(pm.GASMA, pairs, k=3), oracle.greedy(hb, k=3, mode=1))


# ---- filtering stage: bit-parallel LEAP (SIMD_ED) and SHD -------------------------------------------------
@pytest.mark.parametrize("wl,n", [("C1", 6000), ("C2", 20000), ("C4", 6000), ("C5", 12000)])
@pytest.mark.parametrize("ed_t,shd", [(1, True), (2, False), (3, True), (3, False), (5, True), (8, False), (10, True),
                                       (16, True), (20, False)])
def test_simd_ed_filter_matches_oracle(asm, engine, oracle, wl, n, ed_t, shd):
    """SIMD_ED (Levenshtein, ED_GLOBAL) per pair: register-resident lanes (T <= 8) and the run-time-T kernel, with and
    without the SHD pre-filter, in sequential mode (verdict state carried in batch order from the reference harness's
    warm-up state) and in clean mode."""
    from tests.oracle_binding import SIMD_WARM_STATE
    cfg, _, _ = asm.workload(wl)
    hb = asm.generate_pairs(cfg, 41, n)
    batch = engine.upload(hb, asm.GREEDY_CLEAN)
    for mode in (asm.FILTER_SEQUENTIAL, asm.FILTER_CLEAN):
        want, _, want_pass = oracle.simd_ed(hb, ed_t, shd, mode, SIMD_WARM_STATE)
        got = engine.simd_ed(batch, ed_t, shd, mode, SIMD_WARM_STATE)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (wl, ed_t, shd, mode, bad[:5], got[bad[:5]], want[bad[:5]])
        assert ((got >= 0) == (want_pass == 1)).all()


@pytest.mark.parametrize("wl,n", [("C1", 6000), ("C2", 12000), ("C4", 6000), ("C5", 9000), ("C3", 4000)])
@pytest.mark.parametrize("setting", [(3, 60, 2, 3, 1), (6, 30, 1, 1, 1), (12, 120, 4, 6, 2), (2, 25, 3, 5, 2), (32, 200, 15, 15, 15),
                                     (20, 40, 1, 2, 1)])
def test_simd_ed_affine_filter_matches_oracle(asm, engine, oracle, wl, n, setting):
    """SIMD_ED affine mode (init_affine / run_affine), clean: every pair from init_affine's tables; narrow and wide bands, deep
    rings (x = o = e = 15: fewer threads per workgroup), thresholds that reject."""
    g, af, x, o, e = setting
    cfg, _, _ = asm.workload(wl)
    hb = asm.generate_pairs(cfg, 43, n)
    batch = engine.upload(hb, asm.GREEDY_CLEAN)
    want, want_pass = oracle.simd_ed_affine(hb, g, af, x, o, e)
    got = engine.simd_ed_affine(batch, g, af, x, o, e)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (wl, setting, bad[:5], got[bad[:5]], want[bad[:5]])


@pytest.mark.parametrize("wl,n", [("C2", 12000), ("C5", 9000), ("C3", 4000)])
@pytest.mark.parametrize("setting", [(3, 60, 2, 3, 1, 3), (6, 30, 1, 1, 1, 2), (12, 120, 4, 6, 2, 5), (16, 90, 2, 3, 1, 16), (20, 40, 1, 2, 1, 0)])
def test_simd_ed_affine_filter_with_shd_matches_oracle(asm, engine, oracle, wl, n, setting):
    """init_affine(..., SHD_enable = true, SHD_threshold): the mask-array SHD in front of run_affine, over the first
    2*SHD_threshold+1 lane masks (centred on the main lane only when the thresholds are equal); both kernel forms (four threads
    per pair up to 128 characters, thread per pair beyond) and argument errors."""
    g, af, x, o, e, st = setting
    cfg, _, _ = asm.workload(wl)
    hb = asm.generate_pairs(cfg, 47, n)
    batch = engine.upload(hb, asm.GREEDY_CLEAN)
    want, _ = oracle.simd_ed_affine(hb, g, af, x, o, e, shd_t=st)
    got = engine.simd_ed_affine(batch, g, af, x, o, e, shd_threshold=st)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (wl, setting, bad[:5], got[bad[:5]], want[bad[:5]])
    for bad_t in (-1, g + 1, 17):
        if bad_t <= g and 0 <= bad_t <= 16:
            continue
        with pytest.raises(asm.AsmError):
            engine.simd_ed_affine(batch, g, af, x, o, e, shd_threshold=bad_t)


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("wl,n", [("C2", 6000), ("C5", 4000), ("C3", 1500)])
def test_leap_ed_modes(asm, engine, oracle, wl, n, mode):
    """asm_params.leap_mode = LV::init's ED_modes LOCAL / SEMI_FREE_BEGIN / SEMI_FREE_END (no caller in the reference; one
    kernel serves them): narrow and wide bands, unit and general penalties, mixed lengths; and the argument check."""
    cfg, _, _ = asm.workload(wl)
    hb = asm.generate_pairs(cfg, 53, n)
    batch = engine.upload(hb, asm.GREEDY_CLEAN)
    ok = np.maximum(*hb.lengths()) <= 256
    for k, x, o, e in ((3, 1, 1, 1), (5, 2, 3, 1), (12, 1, 1, 1), (8, 4, 6, 2), (30, 1, 2, 1)):
        want = oracle.leap(hb, k, x, o, e, mode)
        got = engine.align(batch, asm.LEAP, asm.Params.default(k=k, x=x, o=o, e=e, leap_mode=mode))
        bad = np.nonzero((got != want) & ok)[0]
        assert bad.size == 0, (wl, mode, k, x, o, e, bad[:5], got[bad[:5]], want[bad[:5]])
    with pytest.raises(asm.AsmError):
        engine.align(batch, asm.LEAP, asm.Params.default(k=3, leap_mode=4))
    # GLOBAL is untouched by the field's new meaning
    assert np.array_equal(engine.align(batch, asm.LEAP, asm.Params.default(k=3, leap_mode=asm.LEAP_GLOBAL))[ok], oracle.leap(hb, 3)[ok])


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("wl,n", [("C2", 8000), ("C5", 6000), ("C3", 3000)])
def test_simd_ed_affine_filter_ed_modes(asm, engine, oracle, wl, n, mode):
    """init_affine's ED_modes (LOCAL / SEMI_FREE_BEGIN / SEMI_FREE_END) in both kernel forms, with and without the SHD pre-filter;
    get_ED() is final_ED in LOCAL and SEMI_FREE_END (0 for a pair exact at generation 0), converge_ED in the other two."""
    cfg, _, _ = asm.workload(wl)
    hb = asm.generate_pairs(cfg, 59, n)
    batch = engine.upload(hb, asm.GREEDY_CLEAN)
    for g, af, x, o, e, st in ((3, 60, 2, 3, 1, None), (6, 30, 1, 1, 1, 2), (12, 120, 4, 6, 2, None), (20, 40, 1, 2, 1, 5), (32, 200, 15, 15, 15, None)):
        want, _ = oracle.simd_ed_affine(hb, g, af, x, o, e, shd_t=st, mode=mode)
        got = engine.simd_ed_affine(batch, g, af, x, o, e, shd_threshold=st, mode=mode)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (wl, mode, g, af, x, o, e, st, bad[:5], got[bad[:5]], want[bad[:5]])
    with pytest.raises(asm.AsmError):
        engine.simd_ed_affine(batch, 3, 60, 2, 3, 1, mode=4)


@pytest.mark.parametrize("ed_mode", [1, 2, 3])
@pytest.mark.parametrize("wl,n", [("C2", 8000), ("C4", 6000), ("C5", 6000)])
def test_simd_ed_levenshtein_ed_modes(asm, engine, oracle, wl, n, ed_mode):
    """init_levenshtein's ED_modes (LOCAL / SEMI_FREE_BEGIN / SEMI_FREE_END): sequential (state carried, as the reference runs)
    and clean, with and without SHD, register-sized and larger thresholds; the state a sequential call returns."""
    from tests.oracle_binding import SIMD_WARM_STATE
    cfg, _, _ = asm.workload(wl)
    hb = asm.generate_pairs(cfg, 61, n)
    batch = engine.upload(hb, asm.GREEDY_CLEAN)
    for t, shd in ((3, True), (5, False), (12, True), (20, False)):
        for fmode in (asm.FILTER_SEQUENTIAL, asm.FILTER_CLEAN):
            want, _, _ = oracle.simd_ed(hb, t, shd, fmode, SIMD_WARM_STATE, ed_mode=ed_mode)
            got = engine.simd_ed(batch, t, shd, fmode, SIMD_WARM_STATE, ed_mode=ed_mode)
            bad = np.nonzero(got != want)[0]
            assert bad.size == 0, (wl, ed_mode, t, shd, fmode, bad[:5], got[bad[:5]], want[bad[:5]])
    with pytest.raises(asm.AsmError):
        engine.simd_ed(batch, 3, True, asm.FILTER_CLEAN, None, ed_mode=7)


def test_filters_ignore_the_stale_tails_of_sequential_batches(asm, engine, oracle):
    """A batch packed for Greedy's sequential mode keeps the reference's stale buffer tails beyond each string's end; NW, LEAP
    and the three filters must not see them (mixed lengths: long pairs leave long tails for the short ones that follow)."""
    cfg, _, _ = asm.workload("C5")
    hb = asm.generate_pairs(cfg, 3, 8000)
    batch = engine.upload(hb, asm.GREEDY_SEQUENTIAL)
    want, _, _ = oracle.simd_ed(hb, 5, True, asm.FILTER_CLEAN, (0, 0, 0))
    assert np.array_equal(engine.simd_ed(batch, 5, True, asm.FILTER_CLEAN), want)
    want, _ = oracle.simd_ed_affine(hb, 6, 80, 2, 3, 1)
    assert np.array_equal(engine.simd_ed_affine(batch, 6, 80, 2, 3, 1), want)
    assert np.array_equal(engine.shd_filter(batch, 5), oracle.shd(hb, 5))


def test_simd_ed_affine_filter_edges_and_reference(asm, engine, oracle):
    """Ragged and empty strings, reads beyond 256 characters, argument errors; and the compiled reference itself (run_affine with
    init_affine before every pair) where oracle/_ref travelled with the snapshot."""
    from tests import oracle_binding as ob
    from tests.util import random_ragged_batch
    hb = random_ragged_batch(asm, 9, 2000, 0, 300, err=0.2)
    batch = engine.upload(hb, asm.GREEDY_CLEAN)
    for g, af, x, o, e in ((4, 50, 2, 3, 1), (9, 90, 3, 4, 2)):
        want, _ = oracle.simd_ed_affine(hb, g, af, x, o, e)
        assert np.array_equal(engine.simd_ed_affine(batch, g, af, x, o, e), want), (g, af)
    for bad in ((0, 10, 1, 1, 1), (33, 10, 1, 1, 1), (3, 0, 1, 1, 1), (3, 600, 1, 1, 1), (3, 10, 1, 1, 2), (3, 10, 0, 1, 1), (3, 10, 1, 16, 1)):
        with pytest.raises(asm.AsmError):
            engine.simd_ed_affine(batch, *bad)
    if ob.have_reference_simd():
        ref = ob.load_reference_simd()
        cfg, _, _ = asm.workload("C2")
        hb2 = asm.generate_pairs(cfg, 61, 8000)
        b2 = engine.upload(hb2, asm.GREEDY_CLEAN)
        r_ed, r_ps = ref.simd_ed_affine(hb2, 5, 22, 2, 3, 1)
        got = engine.simd_ed_affine(b2, 5, 22, 2, 3, 1)
        assert ((got >= 0) == (r_ps == 1)).all() and (got[r_ps == 1] == r_ed[r_ps == 1]).all()
        assert 0.05 < r_ps.mean() < 0.999


@pytest.mark.parametrize("max_error", [0, 1, 3, 5, 9, 16])
def test_shd_filter_matches_oracle(asm, engine, oracle, max_error):
    for wl, n in (("C1", 5000), ("C2", 20000), ("C5", 12000)):
        cfg, _, _ = asm.workload(wl)
        hb = asm.generate_pairs(cfg, 43, n)
        batch = engine.upload(hb, asm.GREEDY_CLEAN)
        got, want = engine.shd_filter(batch, max_error), oracle.shd(hb, max_error)
        assert (got == want).all(), (wl, max_error, int((got != want).sum()))


def test_filters_on_ragged_and_dirty_input(asm, engine, oracle):
    """Empty strings, lengths around 64/128/256 (the 128-bit halves of S1), reads longer than 256 (cut at _MAX_LENGTH_),
    references shorter and longer than the read, and bytes outside ACGT."""
    from tests.oracle_binding import SIMD_WARM_STATE
    hb = random_ragged_batch(asm, 47, 4000, 0, 300)
    rng = np.random.default_rng(5)
    reads = hb.reads.copy()
    hits = rng.random(reads.size) < 0.01
    reads[hits] = rng.choice(np.frombuffer(b"NnacgtRY-*", np.uint8), int(hits.sum()))
    hb = asm.HostBatch(reads, hb.read_off, hb.refs, hb.ref_off)
    batch = engine.upload(hb, asm.GREEDY_CLEAN)
    for ed_t, shd in ((3, True), (7, False), (12, True)):
        for mode in (asm.FILTER_SEQUENTIAL, asm.FILTER_CLEAN):
            want, _, _ = oracle.simd_ed(hb, ed_t, shd, mode, SIMD_WARM_STATE)
            got = engine.simd_ed(batch, ed_t, shd, mode, SIMD_WARM_STATE)
            assert (got == want).all(), (ed_t, shd, mode, int((got != want).sum()))
    for me in (2, 6):
        assert (engine.shd_filter(batch, me) == oracle.shd(hb, me)).all()


def test_filter_argument_errors(asm, engine):
    cfg, _, _ = asm.workload("C1")
    batch = engine.generate(cfg, 0, 64)
    d = engine.malloc(4 * 64)
    for args in ((0, True), (33, False), (17, True)):
        with pytest.raises(asm.AsmError):
            engine.simd_ed_async(batch, args[0], d, args[1])
    with pytest.raises(asm.AsmError):
        engine.shd_filter_async(batch, 17, d)
    with pytest.raises(asm.AsmError):
        engine.simd_ed_async(batch, 3, d, True, 7)
    engine.free(d)


def test_filters_against_the_real_reference(asm, engine):
    """The compiled SIMD_ED / SHD sources (oracle/_ref/libasm_ref_simd.so, built in the dev container, travels with the
    snapshot): verdicts and get_ED() of passing pairs, run in batch order after the harness's warm-up pair."""
    from tests import oracle_binding as ob
    if not ob.have_reference_simd():
        pytest.skip("oracle/_ref/libasm_ref_simd.so not built")
    ref = ob.load_reference_simd()
    for wl, n in (("C2", 20000), ("C5", 10000)):
        cfg, _, _ = asm.workload(wl)
        hb = asm.generate_pairs(cfg, 53, n)
        batch = engine.upload(hb, asm.GREEDY_CLEAN)
        for ed_t, shd in ((3, True), (5, False), (12, True)):
            r_ed, r_pass = ref.simd_ed(hb, ed_t, shd)
            got = engine.simd_ed(batch, ed_t, shd, asm.FILTER_SEQUENTIAL, ob.SIMD_WARM_STATE)
            assert ((got >= 0) == (r_pass == 1)).all(), (wl, ed_t, shd)
            assert (got[r_pass == 1] == r_ed[r_pass == 1]).all(), (wl, ed_t, shd)
        for me in (3, 8):
            assert (engine.shd_filter(batch, me) == ref.shd(hb, me)).all(), (wl, me)


def test_filter_state_chains_across_chunks_and_stdin_driver(asm, engine, oracle, tmp_path):
    """A file filtered in chunks (BATCH_RUN, LEAP_SIMD/main.cpp:20,104-140) with the state handed from chunk to chunk gives
    the verdicts of one pass over the whole file; asm-bench --leap-simd is that driver (passNum / totalNum lines)."""
    import os
    import subprocess

    cfg, _, _ = asm.workload("C5")
    hb = asm.generate_pairs(cfg, 59, 9000)
    for ed_t, shd in ((3, True), (6, False)):
        want, _, ps = oracle.simd_ed(hb, ed_t, shd, 0, (0, 0, 0))
        state, got = (0, 0, 0), []
        for lo in range(0, hb.n, 2500):
            part = hb.slice(lo, min(hb.n, lo + 2500))
            b = engine.upload(part, asm.GREEDY_CLEAN)
            d = engine.malloc(4 * part.n)
            state = engine.simd_ed_async(b, ed_t, d, shd, asm.FILTER_SEQUENTIAL, state)
            got.append(engine.to_host(d, part.n))
            engine.free(d)
        assert (np.concatenate(got) == want).all(), (ed_t, shd)
        text = "".join("%s\n%s\n" % hb.pair(i) for i in range(hb.n)) + "end_of_file\n"
        exe = os.path.join(os.path.dirname(asm.LIB_PATH), "asm-bench")
        out = subprocess.run([exe, "--leap-simd", str(ed_t), "--shd", "1" if shd else "0", "--batch-run", "2000"],
                             input=text, capture_output=True, text=True, check=True).stdout.splitlines()
        assert out[0] == "passNum:\t%d" % int(ps.sum()) and out[1] == "totalNum:\t%d" % hb.n, out[:3]


@pytest.mark.parametrize("wl,n,k,pen", [("C2", 20000, 3, (1, 1, 1)), ("C5", 8000, 3, (2, 3, 1)), ("C3", 2500, 30, (1, 1, 1)),
                                        ("C2", 3000, 10, (4, 6, 2)), ("C2", 2000, 40, (1, 1, 1)), ("C1", 4000, 5, (1, 2, 1))])
def test_greedy_semi_global(asm, engine, oracle, wl, n, k, pen):
    """asm_params.alignment_type = SEMI_GLOBAL (hurdle_matrix's constructor argument): every Greedy kernel family, cost and
    CIGAR, both tail modes."""
    cfg, _, _ = asm.workload(wl)
    hb = asm.generate_pairs(cfg, 17, n)
    x, o, e = pen
    params = asm.Params.default(k=k, x=x, o=o, e=e, alignment_type=asm.ALIGN_SEMI_GLOBAL)
    for mode in (asm.GREEDY_CLEAN, asm.GREEDY_SEQUENTIAL):
        batch = engine.upload(hb, mode)
        want, want_cig = oracle.greedy(hb, k, x, o, e, mode=mode, cigars=True, semi=True)
        _check(f"semi greedy mode={mode}", engine.align(batch, asm.GREEDY, params), want, hb)
        cost, cig, _ = engine.greedy_with_cigar(batch, params, cap=96)
        _check("semi cigar cost", cost, want, hb)
        assert cig == want_cig
    with pytest.raises(asm.AsmError):
        engine.align(batch, asm.GREEDY, asm.Params.default(k=k, alignment_type=2))


@pytest.mark.parametrize("wl,n", [("C2", 30000), ("C5", 12000)])
def test_pipelined_repack_gives_the_same_results(asm, engine, oracle, wl, n):
    """asm_run_benchmark_async with repack = 2: the pack of call s+1 fills a second set of planes on its own stream while the
    aligners of call s still read the first.  Many back-to-back calls without a sync in between (so that they really overlap),
    mixed with in-order repacks and plain runs: penalties and counters equal the oracle's every time."""
    cfg, _, params = asm.workload(wl)
    hb = asm.generate_pairs(cfg, 13, n)
    batch = engine.upload(hb, asm.GREEDY_SEQUENTIAL)
    d = [engine.malloc(4 * n) for _ in range(3)]
    d_cnt = engine.malloc(32)
    engine.memset_async(d_cnt, 0, 32)
    calls = [2, 2, 2, 1, 2, 0, 2, 2, 1, 1, 2]
    for mode in calls:
        engine.run_benchmark_async(batch, params, d[0], d[1], d[2], d_cnt, repack=mode)
    nw, leap, greedy = oracle.nw(hb), oracle.leap(hb, params.k), oracle.greedy(hb, params.k, mode=0)
    ok = np.maximum(*hb.lengths()) <= 256
    assert np.array_equal(engine.to_host(d[0], n), nw)
    assert np.array_equal(engine.to_host(d[1], n)[ok], leap[ok])
    assert np.array_equal(engine.to_host(d[2], n), greedy)
    cnt = engine.to_host(d_cnt, 4, np.uint64)
    got_leap = engine.to_host(d[1], n)
    assert cnt.tolist() == [len(calls) * n, len(calls) * n, len(calls) * int((got_leap == nw).sum()), len(calls) * int((greedy == nw).sum())]
    # the batch is still a normal batch afterwards
    assert np.array_equal(engine.align(batch, asm.GREEDY, params), greedy)
    for x in d + [d_cnt]:
        engine.free(x)


@pytest.mark.parametrize("wl,n", [("C2", 30000), ("C5", 12000)])
def test_overlapped_calls_give_the_same_results(asm, engine, oracle, wl, n):
    """asm_run_benchmark_async with repack = 3: consecutive calls overlap (no call waits for the previous call's Greedy, the
    counters run on their own stream), the caller alternates two sets of output arrays and joins at the end.  Two DIFFERENT
    batches take turns, so that a kernel running in the wrong order, or counters reading arrays a later call already rewrites,
    would show: both output sets must hold the penalties of the batch that wrote them last and the counters the exact sum."""
    cfg, _, params = asm.workload(wl)
    hbs = [asm.generate_pairs(cfg, 13, n), asm.generate_pairs(cfg, 977, n)]
    batches = [engine.upload(hb, asm.GREEDY_CLEAN) for hb in hbs]
    sets = [[engine.malloc(4 * n) for _ in range(3)] for _ in range(2)]
    d_cnt = engine.malloc(32)
    engine.memset_async(d_cnt, 0, 32)
    want = []
    for hb in hbs:
        want.append((oracle.nw(hb), oracle.leap(hb, params.k), oracle.greedy(hb, params.k, mode=1)))
    ok = [np.maximum(*hb.lengths()) <= 256 for hb in hbs]
    order = [0, 1, 1, 0, 0, 0, 1, 0, 1, 1, 1, 0, 1]
    for c, which in enumerate(order):
        o = sets[c & 1]
        engine.run_benchmark_async(batches[which], params, o[0], o[1], o[2], d_cnt, repack=3)
    engine.pipeline_join_async()
    last = {(len(order) - 1) & 1: order[-1], (len(order) - 2) & 1: order[-2]}
    expect = np.zeros(4, np.int64)
    got_leap = {}
    for s_i, which in last.items():
        nw, leap, greedy = want[which]
        assert np.array_equal(engine.to_host(sets[s_i][0], n), nw)
        got_leap[which] = engine.to_host(sets[s_i][1], n)
        assert np.array_equal(got_leap[which][ok[which]], leap[ok[which]])
        assert np.array_equal(engine.to_host(sets[s_i][2], n), greedy)
    for which in order:
        nw, leap, greedy = want[which]
        gl = got_leap.get(which)
        if gl is None:  # (both batches are among the last two calls in `order`)
            gl = leap
        expect += np.array([n, n, int((gl == nw).sum()), int((greedy == nw).sum())])
    assert engine.to_host(d_cnt, 4, np.uint64).tolist() == expect.tolist()
    # a plain call afterwards joins by itself, and the batches are still normal batches
    engine.run_benchmark_async(batches[0], params, sets[0][0], sets[0][1], sets[0][2], d_cnt, repack=1)
    assert np.array_equal(engine.to_host(sets[0][2], n), want[0][2])
    assert np.array_equal(engine.align(batches[1], asm.GREEDY, params), want[1][2])
    # ... and a new run of overlapped calls may start with either set
    for c in range(3):
        o = sets[(c + 1) & 1]
        engine.run_benchmark_async(batches[1], params, o[0], o[1], o[2], d_cnt, repack=3)
    # the same output arrays twice in a row is refused (the previous call's counters may still read them) ...
    with pytest.raises(asm.AsmError):
        engine.run_benchmark_async(batches[1], params, o[0], o[1], o[2], d_cnt, repack=3)
    engine.pipeline_join_async()  # ... until the caller has joined
    engine.run_benchmark_async(batches[1], params, o[0], o[1], o[2], d_cnt, repack=3)
    engine.synchronize()  # waits for the overlapped calls too
    assert np.array_equal(engine.to_host(sets[0][0], n), want[1][0])
    assert np.array_equal(engine.to_host(sets[1][2], n), want[1][2])
    for x in sets[0] + sets[1] + [d_cnt]:
        engine.free(x)


def test_overlapped_calls_followed_by_a_call_that_cannot_overlap(asm, engine, oracle):
    """A run of overlapped calls (repack = 3) and then, WITHOUT a join, a call asking for the same form that cannot have it: the
    Greedy-first shape of config 3 (LEAP + Greedy without NW at a wide band: LEAP is scheduled by Greedy's penalties) and an empty
    batch.  The library runs those as pipelined-pack calls behind everything enqueued before; the misuse guard starts afresh; the
    counters of both kinds of call add up exactly (total_tests counts without NW too)."""
    cfg2, _, p2 = asm.workload("C2")
    cfg3, _, p3 = asm.workload("C3")
    n2, n3 = 40_000, 6_000
    hb2, hb3 = asm.generate_pairs(cfg2, 5, n2), asm.generate_pairs(cfg3, 6, n3)
    b2, b3 = engine.upload(hb2, asm.GREEDY_CLEAN), engine.upload(hb3, asm.GREEDY_CLEAN)
    empty = engine.upload(asm.HostBatch.from_strings([]), asm.GREEDY_CLEAN)
    sets = [[engine.malloc(4 * n2) for _ in range(3)] for _ in range(2)]
    o3 = [engine.malloc(4 * n3) for _ in range(2)]
    d_cnt = engine.malloc(32)
    engine.memset_async(d_cnt, 0, 32)
    for c in range(5):
        o = sets[c & 1]
        engine.run_benchmark_async(b2, p2, o[0], o[1], o[2], d_cnt, repack=3)
    engine.run_benchmark_async(b3, p3, None, o3[0], o3[1], d_cnt, repack=3)      # Greedy first: cannot overlap
    engine.run_benchmark_async(empty, p2, sets[0][0], sets[0][1], sets[0][2], d_cnt, repack=3)   # nothing to do
    # the arrays of the last overlapped call may be named again at once: the fall-back calls reset the guard
    engine.run_benchmark_async(b2, p2, sets[0][0], sets[0][1], sets[0][2], d_cnt, repack=3)
    engine.synchronize()
    nw, leap, greedy = oracle.nw(hb2), oracle.leap(hb2, p2.k), oracle.greedy(hb2, p2.k, mode=1)
    assert np.array_equal(engine.to_host(sets[0][0], n2), nw) and np.array_equal(engine.to_host(sets[0][1], n2), leap)
    assert np.array_equal(engine.to_host(sets[0][2], n2), greedy) and np.array_equal(engine.to_host(sets[1][2], n2), greedy)
    assert np.array_equal(engine.to_host(o3[0], n3), oracle.leap(hb3, p3.k))
    assert np.array_equal(engine.to_host(o3[1], n3), oracle.greedy(hb3, p3.k, mode=1))
    per = np.array([n2, n2, int((leap == nw).sum()), int((greedy == nw).sum())])
    assert engine.to_host(d_cnt, 4, np.uint64).tolist() == (6 * per + np.array([n3, 0, 0, 0])).tolist()
    for x in sets[0] + sets[1] + o3 + [d_cnt]:
        engine.free(x)


def test_profile_events_inside_run_benchmark(asm, engine):
    """asm_profile_enable / asm_profile_read: per-kernel HIP events recorded by the library inside asm_run_benchmark_async
    (what bench.py uses for the dominant kernel's duration inside its timed region)."""
    cfg, _, params = asm.workload("C2")
    n = 200_000
    batch = engine.generate(cfg, 0, n)
    d = [engine.malloc(4 * n) for _ in range(3)]
    d_cnt = engine.malloc(32)
    engine.memset_async(d_cnt, 0, 32)
    for mask, want in ((0xF, [True] * 4), (0x8, [False, False, False, True])):
        engine.profile_enable(3, mask)
        for _ in range(5):  # only the first three calls are recorded
            engine.run_benchmark_async(batch, params, d[0], d[1], d[2], d_cnt, repack=True)
        ms = engine.profile_read(8)
        assert ms.shape == (3, 4)
        for q in range(4):
            assert ((ms[:, q] > 0) & (ms[:, q] < 50)).all() if want[q] else (ms[:, q] == -1).all(), (mask, q, ms)
    engine.profile_enable(0, 0)
    engine.run_benchmark_async(batch, params, d[0], d[1], d[2], d_cnt, repack=True)
    assert engine.profile_read(8).shape[0] == 0
    for x in d + [d_cnt]:
        engine.free(x)


def test_per_pair_classes_of_the_reference(asm, oracle):
    """host/asm_compat.hpp's hurdle_matrix (reset/run/get_cost/get_CIGAR), LV and SIMD_ED on single pairs, through
    `asm-bench --pair`: the demo pair of GASMA/main.cpp:7-8 gives 22M1D50M1D28M at cost 6; everything equals the oracle."""
    import json
    import os
    import subprocess

    exe = os.path.join(os.path.dirname(asm.LIB_PATH), "asm-bench")
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "index.json")))["known_answers"]
    for key in ("KA-0", "KA-5", "KA-6"):
        a, b = KNOWN_PAIRS[key]
        out = subprocess.check_output([exe, "--pair", a, b, "--k", "3"], text=True).splitlines()
        hb = asm.HostBatch.from_strings([(a, b)])
        cost, cig = oracle.greedy(hb, k=3, mode=1, cigars=True)
        assert out[0] == "greedy cost %d CIGAR %s" % (int(cost[0]), cig[0]), (key, out[0])
        if key == "KA-0":
            assert out[0] == "greedy cost 6 CIGAR 22M1D50M1D28M" == "greedy cost %d CIGAR %s" % (gold[key]["greedy_k3"], gold[key]["greedy_cigar_k3"])
        leap = int(oracle.leap(hb, k=3)[0])
        assert out[1] == "leap pass %d ED %d" % (1 if leap >= 0 else 0, leap), (key, out[1])
        ed, _, ps = oracle.simd_ed(hb, 3, True, 0, (0, 0, 0))
        assert out[2] == "simd_ed pass %d ED %d" % (int(ps[0]), int(ed[0])), (key, out[2])
        aed, aps = oracle.simd_ed_affine(hb, 3, 200, 1, 1, 1)  # SIMD_ED::init_affine(k, 200, ED_GLOBAL, x, o, e), clean
        assert out[3] == "simd_ed affine pass %d ED %d" % (int(aps[0]), int(aed[0])), (key, out[3])


def test_batches_may_outlive_their_engine(asm):
    """Device blocks of a batch belong to its handle's pool: destroying the handle first (Python's garbage collector does, at
    interpreter exit) must leave the batch record freeable, and a second engine must work afterwards."""
    cfg, _, params = asm.workload("C2")
    eng = asm.Engine(0)
    b1, b2 = eng.generate(cfg, 0, 5000), eng.upload(asm.generate_pairs(cfg, 0, 100))
    d = eng.malloc(4 * 5000)
    eng.align_async(b1, asm.GREEDY, params, d)
    eng.close()
    b1.free(), b2.free()
    eng2 = asm.Engine(0)
    assert eng2.align(eng2.generate(cfg, 0, 100), asm.NW, params).shape == (100,)
    eng2.close()
    # A stale batch freed while a NEW engine is alive — which the allocator likes to put at the destroyed engine's address: the
    # batch names its owner by (address, serial), so its release must not reach into the new engine's pool.
    for _ in range(4):
        a = asm.Engine(0)
        stale = a.generate(cfg, 0, 3000)
        a.close()
        b = asm.Engine(0)
        live = b.generate(cfg, 7, 3000)
        want = b.align(live, asm.GREEDY, params)
        stale.free()                      # owner gone: only the record is dropped
        again = b.generate(cfg, 7, 3000)  # would be handed a block of `live` if the stale release had idled one
        assert np.array_equal(b.align(live, asm.GREEDY, params), want)
        assert np.array_equal(b.align(again, asm.GREEDY, params), want)
        b.close()


@pytest.mark.parametrize("k", [1, 2, 3])
def test_fast_greedy_kernel_slow_path_and_corners(asm, engine, oracle, k):
    """The straight-line Greedy kernel (csrc/asm_greedy3.h, k <= 3, unit penalties): solid blocks of mismatches leave the rank
    table's domain (the FP64 slow path), 128/128 pairs hit the lane-0 destination corner, strings shorter than the band have
    negative lane destinations, destinations outside the band rebuild their lane from the planes; with CIGARs, both tail modes."""
    rng = np.random.default_rng(8)
    acgt = "ACGT"
    rnd = lambda L: "".join(acgt[i] for i in rng.integers(0, 4, L))
    pairs = [("A" * 128, "T" * 128), ("A" * 100, "T" * 100), ("A" * 128, "A" * 128), ("", ""), ("A", ""), ("", "A"), ("A", "C"),
             ("AC", "CA"), ("A" * 128, "A" * 120), ("A" * 120, "A" * 128), ("A" * 70 + "C" * 58, "C" * 58 + "A" * 70),
             ("ACGTN" * 20, "ACGTN" * 20), ("A" * 64 + "T" * 64, "T" * 64 + "A" * 64)]
    for L in (128, 127, 100, 65, 64, 63, 10, 3, 2):
        a = rnd(L)
        pairs += [(a, a), (a, a[1:]), (a[1:], a), (a, "T" * 40 + a[40:]), (a, rnd(L)), (a[: L // 2] + "T" * 66, a)]
    for _ in range(3000):
        L = int(rng.integers(1, 200))
        a = list(rnd(L))
        b = list(a)
        for _ in range(int(rng.integers(0, 4))):
            p, w = int(rng.integers(0, L)), int(rng.integers(1, 90))
            b[p:p + w] = list(acgt[(acgt.index(c) + 1) % 4] for c in b[p:p + w])
        if rng.random() < 0.4:
            q = int(rng.integers(0, len(b) + 1))
            b[q:q] = list(rnd(int(rng.integers(1, 6))))
        pairs.append(("".join(a), "".join(b)))
    hb = asm.HostBatch.from_strings(pairs)
    params = asm.Params.default(k=k)
    for mode in (asm.GREEDY_CLEAN, asm.GREEDY_SEQUENTIAL):
        batch = engine.upload(hb, mode)
        want, want_cig = oracle.greedy(hb, k=k, mode=1 if mode == asm.GREEDY_CLEAN else 0, cigars=True)
        assert np.array_equal(engine.align(batch, asm.GREEDY, params), want)
        cost, cig, _ = engine.greedy_with_cigar(batch, params, cap=128)
        assert np.array_equal(cost, want)
        assert cig == want_cig
        batch.free()


def _results_block(stdout):
    """[total, nw %, leap %, greedy %, coverage %] out of the harness's print() block (benchmark_utils.h:391-401)."""
    import re

    total = int(re.search(r"Total number of alignments: (\d+)", stdout).group(1))
    acc = stdout[stdout.index("[Accuracy]"):]
    pct = [float(v) for v in re.findall(r"\| (\d+\.\d+) %", acc)]
    return [total] + pct[:4]


def test_the_reference_mains_run_on_the_library(asm, oracle, tmp_path):
    """The drop-in claim on the reference's own callers, run HERE on the GPU (executables built by oracle/Makefile `shim` in the
    build container, reference sources compiled in place and unmodified):
      * gasma_main_on_shim      GASMA/main.cpp against host/compat: prints what the reference's own build of the same file printed
                                (tests/golden/gasma_main_stdout.txt: the seven lane rows, `22M1D50M1D28M`, `cost: 6`, the LCM string);
      * benchmark_main_on_shim  GASMA/benchmark/benchmark.cpp against host/compat's batched `benchmark` class;
      * ref_harness_on_shim     GASMA/benchmark/benchmark.cpp WITH THE REFERENCE'S OWN benchmark_utils.h (class benchmark, its
                                parasail / LV / hurdle_matrix calls and its coverage check) over host/compat/parasail/parasail.h and
                                the per-pair LV / hurdle_matrix objects.
    benchmark.cpp reads a hard-coded /home/zhenhao/... file: without it both harness executables go through their whole call
    sequence over zero pairs (the compat harness; the reference's class aligns 100000 empty pairs then); with a 2000-pair SRR-shaped (C4) file answered under that name (oracle/_ref/libpath_redirect.so:
    the path is redirected at the C-library boundary, the programs are not touched) both print the [Accuracy] / [Coverage] block
    with exactly the oracle's counters — Greedy in the reference's as-run, order-dependent mode."""
    import subprocess

    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    demo, bench_main = os.path.join(ref_dir, "gasma_main_on_shim"), os.path.join(ref_dir, "benchmark_main_on_shim")
    ref_harness, redirect = os.path.join(ref_dir, "ref_harness_on_shim"), os.path.join(ref_dir, "libpath_redirect.so")
    if not all(os.path.exists(f) for f in (demo, bench_main, ref_harness, redirect)):
        pytest.skip("oracle/_ref/*_on_shim are built only where /root/reference exists")
    out = subprocess.run([demo], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    want = open(os.path.join(ROOT, "tests", "golden", "gasma_main_stdout.txt")).read()
    assert out.stdout == want
    data_path = "/home/zhenhao/dna-align-dataset/SRR611076.data"   # benchmark.cpp:28
    out = subprocess.run([bench_main], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Unable to open data file: " + data_path in out.stdout
    assert "===================== Benchmark Results =====================" in out.stdout
    assert "Total number of alignments: 0" in out.stdout
    # (the reference's own class does not notice the missing file: read_string_file leaves max_tests at 100000 and run() aligns
    # that many pairs of empty strings, benchmark_utils.h:325-352,373-385 — a minute of one-pair device calls; not repeated here)
    # ... and on data
    cfg, _, params = asm.workload("C4")
    n = 2000
    hb = asm.generate_pairs(cfg, 4242, n)
    path = str(tmp_path / "SRR611076.data")
    hb.write_seq_file(path)
    nw, ncig = oracle.nw_cigar(hb)
    leap = oracle.leap(hb, k=3)
    greedy, gcig = oracle.greedy(hb, k=3, mode=0, cigars=True)   # mode 0: the stale-buffer chain of one object over the whole file
    cov = oracle.coverage(hb, gcig, 1, ncig, 3)
    want = [n, 100.0, round(100.0 * float((leap == nw).mean()), 3), round(100.0 * float((greedy == nw).mean()), 3),
            round(100.0 * float(cov.mean()), 3)]
    env = dict(os.environ, LD_PRELOAD=redirect, ASM_REDIRECT_FROM=data_path, ASM_REDIRECT_TO=path)
    for exe in (bench_main, ref_harness):
        out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, (exe, out.stderr[-2000:])
        assert "Processed data file: " + data_path in out.stdout
        assert _results_block(out.stdout) == want, (exe, out.stdout[-1500:], want)


def test_parasail_shim_and_per_pair_objects(asm, oracle, tmp_path):
    """host/compat/parasail/parasail.h and the per-pair hurdle_matrix / LV objects, called the way the reference's harness calls
    them (tests/cxx/compat_probe.cpp, built here with g++): NW penalty and CIGAR, LEAP's get_ED, and Greedy cost and CIGAR of ONE
    object reused for the whole file (the reference's stale-buffer chain) equal the oracle's, pair by pair."""
    import subprocess

    pkg = os.path.join(ROOT, "approximate-string-matching_amd")
    exe = str(tmp_path / "compat_probe")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(pkg, "host", "compat"), "-I", os.path.join(pkg, "host"),
                        "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tests", "cxx", "compat_probe.cpp"),
                        "-L", pkg, "-lasm_mi355x", "-Wl,-rpath," + pkg], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    cfg, _, _ = asm.workload("C5")
    hb = asm.generate_pairs(cfg, 99, 300)
    path = str(tmp_path / "pairs.seq")
    hb.write_seq_file(path)
    out = subprocess.run([exe, path, "3"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [ln.split() for ln in out.stdout.splitlines()]
    assert len(rows) == hb.n
    nw, ncig = oracle.nw_cigar(hb)
    leap = oracle.leap(hb, k=3)
    greedy, gcig = oracle.greedy(hb, k=3, mode=0, cigars=True)
    m, n = hb.lengths()
    for i, row in enumerate(rows):
        assert int(row[0]) == nw[i] and row[1].replace("-", "") == ncig[i], (i, row[:2], int(nw[i]), ncig[i])
        if max(m[i], n[i]) <= 256:   # LEAP beyond 256 bases is undefined in the reference
            assert int(row[2]) == leap[i], (i, row[2], int(leap[i]))
        assert int(row[3]) == greedy[i] and row[4].replace("-", "") == gcig[i], (i, row[3:], int(greedy[i]), gcig[i])
