"""CPU-side checks of the boundary and the host logic: the C-ABI library loads and exports every declared symbol,
fails loudly without a GPU, and the generator / batch containers behave.  No compute kernels run here."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))


def test_library_exports_every_declared_symbol(asm):
    lib = asm.load_library()
    declared = asm.declared_symbols()
    assert len(declared) >= 30
    exported = subprocess.check_output(["nm", "-D", "--defined-only", asm.LIB_PATH], text=True)
    names = set(re.findall(r" T (asm_[a-z0-9_]+)", exported))
    assert set(declared) <= names, sorted(set(declared) - names)
    assert set(declared) == set(lib._asm_symbols)  # the ctypes table binds exactly the header's functions
    assert b"gfx950" in lib.asm_version()


def test_struct_layouts_match_header(asm):
    assert ctypes.sizeof(asm.Params) == 48 and asm.Params.p_match.offset == 16 and asm.Params.alignment_type.offset == 40 and asm.Params.leap_mode.offset == 44
    assert ctypes.sizeof(asm.GenConfig) == 40 and asm.GenConfig.err.offset == 20
    p = asm.Params()
    asm.load_library().asm_default_params(ctypes.byref(p))
    assert (p.k, p.x, p.o, p.e) == (3, 1, 1, 1) and abs(p.p_match - 0.8) < 1e-12 and abs(p.p_indel - 0.4 / 3) < 1e-12
    assert p.alignment_type == asm.ALIGN_GLOBAL and p.leap_mode == asm.LEAP_GLOBAL


def test_no_gpu_means_loud_failure_not_fallback(asm):
    """Without a HIP device the product refuses to run (there is no CPU path)."""
    if asm.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(asm.AsmError) as ei:
        asm.Engine(0)
    assert ei.value.code == -2 and "no HIP device" in str(ei.value)


def test_product_never_touches_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "approximate-string-matching_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "libasm_oracle" not in text and "oracle_binding" not in text and "asm_oracle.h" not in text, f
    deps = subprocess.check_output(["ldd", os.path.join(pkg, "libasm_mi355x.so")], text=True)
    assert "oracle" not in deps and "amdhip64" in deps


def test_generator_is_deterministic_and_sliceable(asm):
    cfg, _, _ = asm.workload("C2")
    a = asm.generate_pairs(cfg, 0, 3000)
    b = asm.generate_pairs(cfg, 0, 3000)
    assert np.array_equal(a.reads, b.reads) and np.array_equal(a.refs, b.refs)
    part = asm.generate_pairs(cfg, 1000, 500)  # any slice of the stream stands alone (how shards are made)
    sub = a.slice(1000, 1500)
    assert np.array_equal(part.reads, sub.reads) and np.array_equal(part.refs, sub.refs)
    assert np.array_equal(part.read_off, sub.read_off) and np.array_equal(part.ref_off, sub.ref_off)
    other = asm.generate_pairs(asm.GenConfig.exact(99, 100, 0.10), 0, 100)
    assert not np.array_equal(other.reads, a.slice(0, 100).reads)


def test_generator_distribution_follows_dataset(asm, oracle):
    """benchmark_dataset.h: exactly ceil(L*err) edits, 96 % substitutions, rest del/ins 50/50."""
    cfg, _, _ = asm.workload("C2")
    hb = asm.generate_pairs(cfg, 0, 20000)
    m, n = hb.lengths()
    assert (m == 100).all() and set(np.unique(hb.reads)) == set(b"ACGT") and set(np.unique(hb.refs)) <= set(b"ACGT")
    ed = oracle.levenshtein(hb)
    assert ed.max() <= 10 and 6.8 < ed.mean() < 7.8            # 10 ops, 25 % of substitutions are no-ops
    assert 0.27 < (m != n).mean() < 0.35                        # ~31 % of pairs have n != m
    assert abs((n - m).mean()) < 0.02 and np.abs(n - m).max() <= 10
    base_freq = np.bincount(hb.reads, minlength=128)[[65, 67, 71, 84]] / hb.reads.size
    assert np.abs(base_freq - 0.25).max() < 0.005
    # float quirk of ceil(length * error_rate) in benchmark_dataset.h:154: 100 * 0.15f -> 16 edits
    hb15 = asm.generate_pairs(asm.GenConfig.exact(1, 100, 0.15, mismatch_rate=0.0), 0, 300)
    m15, n15 = hb15.lengths()
    assert ((n15 - m15) % 2 == 0).all() and np.abs(n15 - m15).max() <= 16 and np.abs(n15 - m15).max() > 6
    # C4: per-base rates of README.md:73-76
    cfg4, _, _ = asm.workload("C4")
    hb4 = asm.generate_pairs(cfg4, 0, 20000)
    ed4 = oracle.levenshtein(hb4)
    assert 2.3 < ed4.mean() < 2.8
    # C5: lengths uniform in [64, 300]
    cfg5, _, _ = asm.workload("C5")
    m5, _ = asm.generate_pairs(cfg5, 0, 20000).lengths()
    assert m5.min() == 64 and m5.max() == 300 and abs(m5.mean() - 182) < 2


def test_generator_rejects_bad_configs(asm):
    with pytest.raises(asm.AsmError):
        asm.generate_pairs(asm.GenConfig.exact(1, 100, 0.9), 0, 10)   # benchmark_dataset.h:194-198: err <= 0.7
    with pytest.raises(asm.AsmError):
        asm.generate_pairs(asm.GenConfig.exact(1, 100, 0.1, mismatch_rate=1.5), 0, 10)
    with pytest.raises(asm.AsmError):
        asm.generate_pairs(asm.GenConfig.exact(1, 600, 0.1), 0, 10)


def test_seq_file_round_trip(asm, tmp_path):
    """The harness's input format (benchmark_utils.h:325-352; written by benchmark_dataset.h:229,234)."""
    cfg, _, _ = asm.workload("C1")
    hb = asm.generate_pairs(cfg, 0, 50)
    path = str(tmp_path / "simulated.seq")
    hb.write_seq_file(path)
    lines = open(path).read().splitlines()
    assert len(lines) == 100 and lines[0][0] == ">" and lines[1][0] == "<"
    back = asm.HostBatch.read_seq_file(path)
    assert np.array_equal(back.reads, hb.reads) and np.array_equal(back.ref_off, hb.ref_off)
    assert asm.HostBatch.read_seq_file(path, max_pairs=7).n == 7


def test_shard_bounds_cover_batch(asm):
    for total, world in ((10, 3), (1_000_000, 8), (5, 8), (0, 2)):
        spans = [asm.shard_bounds(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_tail_state_advance_is_the_reference_buffer_model(asm, oracle):
    """asm_tail_state_advance (host function of the product, no GPU) against the oracle's model of the reference's two
    persistent buffers (hurdle_matrix.h:136-137,630-631; bit_convert.cpp:265-330): folding per-chunk summaries reproduces the
    buffer content after every chunk, for chunk sizes with every phase mod 10, empty chunks included."""
    from tests import oracle_binding as ob

    cfg, _, _ = asm.workload("C5")
    sizes = [1, 9, 10, 0, 11, 137, 2560, 3, 255, 64]
    state = np.zeros(256, np.uint8)
    first = 0
    oracle.set_initial_buffers(None)
    whole = asm.generate_pairs(cfg, 0, sum(sizes))
    for n in sizes:
        hb = asm.generate_pairs(cfg, first, n)
        state = asm.tail_state_advance(state, oracle.tail_summary(hb), n)
        first += n
        oracle.greedy_views(whole.slice(0, first), 0)          # the reference's chain over the file so far
        want = np.array([ob.base_code(c) for c in oracle.final_buffers()], np.uint8)
        assert np.array_equal(state, want), f"after {first} pairs"
    with pytest.raises(asm.AsmError):
        asm.tail_state_advance(state, np.full(256, 7, np.uint8), 5)


def test_the_reference_mains_compile_unmodified_against_the_compat_headers(asm):
    """The drop-in claim on the reference's own callers: GASMA/main.cpp (hurdle_matrix's string constructor, print(), run,
    get_CIGAR, get_cost, long_consecutive_matching_substring) and GASMA/benchmark/benchmark.cpp (Dataset, benchmark) are
    compiled IN PLACE, unmodified, with host/compat first on the quote include path, and linked against libasm_mi355x.so
    (`make -C oracle shim`).  Only where /root/reference exists; the executables travel to the GPU box, where
    tests/test_gpu_parity.py::test_the_reference_mains_run_on_the_library runs them."""
    import subprocess

    if not os.path.isdir("/root/reference/GASMA"):
        pytest.skip("the reference tree is not on this machine")
    if not os.path.exists(asm.LIB_PATH):
        pytest.skip("libasm_mi355x.so is not built")
    oracle_dir = os.path.join(ROOT, "oracle")
    for exe in ("gasma_main_on_shim", "benchmark_main_on_shim"):  # force the compile + link, whatever is there already
        path = os.path.join(oracle_dir, "_ref", exe)
        if os.path.exists(path):
            os.remove(path)
    r = subprocess.run(["make", "-s", "-C", oracle_dir, "shim"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for exe in ("gasma_main_on_shim", "benchmark_main_on_shim"):
        path = os.path.join(oracle_dir, "_ref", exe)
        assert os.path.exists(path), exe
        needed = subprocess.run(["readelf", "-d", path], capture_output=True, text=True).stdout
        assert "libasm_mi355x.so" in needed, "the executable must run on the product library"
