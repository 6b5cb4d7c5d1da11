"""Input side: the harness's file format (benchmark_utils.h:325-352) parsed on the device, and the streamed
file path — reader threads, pinned buffers in rotation, H2D overlapped with the aligners of the chunk before — against the oracle
run over the whole file."""
import os

import numpy as np
import pytest

from tests.util import random_ragged_batch

pytestmark = pytest.mark.gpu


def _text(hb, first_chars=(">", "<"), trailing_newline=True):
    lines = []
    for i in range(hb.n):
        a, b = hb.pair(i)
        lines.append(first_chars[0] + a)
        lines.append(first_chars[1] + b)
    return ("\n".join(lines) + ("\n" if trailing_newline and lines else "")).encode()


def _same(asm, got, want):
    assert got.n == want.n
    assert np.array_equal(got.read_off, want.read_off) and np.array_equal(got.ref_off, want.ref_off)
    assert np.array_equal(got.reads, want.reads) and np.array_equal(got.refs, want.refs)


def test_text_is_parsed_on_the_device(asm, engine, oracle):
    """Every line's first character is skipped blindly (benchmark_utils.h:337,343), whatever it is; empty strings, a missing
    final newline and a read without its reference line (-> empty reference) are handled; lengths 0..300 mixed."""
    hb = random_ragged_batch(asm, 71, 3000, 0, 300)
    _same(asm, engine.batch_from_text(_text(hb)).download(), hb)
    _same(asm, engine.batch_from_text(_text(hb, ("x", "#"), trailing_newline=False)).download(), hb)
    assert engine.batch_from_text(b"").n == 0
    odd = engine.batch_from_text(b">ACGT\n<ACGA\n>TTTT").download()
    assert [odd.pair(i) for i in range(odd.n)] == [("ACGT", "ACGA"), ("TTTT", "")]
    cfg, _, params = asm.workload("C2")
    hb = asm.generate_pairs(cfg, 5, 20000)
    batch = engine.batch_from_text(_text(hb), asm.GREEDY_SEQUENTIAL)
    assert np.array_equal(engine.align(batch, asm.GREEDY, params), oracle.greedy(hb, 3, mode=0))
    assert np.array_equal(engine.align(batch, asm.NW, params), oracle.nw(hb))


@pytest.mark.parametrize("wl,n,chunk", [("C5", 30000, 1 << 18), ("C2", 50000, 1 << 20), ("C2", 3000, 0)])
def test_streamed_file_equals_the_whole_file_run(asm, engine, oracle, tmp_path, wl, n, chunk):
    """`read_string_file` + `run` through the streaming path with chunks far smaller than the file (dozens of chunk boundaries,
    each cutting a line somewhere): all three aligners equal the oracle over the WHOLE file — Greedy in sequential mode, whose
    stale-tail chain has to run through every chunk boundary — and the device counters equal a host recount."""
    cfg, _, params = asm.workload(wl)
    hb = asm.generate_pairs(cfg, 11, n)
    path = str(tmp_path / "pairs.seq")
    hb.write_seq_file(path)
    got, st = engine.stream_seq_file(path, params, asm.GREEDY_SEQUENTIAL, chunk_bytes=chunk)
    assert st.pairs == n and st.bytes == os.path.getsize(path) and st.max_length == int(max(hb.lengths()[0].max(), hb.lengths()[1].max()))
    if chunk:
        assert st.chunks >= os.path.getsize(path) // chunk
    nw, leap, greedy = oracle.nw(hb), oracle.leap(hb, params.k), oracle.greedy(hb, params.k, mode=0)
    ok_leap = np.maximum(*hb.lengths()) <= 256
    assert np.array_equal(got[asm.NW], nw)
    assert np.array_equal(got[asm.LEAP][ok_leap], leap[ok_leap])
    assert np.array_equal(got[asm.GREEDY], greedy)
    assert list(st.counters) == [n, n, int((got[asm.LEAP] == nw).sum()), int((greedy == nw).sum())]
    # max_test_num (benchmark_utils.h:331) cutting inside a chunk, one aligner only, clean tails
    cut = n // 3 + 7
    part, st2 = engine.stream_seq_file(path, params, asm.GREEDY_CLEAN, aligners=(asm.GREEDY,), chunk_bytes=chunk, max_pairs=cut)
    assert st2.pairs == cut and np.array_equal(part[asm.GREEDY], oracle.greedy(hb.slice(0, cut), params.k, mode=1))
    # without NW in the mask (the C3 shape: LEAP + Greedy): the staging buffers of the two other aligners alone
    two, st4 = engine.stream_seq_file(path, params, asm.GREEDY_SEQUENTIAL, aligners=(asm.LEAP, asm.GREEDY), chunk_bytes=chunk)
    assert st4.pairs == n and np.array_equal(two[asm.GREEDY], greedy) and np.array_equal(two[asm.LEAP][ok_leap], leap[ok_leap])
    # an answers file (benchmark_utils.h:358-368) shorter than the input: the rest falls back to the NW penalty
    answers = nw[: n // 2].copy()
    answers[::5] += 1
    _, st3 = engine.stream_seq_file(path, params, asm.GREEDY_SEQUENTIAL, chunk_bytes=chunk, answers=answers)
    want = np.concatenate([answers, nw[n // 2:]])
    assert list(st3.counters) == [n, int((nw == want).sum()), int((got[asm.LEAP] == want).sum()), int((greedy == want).sum())]


def test_streaming_degenerate_files(asm, engine, oracle, tmp_path):
    """An empty file, a lone read line, a file without a final newline, chunk size below one pair's text."""
    p = asm.Params.default()
    empty = str(tmp_path / "empty.seq")
    open(empty, "w").close()
    got, st = engine.stream_seq_file(empty, p)
    assert st.pairs == 0 and st.chunks == 0 and all(v.size == 0 for v in got.values())
    lone = str(tmp_path / "lone.seq")
    with open(lone, "w") as fh:
        fh.write(">ACGTACGTAC")
    got, st = engine.stream_seq_file(lone, p, asm.GREEDY_CLEAN)
    assert st.pairs == 1 and got[asm.NW].tolist() == [10]          # against an empty reference: gap of ten
    cfg, _, _ = asm.workload("C1")
    hb = asm.generate_pairs(cfg, 0, 500)
    path = str(tmp_path / "nonl.seq")
    hb.write_seq_file(path)
    with open(path, "rb+") as fh:                                   # drop the final newline
        fh.seek(-1, 2)
        fh.truncate()
    got, st = engine.stream_seq_file(path, p, asm.GREEDY_SEQUENTIAL, chunk_bytes=4096)   # 4 KiB chunks: ~20 pairs each
    assert st.pairs == 500 and st.chunks >= 20
    assert np.array_equal(got[asm.GREEDY], oracle.greedy(hb, 3, mode=0)) and np.array_equal(got[asm.LEAP], oracle.leap(hb, 3))


def test_streaming_errors(asm, engine, tmp_path):
    with pytest.raises(asm.AsmError):
        engine.stream_seq_file(str(tmp_path / "missing.seq"), asm.Params.default())
    path = str(tmp_path / "long.seq")
    with open(path, "w") as fh:
        fh.write(">" + "A" * 600 + "\n<" + "A" * 600 + "\n")
    with pytest.raises(asm.AsmError):
        engine.stream_seq_file(path, asm.Params.default())
