"""bench.py's own launch paths on the one-GPU box: the self-started ranks (two gloo ranks sharing the card: the N>1 code path,
not a scaling measurement), the RCCL collective with a single rank, and the refusal to run N ranks on fewer GPUs.  The
counters of the timed region are checked against the oracle: with everything on ONE stream they are exact, not racy."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, *argv, expect_ok=True):
    env = dict(os.environ, **extra_env)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=900)
    if not expect_ok:
        return r
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]          # the contract: ONE JSON line on stdout
    return json.loads(lines[0])


def _expected_counts(asm, oracle, first, n, wl="C2"):
    cfg, _, params = asm.workload(wl)
    hb = asm.generate_pairs(cfg, first, n)
    nw, leap, greedy = oracle.nw(hb), oracle.leap(hb, params.k), oracle.greedy(hb, params.k, mode=1)
    return np.array([n, n, int((leap == nw).sum()), int((greedy == nw).sum())])


def test_self_started_ranks_share_one_stream_with_the_collective(asm, oracle):
    n, steps, rot = 20000, 4, 3
    extra_total, extra_steps = 30001, 2   # an odd total: the strong split gives the ranks different shard sizes
    out = _bench({"ASM_DIST_BACKEND": "gloo"}, "--gpus", "2", "--pairs", str(n), "--steps", str(steps), "--warmup", "1", "--rotate", str(rot),
                 "--cpu-sample", "20000", "--no-cpu-baseline", "--extra-pairs", str(extra_total), "--extra-steps", str(extra_steps))
    assert out["n_gpus"] == 2 and len(out["ms_per_step_per_rank"]) == 2 and out["allreduce_ms"] is not None
    # rotating inputs: step s of rank r reads batch s mod R, batch j of rank r = shard r + j * world of the seeded stream, so the
    # counters are the oracle's counts of shards 0..5, shards 0 and 1 (used by steps 0 and 3) twice
    shard = [_expected_counts(asm, oracle, q * n, n) for q in range(2 * rot)]
    want = sum(shard[r + 2 * (s % rot)] for r in range(2) for s in range(steps))
    c = out["counters"]
    assert [c["total"], c["nw_ok"], c["leap_ok"], c["greedy_ok"]] == want.tolist()
    assert c["total"] == c["expected_total"] and c["as_expected"] and c["steps_per_batch"] == [2, 1, 1]
    assert out["config"]["rotation"].startswith("3 resident batches")
    assert out["sequential_mode"]["ms_per_step"] > 0
    # BASELINE configs 4 and 5 ride along in the same N-rank job: C4 strong scaling, C5 bucketed by length (bench.extra_leg)
    for key, wl in (("c4_strong", "C4"), ("c5_bucketed", "C5")):
        leg = out[key]
        assert leg["scaling"] == "strong" and leg["steps"] == extra_steps and len(leg["ms_per_step_per_rank"]) == 2
        assert leg["ms_per_step"] > 0 and leg["pairs_per_s"] > 0 and leg["allreduce_ms"] is not None
        assert leg["counters_as_expected"] and leg["counters"]["total"] == extra_total * extra_steps
        cfg, _, params = asm.workload(wl)
        hb = asm.generate_pairs(cfg, 0, extra_total * extra_steps)   # step s reads block s of the stream, the shards are contiguous slices of it
        nw, leap, greedy = oracle.nw(hb), oracle.leap(hb, params.k), oracle.greedy(hb, params.k, mode=1)
        assert leg["counters"]["greedy_ok"] == int((greedy == nw).sum())
        ok = np.maximum(*hb.lengths()) <= 256   # LEAP beyond 256 bases is undefined in the reference
        if ok.all():
            assert leg["counters"]["leap_ok"] == int((leap == nw).sum())


def test_single_rank_through_rccl_and_sequential_leg(asm, oracle):
    n, steps = 30000, 4
    out = _bench({"ASM_FORCE_DIST": "1"}, "--pairs", str(n), "--steps", str(steps), "--warmup", "1", "--cpu-sample", "30000", "--rotate", "2")
    want = 2 * _expected_counts(asm, oracle, 0, n) + 2 * _expected_counts(asm, oracle, n, n)   # two batches, two steps each
    c = out["counters"]
    assert [c["total"], c["nw_ok"], c["leap_ok"], c["greedy_ok"]] == want.tolist()
    assert c["as_expected"] and c["per_batch_single_pass_rank0"][1] == _expected_counts(asm, oracle, n, n).tolist()
    assert out["ms_per_step_same_batch"] > 0 and out["pack_GBps"]["rotating"] > 0 and out["pack_GBps"]["same_batch"] > 0
    assert out["bit_exact_pct_vs_oracle"]["greedy"] == 100.0 and out["bit_exact_pct_vs_oracle"]["nw"] == 100.0
    seq = out["sequential_mode"]
    assert seq["greedy_bit_exact_pct_vs_oracle_sequential"] == 100.0
    assert seq["pairs_where_sequential_differs_from_clean_pct"] > 0       # the two modes really differ on this input
    assert out["roofline"]["frac"] > 0 and out["cpu_baseline"]["value"] > 0


def test_more_ranks_than_gpus_is_refused(asm):
    have = asm.device_count()
    r = _bench({}, "--gpus", str(have + 1), "--pairs", "1000", "--steps", "1", expect_ok=False)
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)
    # a launcher that started a different number of ranks than --gpus says
    r = _bench({}, "--gpus", "1", "--pairs", "1000", "--steps", "1", expect_ok=False)
    assert r.returncode == 0
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--pairs", "1000", "--steps", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_cxx_harness_multi_gpu_mode_and_streaming(asm, oracle, tmp_path):
    """asm-bench, the C++ host over the same C ABI: `--gpus N` (one thread and one handle per GPU, RCCL all-reduce of the four
    counters on the stream the kernels run on; N = 1 on this box, N + 1 refused) and `--stream` (the file through
    asm_stream_seq_file).  Percentages equal the oracle's counts exactly."""
    import re

    exe = os.path.join(ROOT, "approximate-string-matching_amd", "asm-bench")
    n, steps = 20000, 3
    r = subprocess.run([exe, "--gpus", "1", "--n", str(n), "--steps", str(steps), "--mode", "clean"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    want = _expected_counts(asm, oracle, 0, n)
    assert f"Total number of alignments: {n * steps} " in r.stdout
    got = [float(v) for v in re.findall(r"\| (\d+\.\d+) %", r.stdout)]
    assert got == [100.0, round(100.0 * want[2] / n, 3), round(100.0 * want[3] / n, 3)], r.stdout
    r = subprocess.run([exe, "--gpus", str(asm.device_count() + 1), "--n", "1000"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr
    # --stream: generate the reference-shaped file, then stream it in 1 MiB chunks (sequential mode = the reference as run)
    cfg, _, params = asm.workload("C2")
    hb = asm.generate_pairs(cfg, 0, n)
    path = str(tmp_path / "pairs.seq")
    hb.write_seq_file(path)
    r = subprocess.run([exe, "--file", path, "--n", str(n), "--stream", "--chunk-mb", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    nw, leap, greedy = oracle.nw(hb), oracle.leap(hb, 3), oracle.greedy(hb, 3, mode=0)
    got = [float(v) for v in re.findall(r"\| (\d+\.\d+) %", r.stdout)]
    assert got[:3] == [100.0, round(100.0 * float((leap == nw).mean()), 3), round(100.0 * float((greedy == nw).mean()), 3)], r.stdout
    assert "[Streamed] %d pairs" % n in r.stdout


def test_cxx_harness_sequential_mode_and_a_failing_rank(asm, oracle):
    """asm-bench --gpus N in its default mode (sequential: the reference as run — the shards are chained through their 256-byte
    tail summaries before the timed loop, as bench.py's ranks do) prints the oracle's sequential-mode Greedy percentage; and a
    rank that fails before the collective makes the program exit non-zero instead of leaving the other ranks blocked in it."""
    import re

    exe = os.path.join(ROOT, "approximate-string-matching_amd", "asm-bench")
    n, steps = 20000, 2
    r = subprocess.run([exe, "--gpus", "1", "--n", str(n), "--steps", str(steps)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    cfg, _, params = asm.workload("C2")
    hb = asm.generate_pairs(cfg, 0, n)
    nw, leap, greedy = oracle.nw(hb), oracle.leap(hb, params.k), oracle.greedy(hb, params.k, mode=0)
    got = [float(v) for v in re.findall(r"\| (\d+\.\d+) %", r.stdout)]
    assert got == [100.0, round(100.0 * float((leap == nw).mean()), 3), round(100.0 * float((greedy == nw).mean()), 3)], r.stdout
    r = subprocess.run([exe, "--gpus", "1", "--n", "5000", "--steps", "1"], env=dict(os.environ, ASM_BENCH_FAIL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "rank 0 failed" in r.stderr
