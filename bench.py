#!/usr/bin/env python3
"""bench.py — the hot path on N GPUs of one node: `python bench.py --gpus N --steps K --warmup W`.

A *step* is one pass of the reference's per-pair benchmark work (`benchmark::_run_benchmark`,
GASMA/benchmark/benchmark_utils.h:231-259) over one resident batch: pack (ASCII -> bit planes, the reference
converts inside its timed Greedy call), NW, LEAP, Greedy, and the three accuracy counters.  The batch is
BASELINE.json's configs[1] ("C2": 1e6 simulated 100 bp pairs, err 0.10, k=3, x=o=e=1) per GPU, generated on the
device from the seeded stream (rank r owns pairs [r*n, (r+1)*n)) — inputs are resident in HBM before the timed
region.  Pairs are independent, so ranks share nothing on the data path (weak scaling); one RCCL all-reduce of the
four int64 counters {total, nw_ok, leap_ok, greedy_ok} closes the timed region.

Prints ONE JSON line on rank 0 (see the contract in the task statement) with `roofline` for the dominant kernel and
`cpu_baseline` (the oracle = CPU port of the reference, timed on this host's cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md §Chip-level parameters)


def pmc_traffic(kernel, pairs):
    """HBM bytes per launch of `kernel` from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, collected offline with
    tools/pmc_traffic.sh in separate rocprofv3 --pmc passes and committed as profiles/r01_pmc_traffic.json), scaled
    to this launch's pair count.  None when no measurement is on file."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
        return d["kernels"][kernel]["traffic_bytes"] * pairs / d["pairs"]
    except (OSError, KeyError, ValueError):
        return None


def valu_roofline(kernel, pairs, launch_ms):
    """The roofline that actually bounds these kernels: VALU issue.  Instructions per launch come from the
    SQ_INSTS_VALU counter (profiles/r01_pmc_valu.json, collected with tools/pmc_sq.sh); the ceiling is what the SIMDs can
    issue for this instruction mix (measured ~4.3 cycles per wave-instruction, tools/ubench/valu_rates)."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_valu.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
        insts = d["kernels"][kernel]["insts_valu"] * pairs / d["pairs"]
        peak = d["simd_count"] * d["nominal_clock_hz"] / d["cycles_per_inst_mix"]
        achieved = insts / (launch_ms * 1e-3)
        return {"bound": "valu-issue", "insts_per_launch": insts, "achieved": achieved, "peak": peak,
                "unit": "wave-instructions/s", "frac": achieved / peak}
    except (OSError, KeyError, ValueError):
        return None


def algorithmic_bytes(m, n, aligners=1):
    """2-bit packed inputs read once + one int32 penalty per aligner, per pair."""
    return (np.ceil(2 * m / 8) + np.ceil(2 * n / 8) + 4 * aligners).sum()


class QuietStdout:
    """The contract is ONE JSON line on stdout.  RCCL, Gloo and the HIP runtime print banners to fd 1 from C code, so fd 1
    points at stderr while the benchmark runs and is put back (after flushing C stdio) only for the JSON line."""

    def __init__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def emit(self, line):
        import ctypes

        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)  # whatever C libraries still hold in their stdout buffer goes to stderr
        except OSError:
            pass
        os.dup2(self.saved, 1)
        os.write(1, (line + "\n").encode())
        os.dup2(2, 1)


def main():
    quiet = QuietStdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--pairs", type=int, default=0, help="pairs per GPU (default: the workload's batch, capped 1e6)")
    ap.add_argument("--total-pairs", type=int, default=0,
                    help="strong scaling: this many pairs in total, split into contiguous shards over the ranks "
                         "(e.g. --workload C4 --total-pairs 10000000); default is weak scaling with --pairs per GPU")
    ap.add_argument("--cpu-sample", type=int, default=300_000)
    ap.add_argument("--no-standalone", action="store_true",
                    help="skip the separately instrumented stand-alone kernel pass (profiling runs: every launch is a timed-region launch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    import approximate_string_matching_amd as asm

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if os.environ.get("ASM_DIST_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % torch.cuda.device_count()  # rehearsal: several ranks share the one visible GPU
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("ASM_FORCE_DIST") == "1":  # ASM_FORCE_DIST=1: exercise the RCCL path with one rank
        import torch.distributed as dist

        # backend "nccl" IS RCCL on ROCm; ASM_DIST_BACKEND=gloo only to rehearse the N>1 code path on a one-GPU box
        backend = os.environ.get("ASM_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    cfg, n_default, params = asm.workload(args.workload)
    n = args.pairs or min(n_default, 1_000_000)
    first = asm.weak_shard_first(rank, n)
    if args.total_pairs:
        lo, hi = asm.shard_bounds(args.total_pairs, world, rank)
        first, n = lo, hi - lo
    eng = asm.Engine(local_rank)
    # everything (kernels, counters, the all-reduce) is ordered on torch's current stream
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    batch = eng.generate(cfg, first, n)  # this rank's shard of the seeded stream, straight into HBM
    aligners = [asm.NW, asm.LEAP, asm.GREEDY] if args.workload != "C3" else [asm.LEAP, asm.GREEDY]
    d_pen = {a: eng.malloc(4 * n) for a in aligners}
    counters = torch.zeros(4, dtype=torch.int64, device="cuda")  # total, nw_ok, leap_ok, greedy_ok
    d_cnt = counters.data_ptr()

    d_nw, d_leap, d_greedy = d_pen.get(asm.NW), d_pen.get(asm.LEAP), d_pen.get(asm.GREEDY)

    def step(timers=None):
        if timers is None:
            # `_run_benchmark` for the whole batch: pack, aligners, counters — one C-ABI call, five launches
            eng.run_benchmark_async(batch, params, d_nw, d_leap, d_greedy, d_cnt, repack=True)
            return
        seq = [("pack", lambda: eng.pack_async(batch))]
        for a in aligners:
            hint = d_nw if a == asm.LEAP else None  # as asm_run_benchmark_async does: LEAP scheduled by the NW penalties
            seq.append((asm.ALIGNER_NAMES[a], lambda a=a, hint=hint: eng.align_hinted_async(batch, a, params, hint, d_pen[a])))
        for name, fn in seq:
            t = eng.timer()
            t.start()
            fn()
            t.stop()
            timers.setdefault(name, []).append(t)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # stand-alone durations: a separately instrumented pass with the kernels one after the other on one stream (what a
    # kernel costs when it has the GPU to itself; in the timed region Greedy shares it with NW and LEAP) — run before
    # the timed region, where it also tells which kernel is the dominant one
    names = ("pack", "nw", "leap", "greedy")
    timers = {}
    for _ in range(3 if args.no_standalone else min(args.steps, 20)):
        step(timers)
    eng.synchronize()
    kernel_ms = {k: float(np.mean([t.elapsed_ms() for t in v])) for k, v in timers.items()}
    dom = max(kernel_ms, key=kernel_ms.get)
    barrier()
    counters.zero_()
    # the dominant kernel's duration INSIDE the timed region: HIP events recorded by the library on the stream that kernel
    # is launched on (Greedy: the handle's side stream, beside NW -> LEAP); only this one kernel is bracketed, because an
    # event record keeps the next kernel of its stream from starting early (all four cost ~8 % of the step)
    eng.profile_enable(args.steps, 1 << names.index(dom))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if dist is not None:
        dist.all_reduce(counters)  # the only collective: 32 bytes over xGMI
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    cnt = counters.cpu().numpy().copy()
    region = eng.profile_read(args.steps)
    eng.profile_enable(0, 0)
    q_dom = names.index(dom)
    region_ms = {dom: float(region[:, q_dom].mean())} if region.shape[0] and (region[:, q_dom] >= 0).all() else {dom: kernel_ms[dom]}

    coverage = None
    if rank == 0 and asm.NW in aligners and (params.x, params.o, params.e) == (1, 1, 1):
        # the harness's fourth counter (benchmark_utils.h:256-258), once, outside the timed region
        cov = eng.coverage(batch, params, window=64)
        coverage = {"greedy_pct": 100.0 * cov["covered"] / max(n - cov["undetermined"], 1),
                    "undetermined_pairs": cov["undetermined"],
                    "note": "NW traceback tie-break is this library's (parasail's is unpinned); README.md:36 reports 94.213"}
    if rank == 0:
        total_pairs = (args.total_pairs if args.total_pairs else world * n) * args.steps
        value = total_pairs / elapsed
        hb = batch.download()
        m_len, n_len = hb.lengths()
        if dom == "pack":
            alg_bytes = float((m_len + n_len).sum() + 68 * n)  # ASCII in, planes + lengths out
        else:
            alg_bytes = float(algorithmic_bytes(m_len, n_len, 1))
        achieved = alg_bytes / (region_ms[dom] * 1e-3) / 1e9
        alone = alg_bytes / (kernel_ms[dom] * 1e-3) / 1e9
        out = {
            "metric": "alignments/sec (1e6-pair batch, 100bp, err=0.10) per GPU; NW penalty bit-exact %",
            "value": value,
            "unit": "read pairs/s through NW+LEAP+Greedy (whole job)",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.total_pairs else "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {n} pairs per GPU, len {cfg.len_lo}-{cfg.len_hi}, err {cfg.err:.2f}, "
                            f"k={params.k}, x=o=e={params.x},{params.o},{params.e}, aligners "
                            + "+".join(asm.ALIGNER_NAMES[a] for a in aligners) + ", greedy tails=clean",
                "pairs_per_gpu": n,
                "sharding": "independent pairs, contiguous shard per rank, one 32-byte all-reduce of counters",
            },
            "kernel_ms_in_timed_region": region_ms,
            "kernel_ms": kernel_ms,
            "kernel_pairs_per_s": {k: n / (v * 1e-3) for k, v in kernel_ms.items()},
            "leap_greedy_pairs_per_s_per_gpu": n / ((kernel_ms.get("leap", 0) + kernel_ms.get("greedy", 0)) * 1e-3),
            "accuracy_pct": {asm.ALIGNER_NAMES[a]: 100.0 * float(cnt[1 + a]) / float(cnt[0]) for a in aligners}
            if asm.NW in aligners else None,
            "coverage_pct": coverage,
            "roofline": {
                "bound": "hbm",
                "kernel": dom,
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": pmc_traffic(dom, n),
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_ms": region_ms[dom],
                "note": "avg_launch_ms: HIP events on the launching stream inside the timed region, where this kernel shares the "
                        "GPU with the other chain; `standalone` = the same kernel with the GPU to itself; integer-VALU-bound path "
                        "(SURVEY.md F7): the binding roofline is in `valu` (stand-alone launch)",
                "standalone": {"avg_launch_ms": kernel_ms[dom], "achieved": alone, "frac": alone / HBM_PEAK_GBPS},
                "valu": valu_roofline(dom, n, kernel_ms[dom]),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out.update(cpu_baseline_and_parity(asm, eng, hb, batch, params, aligners, args.cpu_sample, d_pen))
        quiet.emit(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline_and_parity(asm, eng, hb, batch, params, aligners, sample, d_pen):
    """Rank 0, N=1 only: time the oracle (CPU port of the reference algorithms, bit-identical to the compiled
    reference — tests/test_oracle_vs_reference.py) on a bounded sample, single thread like the reference's own
    harness, and report the GPU-vs-oracle bit-exact percentage on that sample."""
    from tests import oracle_binding

    orc = oracle_binding.load_oracle()
    orc.set_threads(1)
    s = min(sample, hb.n)
    sub = hb.slice(0, s)
    t0 = time.process_time()
    want = {}
    per = {}
    for a in aligners:
        t1 = time.process_time()
        if a == asm.NW:
            want[a] = orc.nw(sub, params.x, params.o, params.e)
        elif a == asm.LEAP:
            want[a] = orc.leap(sub, params.k, params.x, params.o, params.e)
        else:
            want[a] = orc.greedy(sub, params.k, params.x, params.o, params.e, mode=1)
        per[asm.ALIGNER_NAMES[a]] = s / max(time.process_time() - t1, 1e-9)
    cpu_s = time.process_time() - t0
    # the same port on every host core (OpenMP over pairs), whole batch, wall clock
    all_cores = None
    try:
        cores = len(os.sched_getaffinity(0))
        orc.set_threads(cores)
        w0 = time.perf_counter()
        for a in aligners:
            if a == asm.NW:
                orc.nw(hb, params.x, params.o, params.e)
            elif a == asm.LEAP:
                orc.leap(hb, params.k, params.x, params.o, params.e)
            else:
                orc.greedy(hb, params.k, params.x, params.o, params.e, mode=1)
        all_cores = {"cores": cores, "value": hb.n / (time.perf_counter() - w0), "sample": f"all {hb.n} pairs, wall clock"}
        orc.set_threads(1)
    except Exception as exc:
        all_cores = {"error": repr(exc)}
    # where the real reference travelled with the repo (oracle/_ref, built in the authoring container from
    # /root/reference; NW's parasail is absent there), time ITS LEAP and Greedy on the same sample, same single thread
    ref_part = None
    try:
        if oracle_binding.have_reference():
            ref = oracle_binding.load_reference()
            t1 = time.process_time()
            r_leap = ref.leap(sub, params.k, params.x, params.o, params.e, full=True)  # incl. backtrack + get_CIGAR
            t2 = time.process_time()
            r_greedy = ref.greedy(sub, params.k, params.x, params.o, params.e, mode=1)
            t3 = time.process_time()
            m_s, n_s = sub.lengths()
            defined = np.abs(np.minimum(n_s, 128) - np.minimum(m_s, 128)) <= params.k  # else undefined in the reference
            ref_part = {"leap_pairs_per_s": s / max(t2 - t1, 1e-9), "greedy_pairs_per_s": s / max(t3 - t2, 1e-9),
                        "what": "reference sources compiled in place (oracle/_ref), calls as benchmark_utils.h:156-201",
                        "gpu_equals_reference_pct": {
                            "leap": 100.0 * float((eng.to_host(d_pen[asm.LEAP], batch.n)[:s] == r_leap).mean())
                            if asm.LEAP in d_pen else None,
                            "greedy": 100.0 * float((eng.to_host(d_pen[asm.GREEDY], batch.n)[:s] == r_greedy)[defined].mean())
                            if asm.GREEDY in d_pen else None,
                            "greedy_pairs_undefined_in_reference": int((~defined).sum())}}
    except Exception as exc:  # a checker that cannot load is not a bench failure
        ref_part = {"error": repr(exc)}
    exact = {}
    for a in aligners:
        got = eng.to_host(d_pen[a], batch.n)[:s]
        exact[asm.ALIGNER_NAMES[a]] = 100.0 * float((got == want[a]).mean())
    return {
        "cpu_baseline": {
            "value": s / cpu_s,
            "unit": "read pairs/s through " + "+".join(asm.ALIGNER_NAMES[a] for a in aligners),
            "cores": 1,
            "kind": "port",
            "sample": f"first {s} pairs of the same seeded batch, oracle/libasm_oracle.so, 1 thread, "
                      f"{cpu_s:.1f} s CPU; per aligner pairs/s: " + ", ".join(f"{k} {v:.3g}" for k, v in per.items()),
            "host_cpus": os.cpu_count(),
            "all_cores": all_cores,
            "reference_parts": ref_part,
        },
        "bit_exact_pct_vs_oracle": dict(exact, sample=s),
    }


if __name__ == "__main__":
    main()
