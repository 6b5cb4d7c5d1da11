#!/usr/bin/env python3
"""bench.py — the hot path on N GPUs of one node: `python bench.py --gpus N --steps K --warmup W`.

A *step* is one pass of the reference's per-pair benchmark work (`benchmark::_run_benchmark`,
GASMA/benchmark/benchmark_utils.h:231-259) over one resident batch: pack (ASCII -> bit planes, the reference
converts inside its timed Greedy call), NW, LEAP, Greedy, and the three accuracy counters.  The batch is
BASELINE.json's configs[1] ("C2": 1e6 simulated 100 bp pairs, err 0.10, k=3, x=o=e=1) per GPU, generated on the
device from the seeded stream (rank r owns pairs [r*n, (r+1)*n)) — inputs are resident in HBM before the timed
region.  Pairs are independent, so ranks share nothing on the data path (weak scaling); one RCCL all-reduce of the
four int64 counters {total, nw_ok, leap_ok, greedy_ok} closes the timed region.

Launch: one process per GPU.  Under torchrun (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`)
the ranks come from the environment; a plain `python bench.py --gpus N` starts the N ranks itself (fresh child
processes, created before this process has touched a GPU) and relays rank 0's line.

Prints ONE JSON line on rank 0 (see the contract in the task statement) with `roofline` for the dominant kernel and
`cpu_baseline` (the oracle = CPU port of the reference, timed on this host's cores on a bounded sample).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md §Chip-level parameters)
PMC_FILE = os.path.join(ROOT, "profiles", "r04_pmc.json")  # workload C2; other workloads: r04_pmc_<workload>.json


def pmc_path(workload):
    return PMC_FILE if workload == "C2" else os.path.join(ROOT, "profiles", "r04_pmc_%s.json" % workload.lower())
# repack mode of the timed step (ASM_PACK_PIPELINE): "0" -> 1 = strictly in order; "1" -> 2 = pipelined pack (the pack of step
# s+1 beside the aligners of step s, started behind that step's NW so that it does not keep the persistent Greedy kernel's
# workgroups off the CUs); "2" -> 3 = overlapped steps (default; as 2, and no step waits for the previous step's Greedy: three
# chains of kernels through the K steps, two sets of output arrays).  Every step still does all of its work inside the timed
# region; C2 on one box: 0.257 / 0.238 / 0.229 ms per step.  The in-order figure is reported beside the headline.
PACK_MODE = {"0": 1, "1": 2, "2": 3}[os.environ.get("ASM_PACK_PIPELINE", "2")]
STEP_FORMS = {1: "in order: pack, then the aligners, then the counters, every step behind the previous one",
              2: "pipelined pack (repack=2): the pack of step s+1 fills a second set of planes beside the aligners of step s",
              3: "overlapped steps (repack=3): two sets of planes and of output arrays; the pack of step s+1 and the aligners of "
                 "consecutive steps overlap, each step's counters run behind its own aligners; joined before the final barrier"}
KERNEL_SOURCES = ("approximate-string-matching_amd/csrc", "approximate-string-matching_amd/Makefile")


def kernel_source_digest():
    """sha256 over the kernel sources: the PMC file records the digest of the code it measured, so numbers from an older
    kernel can never be quoted for the current one (round 1's stale `traffic`)."""
    h = hashlib.sha256()
    files = []
    for rel in KERNEL_SOURCES:
        path = os.path.join(ROOT, rel)
        if os.path.isdir(path):
            files += sorted(os.path.join(path, f) for f in os.listdir(path) if f.endswith((".h", ".hip")))
        else:
            files.append(path)
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def load_pmc(workload):
    """Counters collected by tools/pmc_collect.py (separate rocprofv3 --pmc passes over this very script).  Returns
    (entry-per-kernel dict, meta) or (None, reason)."""
    path = pmc_path(workload)
    try:
        with open(path) as fh:
            d = json.load(fh)
    except (OSError, ValueError) as exc:
        return None, f"no PMC file {os.path.relpath(path, ROOT)} ({exc.__class__.__name__})"
    if d.get("source_digest") != kernel_source_digest():
        return None, f"{os.path.relpath(path, ROOT)} was measured on other kernel sources (digest {d.get('source_digest')})"
    if d.get("workload") != workload:
        return None, f"{os.path.relpath(path, ROOT)} holds workload {d.get('workload')}"
    d["_path"] = os.path.relpath(path, ROOT)
    return d, None


def pmc_for(pmc, short, pairs):
    """The PMC entry of the kernel family `short` ("greedy", "nw", "leap", "pack"), scaled to this launch's pair count."""
    if pmc is None:
        return None
    k = pmc["kernels"].get(short)
    if not k:
        return None
    scale = pairs / pmc["pairs"]
    out = {"kernel_name": k["kernel_name"], "source": f"{pmc.get('_path', '?')}@{pmc.get('git_head', '?')}"}
    for key in ("traffic_bytes", "fetch_bytes", "write_bytes", "insts_valu", "active_inst_valu", "thread_cycles_valu"):
        if k.get(key) is not None:
            out[key] = k[key] * scale
    out["valu_mix"] = k.get("valu_mix")
    if k.get("members"):  # a mixed-length batch: one kernel per width class, counters summed over the family
        out["members"] = [m["kernel_name"] for m in k["members"]]
    return out


def valu_roofline(entry, pmc, launch_ms):
    """The roofline that actually bounds these kernels: VALU issue.  Wave-instructions per launch from
    SQ_INSTS_VALU; two ceilings, both printed: the instruction mix as measured by tools/ubench/valu_rates on this chip
    (cycles per wave-instruction per SIMD at the measured shader clock) and the architectural 2 cycles per wave64 VALU
    instruction on a SIMD-32 at 2.4 GHz (MI355X_MICROARCH.md, row v_fma_f32).  lane_util = SQ_THREAD_CYCLES_VALU /
    (64 * SQ_ACTIVE_INST_VALU): the share of the 64 lanes that did work in the issued instructions."""
    if not entry or "insts_valu" not in entry:
        return None
    insts = entry["insts_valu"]
    achieved = insts / (launch_ms * 1e-3)
    simds = pmc.get("simd_count", 1024)
    out = {"bound": "valu-issue", "insts_per_launch": insts, "achieved": achieved, "unit": "wave-instructions/s"}
    ub, mix = pmc.get("ubench", {}), entry.get("valu_mix") or {}
    if mix.get("cycles_per_inst_mix") and ub.get("sclk_hz"):
        peak = simds * ub["sclk_hz"] / mix["cycles_per_inst_mix"]
        out["peak_measured_mix"] = {"peak": peak, "frac": achieved / peak, "cycles_per_inst": mix["cycles_per_inst_mix"],
                                    "sclk_hz": ub["sclk_hz"], "source": ub.get("source"),
                                    "what": "this kernel's static instruction histogram priced with the measured issue cost of "
                                            "each opcode (8 waves per SIMD, first loop start to last loop end)"}
    peak2 = simds * 2.4e9 / 2.0
    out["peak_arch_2cyc"] = {"peak": peak2, "frac": achieved / peak2}
    if entry.get("thread_cycles_valu") and entry.get("active_inst_valu"):
        out["lane_util"] = entry["thread_cycles_valu"] / (64.0 * entry["active_inst_valu"])
    return out


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


def parse_args():
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
    ap.add_argument("--no-in-order", action="store_true", help="skip the extra in-order leg (ms_per_step_in_order)")
    ap.add_argument("--no-standalone", action="store_true",
                    help="skip the separately instrumented stand-alone kernel pass (profiling runs: every launch is a timed-region launch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sequential", action="store_true", help="skip the Greedy sequential-mode (reference as run) leg")
    ap.add_argument("--extra-pairs", type=int, default=10_000_000,
                    help="N > 1 only: TOTAL pairs of the two extra legs (BASELINE configs 4 and 5: C4 strong scaling, C5 bucketed)")
    ap.add_argument("--extra-steps", type=int, default=5)
    ap.add_argument("--no-extra", action="store_true", help="N > 1: skip the C4 / C5 legs")
    ap.add_argument("--rotate", type=int, default=0,
                    help="resident batches the timed steps rotate over (step s uses batch s mod R), so that no step finds its "
                         "input in the 256 MiB Infinity Cache where an earlier step left it; 0 = as many as it takes for the "
                         "ASCII of R batches to exceed 1 GiB (6 at C2), 1 = one batch re-read every step")
    return ap.parse_args()


LLC_BYTES = 256 << 20  # MI355X Infinity Cache (MI355X_MICROARCH.md): what a re-read working set must exceed to come from HBM


def default_rotation(ascii_bytes):
    """Batches to rotate over: enough that their ASCII alone is 4x the Infinity Cache (the packed planes and the penalty arrays
    come on top), at most 8; a batch that is already that large is not rotated."""
    return int(max(1, min(8, -(-4 * LLC_BYTES // max(int(ascii_bytes), 1)))))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) and wait for them.  This
    parent has not touched a GPU (no HIP call, no torch.cuda initialisation — counting devices does not initialise them), so
    starting children is safe; rank 0 prints the JSON line on the stdout it inherits."""
    rehearsal = os.environ.get("ASM_DIST_BACKEND", "nccl") != "nccl"  # gloo: several ranks may share the one visible GPU
    import torch

    have = torch.cuda.device_count()
    if have < 1 or (have < args.gpus and not rehearsal):
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible")
    env = dict(os.environ, WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(args.gpus):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=dict(env, RANK=str(r), LOCAL_RANK=str(r))))
    code = 0
    deadline = time.time() + 3000
    for p in procs:
        try:
            rc = p.wait(timeout=max(deadline - time.time(), 1))
        except subprocess.TimeoutExpired:
            p.kill()
            rc = 124
        code = code or rc
    if code:
        for p in procs:
            if p.poll() is None:
                p.kill()
    raise SystemExit(code)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)
    quiet = QuietStdout()

    import torch

    import approximate_string_matching_amd as asm

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (torchrun --nproc-per-node "
                         f"{args.gpus}, or plain `python bench.py --gpus {args.gpus}` which starts the ranks itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    backend = os.environ.get("ASM_DIST_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm; gloo only to rehearse N>1 on a one-GPU box
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("ASM_FORCE_DIST") == "1":  # ASM_FORCE_DIST=1: exercise the collective path with one rank
        import torch.distributed as dist

        if world == 1:  # ASM_FORCE_DIST=1 without a launcher: a one-rank group on a free local port
            os.environ.setdefault("RANK", "0"), os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"), os.environ.setdefault("MASTER_PORT", str(free_port()))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    coll_device = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")

    cfg, n_default, params = asm.workload(args.workload)
    n = args.pairs or min(n_default, 1_000_000)
    first = asm.weak_shard_first(rank, n)
    if args.total_pairs:
        lo, hi = asm.shard_bounds(args.total_pairs, world, rank)
        first, n = lo, hi - lo
    eng = asm.Engine(local_rank)
    # ONE stream for everything: the library's kernels, torch's ops on the counters and the all-reduce.  torch's default
    # stream is HIP's legacy stream (handle 0), which the library's own non-blocking stream would never wait for, so the
    # bench runs inside a real torch stream and hands that to the library.
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        eng.set_stream(stream.cuda_stream)
        out = run(args, asm, eng, torch, dist, stream, rank, world, cfg, params, first, n, coll_device)
    if rank == 0:
        quiet.emit(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run(args, asm, eng, torch, dist, stream, rank, world, cfg, params, first, n, coll_device):
    batch = eng.generate(cfg, first, n)  # this rank's shard of the seeded stream, straight into HBM
    # The reference reads every pair once (benchmark_utils.h:373-385); a bench that re-packs ONE resident batch K times reads a
    # 350 MB working set that partly survives in the Infinity Cache from step to step.  So the timed steps rotate over R
    # distinct resident batches of the same seeded stream (batch j of rank r = the shard rank r would own in job j): step s
    # reads batch s mod R, whose ASCII, planes and outputs were last touched R steps — more than 1 GiB of traffic — ago.
    rot = args.rotate or default_rotation(batch.ascii_bytes)
    batches = [batch]
    for j in range(1, rot):
        fj = first + j * args.total_pairs if args.total_pairs else asm.weak_shard_first(rank + j * world, n)
        batches.append(eng.generate(cfg, fj, n))
    aligners = [asm.NW, asm.LEAP, asm.GREEDY] if args.workload != "C3" else [asm.LEAP, asm.GREEDY]
    d_pen = {a: eng.malloc(4 * n) for a in aligners}
    counters = torch.zeros(4, dtype=torch.int64, device="cuda")  # total, nw_ok, leap_ok, greedy_ok
    d_cnt = counters.data_ptr()
    d_nw, d_leap, d_greedy = d_pen.get(asm.NW), d_pen.get(asm.LEAP), d_pen.get(asm.GREEDY)
    # overlapped calls (repack = 3) alternate between two sets of output arrays
    d_alt = {a: eng.malloc(4 * n) for a in aligners} if PACK_MODE == 3 else d_pen
    calls = [0]
    it = [0]  # steps issued since the last reset: which batch is next

    def step(timers=None, b=None, repack=True):
        if b is None:
            b = batches[it[0] % len(batches)]
            it[0] += 1
        if timers is None:
            # `_run_benchmark` for the whole batch: pack, aligners, counters — one C-ABI call, five launches
            if PACK_MODE == 3 and repack:
                o = d_alt if calls[0] & 1 else d_pen
                calls[0] += 1
                eng.run_benchmark_async(b, params, o.get(asm.NW), o.get(asm.LEAP), o.get(asm.GREEDY), d_cnt, repack=3)
            else:
                eng.run_benchmark_async(b, params, d_nw, d_leap, d_greedy, d_cnt, repack=(PACK_MODE if repack else 0))
            return
        seq = [("pack", lambda: eng.pack_async(b))]
        for a in aligners:
            # as asm_run_benchmark_async does: LEAP scheduled by the NW penalties, or by the Greedy penalties where NW is not run
            hint = (d_nw if d_nw is not None else d_greedy) if a == asm.LEAP else None
            seq.append((asm.ALIGNER_NAMES[a], lambda a=a, hint=hint: eng.align_hinted_async(b, a, params, hint, d_pen[a])))
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

    for _ in range(max(args.warmup, len(batches))):  # every batch is packed (and its second plane set allocated) once
        step()
    eng.pipeline_join_async()
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
    same_timers = {}
    if len(batches) > 1 and not args.no_standalone:  # the same pass over ONE batch: what the Infinity Cache gives a re-read
        for _ in range(min(args.steps, 20)):
            step(same_timers, b=batch)
        eng.synchronize()
    kernel_ms_same = {k: float(np.mean([t.elapsed_ms() for t in v])) for k, v in same_timers.items()}
    if dist is not None:  # warm the collective up outside the timed region (RCCL builds its rings on first use)
        dist.all_reduce(torch.zeros(4, dtype=torch.int64, device=coll_device))
    barrier()
    counters.zero_()
    # the dominant kernel's duration INSIDE the timed region: HIP events recorded by the library on the stream that kernel
    # is launched on (Greedy: the handle's side stream, beside NW -> LEAP); only this one kernel is bracketed, because an
    # event record keeps the next kernel of its stream from starting early (all four cost ~8 % of the step)
    eng.profile_enable(args.steps, 1 << names.index(dom))
    ev_ar = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    torch.cuda.synchronize()
    it[0] = 0  # timed step s reads batch s mod R
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.pipeline_join_async()  # overlapped calls end on the library's streams: order them before what follows on `stream`
    ar_wall = None
    if dist is not None:
        if coll_device.type == "cuda":
            ev_ar[0].record(stream)
            dist.all_reduce(counters)  # the only collective: 32 bytes over xGMI, ordered after the last step on `stream`
            ev_ar[1].record(stream)
        else:  # gloo rehearsal: through the host
            stream.synchronize()
            t_ar = time.perf_counter()
            host = counters.cpu()
            dist.all_reduce(host)
            counters.copy_(host)
            ar_wall = (time.perf_counter() - t_ar) * 1e3
    barrier()
    elapsed_local = time.perf_counter() - t0
    elapsed = elapsed_local
    per_rank_ms = [elapsed_local / args.steps * 1e3]
    allreduce_ms = None
    if dist is not None:
        allreduce_ms = ar_wall if ar_wall is not None else float(ev_ar[0].elapsed_time(ev_ar[1]))
        tt = torch.tensor([elapsed_local], dtype=torch.float64, device=coll_device)
        parts = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(parts, tt)
        per_rank_ms = [float(p.item()) / args.steps * 1e3 for p in parts]
        elapsed = max(float(p.item()) for p in parts)

    cnt = counters.cpu().numpy().copy()
    region = eng.profile_read(args.steps)
    eng.profile_enable(0, 0)
    q_dom = names.index(dom)
    region_ms = {dom: float(region[:, q_dom].mean())} if region.shape[0] and (region[:, q_dom] >= 0).all() else {dom: kernel_ms[dom]}

    # what the counters must be: one in-order pass per batch gives that batch's four counts c_j; the timed region used batch j
    # in uses_j of its steps, so counters == sum_j uses_j * c_j exactly (tests/test_gpu_bench.py checks c_j against the oracle)
    uses = [len(range(j, args.steps, len(batches))) for j in range(len(batches))]
    expected = torch.zeros(4, dtype=torch.int64, device="cuda")
    per_batch = []
    for j, bj in enumerate(batches):
        cj = torch.zeros(4, dtype=torch.int64, device="cuda")
        eng.run_benchmark_async(bj, params, d_nw, d_leap, d_greedy, cj.data_ptr(), repack=1)
        eng.synchronize()
        per_batch.append([int(v) for v in cj.cpu().numpy()])
        expected += uses[j] * cj
    if dist is not None:
        if coll_device.type == "cuda":
            dist.all_reduce(expected)
        else:
            host = expected.cpu()
            dist.all_reduce(host)
            expected.copy_(host)
    expected = [int(v) for v in expected.cpu().numpy()]

    # the same K steps strictly in order (repack = 1), for the record: what a caller without a second batch in flight gets
    in_order_ms = None
    scratch_cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    if PACK_MODE != 1 and not args.no_in_order:
        k_in = min(args.steps, 20)
        barrier()
        t1 = time.perf_counter()
        for s_ in range(k_in):
            eng.run_benchmark_async(batches[s_ % len(batches)], params, d_nw, d_leap, d_greedy, scratch_cnt.data_ptr(), repack=1)
        barrier()
        in_order_ms = (time.perf_counter() - t1) / k_in * 1e3
    # ... and the headline's form of the steps over ONE batch (what rounds 1-3 timed): the Infinity Cache's share of the number
    same_batch_ms = None
    if len(batches) > 1 and not args.no_in_order:
        k_in = min(args.steps, 20)
        for _ in range(2):
            step(b=batch)
        eng.pipeline_join_async()
        barrier()
        t1 = time.perf_counter()
        for _ in range(k_in):
            step(b=batch)
        eng.pipeline_join_async()
        barrier()
        same_batch_ms = (time.perf_counter() - t1) / k_in * 1e3

    sequential = None
    if not args.no_sequential and asm.GREEDY in aligners:
        sequential = sequential_leg(args, asm, eng, torch, dist, rank, world, cfg, params, first, n, d_pen, coll_device, barrier)

    # BASELINE.json configs 4 and 5 are multi-GPU configurations the driver's `--gpus N` run never names: after the C2
    # weak-scaling line's timed region, every rank also runs them as STRONG scaling over a fixed total (its contiguous slice)
    extra = None
    if world > 1 and not args.no_extra and args.workload == "C2":
        extra = {key: extra_leg(args, asm, eng, torch, dist, stream, rank, world, wl, coll_device, barrier)
                 for key, wl in (("c4_strong", "C4"), ("c5_bucketed", "C5"))}

    coverage = None
    if rank == 0 and asm.NW in aligners and (params.x, params.o, params.e) == (1, 1, 1):
        # the harness's fourth counter (benchmark_utils.h:256-258), once, outside the timed region
        cov = eng.coverage(batch, params, window=64)
        coverage = {"greedy_pct": 100.0 * cov["covered"] / max(n - cov["undetermined"], 1),
                    "undetermined_pairs": cov["undetermined"],
                    "note": "NW traceback tie-break is this library's (parasail's is unpinned); README.md:36 reports 94.213"}
    if rank != 0:
        return None
    # leave the penalties of the CLEAN-mode batch in d_pen for the parity leg below
    step(b=batch, repack=False)
    eng.synchronize()
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
    pmc, pmc_why = load_pmc(args.workload)
    entry = pmc_for(pmc, dom, n)
    out = {
        "metric": "alignments/sec (1e6-pair batch, 100bp, err=0.10) per GPU; NW penalty bit-exact %",
        "value": value,
        "unit": "read pairs/s through " + "+".join(asm.ALIGNER_NAMES[a].upper() if a == asm.NW else asm.ALIGNER_NAMES[a].capitalize()
                                                   for a in aligners)
                + " (whole job; consecutive steps " + ("overlap" if steps_form(asm, aligners, params) == 3 else
                                                        "pipeline their pack" if steps_form(asm, aligners, params) == 2 else "run in order")
                + ", inputs rotate over resident batches; value_in_order = the same steps strictly in order)",
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
            "workload": f"{args.workload}: {n} pairs per GPU, len {cfg.len_lo}-{cfg.len_hi}, {edit_model(asm, cfg)}, "
                        f"k={params.k}, x=o=e={params.x},{params.o},{params.e}, aligners "
                        + "+".join(asm.ALIGNER_NAMES[a] for a in aligners) + ", greedy tails=clean",
            "pairs_per_gpu": n,
            "sharding": "independent pairs, contiguous shard per rank, one 32-byte all-reduce of counters",
            "steps_run": STEP_FORMS[steps_form(asm, aligners, params)],
            "rotation": f"{len(batches)} resident batches of {n} pairs, step s reads batch s mod {len(batches)} "
                        f"({len(batches) * batch.ascii_bytes / 2**20:.0f} MiB of ASCII against a 256 MiB Infinity Cache)",
        },
        "ms_per_step_in_order": in_order_ms,
        "value_in_order": ((args.total_pairs if args.total_pairs else world * n) / (in_order_ms * 1e-3)) if in_order_ms else None,
        "ms_per_step_same_batch": same_batch_ms,
        "pack_GBps": {"rotating": pack_gbps(batch, n, kernel_ms.get("pack")),
                      "same_batch": pack_gbps(batch, n, kernel_ms_same.get("pack")),
                      "bytes": "ASCII read + 68 B of planes and lengths written per pair"},
        "kernel_ms_same_batch": kernel_ms_same or None,
        "ms_per_step_per_rank": per_rank_ms,
        "allreduce_ms": allreduce_ms,
        "kernel_ms_in_timed_region": region_ms,
        "kernel_ms": kernel_ms,
        "kernel_pairs_per_s": {k: n / (v * 1e-3) for k, v in kernel_ms.items()},
        "leap_greedy_pairs_per_s_per_gpu": n / ((kernel_ms.get("leap", 0) + kernel_ms.get("greedy", 0)) * 1e-3),
        "accuracy_pct": {asm.ALIGNER_NAMES[a]: 100.0 * float(cnt[1 + a]) / float(cnt[0]) for a in aligners}
        if asm.NW in aligners else None,
        "counters": {"total": int(cnt[0]), "nw_ok": int(cnt[1]), "leap_ok": int(cnt[2]), "greedy_ok": int(cnt[3]),
                     "expected_total": (args.total_pairs if args.total_pairs else world * n) * args.steps,
                     "expected": expected, "as_expected": [int(v) for v in cnt] == expected,
                     "per_batch_single_pass_rank0": per_batch, "steps_per_batch": uses},
        "coverage_pct": coverage,
        "sequential_mode": sequential,
        **(extra or {}),
        "roofline": {
            "bound": "hbm",
            "kernel": dom,
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": entry.get("traffic_bytes") if entry else None,
            "traffic_source": (entry or {}).get("source") or pmc_why,
            "traffic_kernel": (entry or {}).get("kernel_name"),
            "traffic_kernels_summed": (entry or {}).get("members"),
            "algorithmic_bytes_per_launch": alg_bytes,
            "avg_launch_ms": region_ms[dom],
            "note": "avg_launch_ms: HIP events on the launching stream inside the timed region, where this kernel shares the "
                    "GPU with the other chain; `standalone` = the same kernel with the GPU to itself; integer-VALU-bound path "
                    "(SURVEY.md F7): the binding roofline is in `valu` (stand-alone launch)",
            "standalone": {"avg_launch_ms": kernel_ms[dom], "achieved": alone, "frac": alone / HBM_PEAK_GBPS},
            "valu": valu_roofline(entry, pmc, kernel_ms[dom]) if pmc else None,
        },
    }
    if world == 1 and not args.no_cpu_baseline:
        out.update(cpu_baseline_and_parity(asm, eng, hb, batch, params, aligners, args.cpu_sample, d_pen))
    return out


def pack_gbps(batch, n, ms):
    """pack_kernel's bytes per launch (ASCII in, 4 x 16-byte plane granules + 4 bytes of lengths out at <= 128 characters) over
    its stand-alone duration."""
    if not ms:
        return None
    return (batch.ascii_bytes + 68.0 * n) / (ms * 1e-3) / 1e9


def steps_form(asm, aligners, params):
    """The form asm_run_benchmark_async really runs: a call with the Greedy-first shape (LEAP + Greedy without NW at a wide
    band, i.e. C3) asked to overlap runs as a pipelined-pack call."""
    if PACK_MODE == 3 and asm.NW not in aligners and params.k > 5:
        return 2
    return PACK_MODE


def extra_leg(args, asm, eng, torch, dist, stream, rank, world, workload, coll_device, barrier):
    """One more configuration of BASELINE.json in the same N-rank job: `--extra-pairs` pairs IN TOTAL (strong scaling), rank r
    owns the contiguous slice asm.shard_bounds gives it, generated on its own GPU; step = pack + NW + LEAP + Greedy + counters as
    in the main line; the four counters are summed over ranks by the same 32-byte all-reduce.  Mixed-length C5 goes through the
    bucketed-by-length launch inside every rank.  Returned on every rank (the caller prints rank 0's)."""
    cfg, _, params = asm.workload(workload)
    total = args.extra_pairs
    lo, hi = asm.shard_bounds(total, world, rank)
    n = hi - lo
    batch = eng.generate(cfg, lo, n)
    # rotating inputs as in the main line: batch j of this rank = its slice of the j-th block of `total` pairs of the stream
    rot = args.rotate or default_rotation(batch.ascii_bytes)
    batches = [batch] + [eng.generate(cfg, lo + j * total, n) for j in range(1, rot)]
    d = [eng.malloc(4 * max(n, 1)) for _ in range(6 if PACK_MODE == 3 else 3)]
    counters = torch.zeros(4, dtype=torch.int64, device="cuda")
    steps = args.extra_steps
    calls = [0]

    def step():
        o = d[3:] if (PACK_MODE == 3 and calls[0] & 1) else d[:3]
        b = batches[calls[0] % rot]
        calls[0] += 1
        eng.run_benchmark_async(b, params, o[0], o[1], o[2], counters.data_ptr(), repack=PACK_MODE)

    for _ in range(rot):
        step()
    eng.pipeline_join_async()
    barrier()
    counters.zero_()
    torch.cuda.synchronize()
    calls[0] = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    eng.pipeline_join_async()
    ar0 = time.perf_counter()
    if coll_device.type == "cuda":
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record(stream)
        dist.all_reduce(counters)
        ev[1].record(stream)
        torch.cuda.synchronize()
        allreduce_ms = ev[0].elapsed_time(ev[1])
    else:  # gloo rehearsal: through the host
        stream.synchronize()
        ar0 = time.perf_counter()
        host = counters.cpu()
        dist.all_reduce(host)
        counters.copy_(host)
        allreduce_ms = (time.perf_counter() - ar0) * 1e3
    barrier()
    mine = time.perf_counter() - t0
    tt = torch.tensor([mine], dtype=torch.float64, device=coll_device)
    parts = [torch.empty_like(tt) for _ in range(world)]
    dist.all_gather(parts, tt)
    elapsed = max(float(p.item()) for p in parts)
    cnt = counters.cpu().numpy()
    for ptr in d:
        eng.free(ptr)
    for b in batches:
        b.free()
    return {
        "workload": f"{workload}: {total} pairs in total over {world} ranks (strong), len {cfg.len_lo}-{cfg.len_hi}, "
                    f"{edit_model(asm, cfg)}, k={params.k}" + (", bucketed by length inside each rank" if cfg.len_hi > cfg.len_lo else "")
                    + f", rotating over {rot} resident batches",
        "scaling": "strong",
        "steps": steps,
        "ms_per_step": elapsed / steps * 1e3,
        "pairs_per_s": total * steps / elapsed,
        "ms_per_step_per_rank": [float(p.item()) / steps * 1e3 for p in parts],
        "allreduce_ms": allreduce_ms,
        "counters": {"total": int(cnt[0]), "nw_ok": int(cnt[1]), "leap_ok": int(cnt[2]), "greedy_ok": int(cnt[3]),
                     "expected_total": total * steps},
        "counters_as_expected": bool(int(cnt[0]) == total * steps and int(cnt[1]) == total * steps),
    }


def sequential_leg(args, asm, eng, torch, dist, rank, world, cfg, params, first, n, d_pen, coll_device, barrier):
    """The reference AS RUN (GASMA/hurdle_matrix.h:625-631): Greedy's conversion of pair t sees the stale bytes earlier pairs
    left in the two persistent buffers.  A step here = resolve the stale tails of the shard from the state the shards before
    it leave behind (three device passes over a clean packing, asm_tails.h) + pack with the tails OR-ed in + the aligners +
    counters.  With N ranks the shards are chained by ONE all-gather of 256-byte summaries (asm.chain_tail_state), so the
    N-GPU result equals the single-process run over the whole file."""
    seq = eng.generate(cfg, first, n)
    summary = eng.tail_summary(seq)
    state = asm.chain_tail_state(summary, n, dist, coll_device)
    d_nw, d_leap, d_greedy = d_pen.get(asm.NW), d_pen.get(asm.LEAP), d_pen.get(asm.GREEDY)
    steps = max(3, min(args.steps, 10))
    eng.resolve_tails(seq, state)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.resolve_tails(seq, state)  # three tail passes over a clean granule-0 packing + pack with the tails OR-ed in
        eng.run_benchmark_async(seq, params, d_nw, d_leap, d_greedy, None, repack=False)
    barrier()
    dt = time.perf_counter() - t0
    out = {"ms_per_step": dt / steps * 1e3, "steps": steps, "pairs_per_s_this_rank": n * steps / dt,
           "what": "tail resolve (clean granule-0 pack + 3 passes) + pack with tails + aligners, all enqueued, wall clock; "
                   "N ranks chain their shards through one all-gather of 256-byte summaries"}
    if rank == 0 and not args.no_cpu_baseline:
        from tests import oracle_binding

        orc = oracle_binding.load_oracle()
        s = min(args.cpu_sample, n)
        sub = seq.download().slice(0, s)
        got = eng.to_host(d_greedy, n)[:s]
        want = orc.greedy(sub, params.k, params.x, params.o, params.e, mode=0)
        clean = orc.greedy(sub, params.k, params.x, params.o, params.e, mode=1)
        out["greedy_bit_exact_pct_vs_oracle_sequential"] = 100.0 * float((got == want).mean())
        out["pairs_where_sequential_differs_from_clean_pct"] = 100.0 * float((want != clean).mean())
        out["sample"] = s
    seq.free()
    return out


def edit_model(asm, cfg):
    """How the workload's pairs were mutated, for `config.workload`."""
    if cfg.kind == asm.GEN_PER_BASE:
        return f"per-base sub {cfg.p_sub:.5f} ins {cfg.p_ins:.6f} del {cfg.p_del:.6f}"
    if cfg.kind == asm.GEN_UP_TO_ERRORS:
        return f"err <= {cfg.err:.2f}"
    return f"err {cfg.err:.2f}"


def usable_cpus():
    """CPUs this process may really use: its affinity mask capped by the cgroup's CPU quota (a GPU box hands a job a share of
    its host — 16 CPUs per GPU on this pool — while the mask still lists every core; threads beyond the quota only get
    throttled, which is what made round 2's "256 cores" leg look as if it did not scale)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


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
        cores = usable_cpus()
        orc.set_threads(cores)
        w0 = time.perf_counter()
        for a in aligners:
            if a == asm.NW:
                orc.nw(hb, params.x, params.o, params.e)
            elif a == asm.LEAP:
                orc.leap(hb, params.k, params.x, params.o, params.e)
            else:
                orc.greedy(hb, params.k, params.x, params.o, params.e, mode=1)
        all_cores = {"cores": cores, "value": hb.n / (time.perf_counter() - w0),
                     "sample": f"all {hb.n} pairs, wall clock, {cores} OpenMP threads = the CPUs this job may use "
                               f"(affinity mask {len(os.sched_getaffinity(0))}, cgroup quota applied)"}
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
