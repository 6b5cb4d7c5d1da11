"""The N>1 path on CPU: two processes over gloo.  Pairs shard with no data-path collective; the only exchange is
the all-reduce of the accuracy counters (the same helper bench.py calls with the nccl/RCCL backend).  The GPU kernels
are not involved here, so per-shard penalties come from the oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import approximate_string_matching_amd as asm
    from tests import oracle_binding

    orc = oracle_binding.load_oracle()
    cfg, _, params = asm.workload("C1")
    lo, hi = asm.shard_bounds(total, world, rank)
    hb = asm.generate_pairs(cfg, lo, hi - lo)          # this rank's shard of the seeded stream
    nw, leap, greedy = orc.nw(hb), orc.leap(hb, params.k), orc.greedy(hb, params.k, mode=1)
    counters = torch.tensor([hb.n, int((nw == nw).sum()), int((leap == nw).sum()), int((greedy == nw).sum())],
                            dtype=torch.int64)
    asm.allreduce_counters(counters, dist)
    gathered = asm.gather_penalties(torch.from_numpy(greedy.copy()), dist, dst=0)
    if rank == 0:
        q.put((counters.tolist(), [g.numpy().tolist() for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_counter_allreduce(asm, oracle):
    total, world, port = 3001, 2, 29500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    counters, gathered = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cfg, _, params = asm.workload("C1")
    hb = asm.generate_pairs(cfg, 0, total)             # the unsharded batch
    nw, leap, greedy = oracle.nw(hb), oracle.leap(hb, params.k), oracle.greedy(hb, params.k, mode=1)
    assert counters == [total, total, int((leap == nw).sum()), int((greedy == nw).sum())]
    assert np.array_equal(np.concatenate([np.array(g, np.int32) for g in gathered]), greedy)


def _chain_file(asm, total):
    """A file whose second half is much shorter than its first: the short pairs never refresh the buffer slots the long pairs
    wrote, so what a late shard sees beyond its strings comes from pairs of an EARLIER shard."""
    long_cfg = asm.GenConfig.exact(21, 128, 0.10)
    short_cfg = asm.GenConfig.exact(22, 40, 0.10, length_hi=108)
    a, b = asm.generate_pairs(long_cfg, 0, total // 2), asm.generate_pairs(short_cfg, 0, total - total // 2)
    return asm.HostBatch(np.concatenate([a.reads, b.reads]), np.concatenate([a.read_off, b.read_off[1:] + a.read_off[-1]]),
                         np.concatenate([a.refs, b.refs]), np.concatenate([a.ref_off, b.ref_off[1:] + a.ref_off[-1]]))


def _chain_worker(rank, world, port, total, q):
    """Greedy's sequential mode over a sharded file: every rank summarises its shard, ONE all-gather carries the 256-byte
    summaries and shard sizes, every rank folds the shards before its own (the product's host function
    asm_tail_state_advance) and starts its chain from that state.  The device kernels are not involved on CPU, so the
    shard summary and the per-shard run come from the oracle's buffer model."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import approximate_string_matching_amd as asm
    from tests import oracle_binding

    orc = oracle_binding.load_oracle()
    lo, hi = asm.shard_bounds(total, world, rank)
    hb = _chain_file(asm, total).slice(lo, hi)
    state = asm.chain_tail_state(orc.tail_summary(hb), hb.n, dist)
    orc.set_initial_buffers(oracle_binding.codes_to_buffers(state))
    greedy = orc.greedy(hb, 3, mode=0)
    views = orc.greedy_views(hb, 0).reshape(-1).astype(np.int32)    # the 2 x 128 bytes every pair's conversion sees
    orc.set_initial_buffers(None)
    gathered = asm.gather_penalties(torch.from_numpy(greedy.copy()), dist, dst=0)
    gathered_views = asm.gather_penalties(torch.from_numpy(views.copy()), dist, dst=0)
    if rank == 0:
        q.put(([g.numpy().tolist() for g in gathered], [g.numpy().astype(np.uint8).tobytes() for g in gathered_views]))
    dist.barrier()
    dist.destroy_process_group()


def test_sequential_mode_chains_across_shards(asm, oracle):
    """world 2 and 3 (uneven shards, sizes not multiples of 10): the sharded sequential run equals the reference-as-run
    over the whole file, i.e. oracle.greedy(mode=0) on the unsharded batch."""
    for world, total in ((2, 2003), (3, 1999)):
        port = 31500 + (os.getpid() + world) % 2000
        ctx = mp.get_context("spawn")
        q = ctx.SimpleQueue()
        procs = [ctx.Process(target=_chain_worker, args=(r, world, port, total, q)) for r in range(world)]
        for p in procs:
            p.start()
        gathered, gathered_views = q.get()
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        hb = _chain_file(asm, total)
        whole = oracle.greedy(hb, 3, mode=0)
        sharded = np.concatenate([np.array(g, np.int32) for g in gathered])
        assert np.array_equal(sharded, whole)
        # stronger than the costs (few pairs are sensitive to their tails): every byte every conversion sees.  The chained
        # state holds 2-bit codes, so compare codes ('A', NUL and anything else are all code 00).
        code = np.zeros(256, np.uint8)
        code[ord("C")], code[ord("G")], code[ord("T")] = 1, 2, 3
        whole_views = oracle.greedy_views(hb, 0).reshape(-1)
        sharded_views = np.frombuffer(b"".join(gathered_views), np.uint8)
        assert np.array_equal(code[sharded_views], code[whole_views])
        # and the chain matters: shards that each start from empty buffers see other bytes than the whole-file run
        alone = np.concatenate([oracle.greedy_views(hb.slice(*asm.shard_bounds(total, world, r)), 0).reshape(-1)
                                for r in range(world)])
        assert (code[alone] != code[whole_views]).sum() > 100
