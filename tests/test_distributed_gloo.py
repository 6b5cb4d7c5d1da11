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
