"""Experiment (development tool): what would Greedy at k = 3 gain if the pairs of a wave needed the same number of steps?
The batch is re-ordered on the host by (a) the CIGAR op count of Greedy itself (a stand-in for its step count: the bound),
(b) cheap predictors, and timed with the persistent kernel and, with ASM_PERSIST=0 in the environment, the one-pair-per-thread
kernel.  PYTHONPATH=. [ASM_PERSIST=0] python tools/exp_greedy_sorted.py"""
import os
import numpy as np
import approximate_string_matching_amd as m

eng = m.Engine(0)
cfg, _, params = m.workload("C2")
n = 1_000_000
hb = m.generate_pairs(cfg, 0, n)
b0 = eng.upload(hb, m.GREEDY_CLEAN)
nw = eng.align(b0, m.NW, params)
cost, cig, nops = eng.greedy_with_cigar(b0, params, cap=64)
L = 100
R = hb.reads.reshape(n, L)
fo = hb.ref_off.astype(np.int64)
flen = np.diff(fo)
# main-diagonal mismatch vector (read[p] != ref[p], p < min(m, n)); references are ragged: gather the first L characters
idx = fo[:-1, None] + np.minimum(np.arange(L)[None, :], (flen - 1)[:, None])
F = hb.refs[idx]
mis = (R != F) | (np.arange(L)[None, :] >= flen[:, None])
ham = mis.sum(1)
# hurdles left after flipping isolated mismatches (a 1 with 0 on both sides), as v_flip_short_hurdles1 does on the lane vector
left = np.pad(mis[:, :-1], ((0, 0), (1, 0)))
right = np.pad(mis[:, 1:], ((0, 0), (0, 1)))
flipped = mis & (left | right)
fl = flipped.sum(1)
print("nops hist", np.bincount(np.minimum(nops, 12))[:13])
for name, key in (("nops", nops), ("hamming", ham), ("flipped", fl), ("nw", nw)):
    print("corr(nops, %s) = %.3f" % (name, np.corrcoef(nops, key)[0, 1]))


def reordered(order):
    lens = flen[order]
    newfo = np.zeros(n + 1, np.uint32)
    newfo[1:] = np.cumsum(lens)
    gi = np.repeat(fo[:-1][order], lens) + (np.arange(lens.sum()) - np.repeat(newfo[:-1].astype(np.int64), lens))
    return m.HostBatch(R[order].reshape(-1), hb.read_off.copy(), hb.refs[gi], newfo)


d = eng.malloc(4 * n)
tm = eng.timer()
rng = np.random.default_rng(1)
for name, key in (("input order", None), ("sorted by greedy ops (bound)", nops), ("sorted by min(flipped, 8)", np.minimum(fl, 8)),
                  ("sorted by min(hamming, 16)", np.minimum(ham, 16)), ("sorted by nw", nw),
                  ("blocks of 256 sorted by flipped", None)):
    if name.startswith("blocks"):
        nb = n // 256
        k2 = np.minimum(fl, 8)[:nb * 256].reshape(-1, 256)
        order = np.concatenate([(np.argsort(k2, axis=1, kind="stable") + (np.arange(nb) * 256)[:, None]).reshape(-1), np.arange(nb * 256, n)])
        b = eng.upload(reordered(order), m.GREEDY_CLEAN)
    elif key is None:
        b = b0
    else:
        b = eng.upload(reordered(np.argsort(key, kind="stable")), m.GREEDY_CLEAN)
    best = 1e9
    for it in range(4):
        tm.start(); eng.align_async(b, m.GREEDY, params, d); tm.stop(); best = min(best, tm.elapsed_ms())
    print("persist=%s %-34s greedy %.4f ms" % (os.environ.get("ASM_PERSIST", "1"), name, best), flush=True)
eng.close()
