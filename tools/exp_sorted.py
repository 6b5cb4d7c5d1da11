"""Experiment: how much do LEAP/Greedy gain when the batch is pre-sorted by NW distance (divergence removed)?"""
import numpy as np, approximate_string_matching_amd as m
eng = m.Engine(0)
cfg, _, params = m.workload("C2")
n = 1_000_000
hb = m.generate_pairs(cfg, 0, n)
b0 = eng.upload(hb)
nw = eng.align(b0, m.NW, params)
leap = eng.align(b0, m.LEAP, params)
order = np.argsort(leap, kind="stable")
ro = hb.read_off.astype(np.int64); fo = hb.ref_off.astype(np.int64)
R = hb.reads.reshape(n, 100)[order].reshape(-1)
lens = np.diff(fo)[order]
newfo = np.zeros(n + 1, np.uint32); newfo[1:] = np.cumsum(lens)
idx = np.repeat(fo[:-1][order], lens) + (np.arange(lens.sum()) - np.repeat(newfo[:-1].astype(np.int64), lens))
F = hb.refs[idx]
hs = m.HostBatch(R, hb.read_off.copy(), F, newfo)
b1 = eng.upload(hs)
d = eng.malloc(4 * n); tm = eng.timer()
for name, b in (("input order", b0), ("sorted by LEAP generations", b1)):
    for a in (m.LEAP, m.GREEDY, m.NW):
        for it in range(3):
            tm.start(); eng.align_async(b, a, params, d); tm.stop(); ms = tm.elapsed_ms()
        print(name, m.ALIGNER_NAMES[a], "ms %.4f" % ms)
