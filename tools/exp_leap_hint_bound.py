"""Experiment (development tool): how good is each work hint for wide-band LEAP at C3?  PYTHONPATH=. python tools/exp_leap_hint_bound.py"""
import approximate_string_matching_amd as m
eng = m.Engine(0)
cfg, _, params = m.workload("C3")
n = 1_000_000
b = eng.generate(cfg, 0, n)
d_g, d_l, d_l2, d_nw = (eng.malloc(4 * n) for _ in range(4))
eng.align_async(b, m.GREEDY, params, d_g)
eng.align_async(b, m.LEAP, params, d_l)
eng.align_async(b, m.NW, params, d_nw)
tm = eng.timer()
for name, hint in (("no hint", None), ("greedy hint", d_g), ("nw hint", d_nw), ("own result as hint (bound)", d_l)):
    best = 1e9
    for it in range(4):
        tm.start(); eng.align_hinted_async(b, m.LEAP, params, hint, d_l2); tm.stop(); best = min(best, tm.elapsed_ms())
    print("%-30s %.3f ms" % (name, best))
import numpy as np
g, l, w = eng.to_host(d_g, n), eng.to_host(d_l, n), eng.to_host(d_nw, n)
print("corr(leap, greedy) %.3f corr(leap, nw) %.3f" % (np.corrcoef(l, g)[0, 1], np.corrcoef(l, w)[0, 1]))
del b
eng.close()
