import time, numpy as np
import approximate_string_matching_amd as m
eng = m.Engine(0)
for wl in ("C2", "C5"):
    cfg, _, p = m.workload(wl)
    n = 10_000_000
    t = time.time(); b = eng.generate(cfg, 0, n, m.GREEDY_SEQUENTIAL); eng.synchronize(); t1 = time.time() - t
    t = time.time(); c = eng.generate(cfg, 0, n, m.GREEDY_CLEAN); eng.synchronize(); t2 = time.time() - t
    gs = eng.align(b, m.GREEDY, p); gc = eng.align(c, m.GREEDY, p)
    print(wl, "generate+pack sequential %.3fs clean %.3fs; pairs whose cost differs between modes: %d of %d" % (t1, t2, int((gs != gc).sum()), n))
    b.free(); c.free()
