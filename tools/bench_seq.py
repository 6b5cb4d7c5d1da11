"""Where the sequential-mode step goes (development tool): PYTHONPATH=. python tools/bench_seq.py [n]"""
import sys, time
import numpy as np
import approximate_string_matching_amd as m

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
eng = m.Engine(0)
cfg, _, p = m.workload("C2")
b = eng.generate(cfg, 0, n)
state = np.zeros(256, np.uint8)
eng.resolve_tails(b, state)
eng.synchronize()
tm = eng.timer()
for name, fn in (("resolve_tails (clean g0 pack + chunk + carry + emit + pack)", lambda: eng.resolve_tails(b, state)),
                 ("pack only", lambda: eng.pack_async(b)),
                 ("tail_summary (g0 pack + chunk + carry + D2H)", lambda: eng.tail_summary(b))):
    best = 1e9
    for it in range(5):
        tm.start(); fn(); tm.stop(); best = min(best, tm.elapsed_ms())
    print("%-60s %.3f ms" % (name, best))
b.free(); eng.close()
