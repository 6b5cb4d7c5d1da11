"""Per-kernel timing with general penalties (development tool): PYTHONPATH=. python tools/bench_general.py [C2] [n]"""
import sys
import approximate_string_matching_amd as m

eng = m.Engine(0)
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
cfg, _, _ = m.workload(name)
batch = eng.generate(cfg, 0, n)
d = eng.malloc(4 * n)
tm = eng.timer()
for (k, x, o, e) in ((3, 2, 3, 1), (3, 4, 6, 2)):
    p = m.Params.default(k=k, x=x, o=o, e=e)
    for a in (m.NW, m.LEAP, m.GREEDY):
        for it in range(2):
            tm.start(); eng.align_async(batch, a, p, d); tm.stop(); ms = tm.elapsed_ms()
        print((k, x, o, e), m.ALIGNER_NAMES[a], "ms %.3f" % ms, "pairs/s %.3e" % (n / ms * 1e3))
