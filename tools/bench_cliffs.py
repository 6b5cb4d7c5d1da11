"""Per-kernel time over the (k, penalties) dispatch grid, to spot slow corners (development tool):
PYTHONPATH=. python tools/bench_cliffs.py [C2] [n]"""
import sys
import approximate_string_matching_amd as m

eng = m.Engine(0)
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
cfg, _, _ = m.workload(name)
batch = eng.generate(cfg, 0, n)
d = eng.malloc(4 * n)
tm = eng.timer()
for pen in ((1, 1, 1), (2, 3, 1), (4, 6, 2)):
    for k in (3, 5, 8, 16, 30, 45):
        p = m.Params.default(k=k, x=pen[0], o=pen[1], e=pen[2])
        row = []
        for a in (m.NW, m.LEAP, m.GREEDY):
            best = 1e9
            for it in range(2):
                tm.start(); eng.align_async(batch, a, p, d); tm.stop(); best = min(best, tm.elapsed_ms())
            row.append("%s %8.3f" % (m.ALIGNER_NAMES[a], best))
        print("pen", pen, "k=%2d" % k, " | ".join(row), "ms")
