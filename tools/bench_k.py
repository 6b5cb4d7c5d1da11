"""LEAP / Greedy kernel time against the band half-width (development tool): PYTHONPATH=. python tools/bench_k.py [C3] [n] [k,k,...]"""
import sys
import approximate_string_matching_amd as m

eng = m.Engine(0)
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
ks = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [5, 6, 7, 8, 10, 12, 15, 16, 20, 24, 30, 31, 32, 40, 50]
cfg, _, _ = m.workload(name)
batch = eng.generate(cfg, 0, n)
d = eng.malloc(4 * n)
tm = eng.timer()
for k in ks:
    p = m.Params.default(k=k)
    row = []
    for a in (m.LEAP, m.GREEDY):
        best = 1e9
        for it in range(3):
            tm.start(); eng.align_async(batch, a, p, d); tm.stop(); best = min(best, tm.elapsed_ms())
        row.append("%s %.3f ms" % (m.ALIGNER_NAMES[a], best))
    print("k=%2d" % k, " | ".join(row), flush=True)
