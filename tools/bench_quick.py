"""Per-kernel timing for one workload (development tool; run on the GPU box as PYTHONPATH=. python tools/bench_quick.py C2 1e6)."""
import sys, time
import numpy as np
import approximate_string_matching_amd as m

eng = m.Engine(0)
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
cfg, _, params = m.workload(name)
t = time.time(); batch = eng.generate(cfg, 0, n); eng.synchronize(); print("generate+pack", time.time() - t, "maxlen", batch.max_length)
d = eng.malloc(4 * n)
tm = eng.timer()
for a in (m.NW, m.LEAP, m.GREEDY):
    for it in range(3):
        tm.start(); eng.align_async(batch, a, params, d); tm.stop(); ms = tm.elapsed_ms()
    print(m.ALIGNER_NAMES[a], "ms", ms, "pairs/s %.3e" % (n / ms * 1e3))
for it in range(3):
    tm.start(); eng.pack_async(batch); tm.stop(); ms = tm.elapsed_ms()
print("pack ms", ms, "pairs/s %.3e" % (n / ms * 1e3))
