"""Affine NW (general penalties) at C2: time per launch and the penalty distribution (development tool):
PYTHONPATH=. python tools/bench_nw_affine.py [C2] [n]; under `rocprofv3 --kernel-trace --stats` it gives the kernel split."""
import sys

import numpy as np

import approximate_string_matching_amd as m

eng = m.Engine(0)
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
cfg, _, _ = m.workload(name)
batch = eng.generate(cfg, 0, n)
d = eng.malloc(4 * n)
tm = eng.timer()
for (x, o, e) in ((2, 3, 1), (4, 6, 2), (1, 2, 1), (3, 5, 2)):
    p = m.Params.default(k=3, x=x, o=o, e=e)
    for it in range(3):
        tm.start(); eng.align_async(batch, m.NW, p, d); tm.stop(); ms = tm.elapsed_ms()
    pen = eng.to_host(d, n)
    q = np.percentile(pen, [50, 90, 99, 100])
    print("pen %s: %.3f ms  penalty mean %.1f p50 %d p90 %d p99 %d max %d; > 2o+14e: %.1f %%, > 2o+30e: %.2f %%, > 2o+46e: %.3f %%"
          % ((x, o, e), ms, pen.mean(), q[0], q[1], q[2], q[3], 100 * (pen > 2 * o + 14 * e).mean(), 100 * (pen > 2 * o + 30 * e).mean(),
             100 * (pen > 2 * o + 46 * e).mean()), flush=True)
del batch
eng.close()
