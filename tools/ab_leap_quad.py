"""A/B of the LEAP wide-band kernels (development tool): run once per ASM_LEAP_QUAD setting, prints time and a digest of the
penalties: PYTHONPATH=. ASM_LEAP_QUAD=0|3 python tools/ab_leap_quad.py"""
import hashlib
import os
import sys

import numpy as np

import approximate_string_matching_amd as m

eng = m.Engine(0)
n = 1_000_000
tm = eng.timer()
d = eng.malloc(4 * n)
d_hint = eng.malloc(4 * n)
for name, ks in (("C2", (6, 8, 16, 30, 45)), ("C3", (8, 30, 50))):
    cfg, _, _ = m.workload(name)
    batch = eng.generate(cfg, 0, n)
    for pen in ((1, 1, 1), (2, 3, 1), (4, 6, 2)):
        for k in ks:
            if name == "C3" and pen != (1, 1, 1) and k != 30:
                continue
            p = m.Params.default(k=k, x=pen[0], o=pen[1], e=pen[2])
            hint = None
            if os.environ.get("AB_HINT"):  # schedule by the NW penalties of the same parameters, as asm_run_benchmark_async does
                eng.align_async(batch, m.NW, p, d_hint)
                hint = d_hint
            for it in range(2):
                tm.start(); eng.align_hinted_async(batch, m.LEAP, p, hint, d); tm.stop(); ms = tm.elapsed_ms()
            out = eng.to_host(d, n)
            print("hint=%s quad=%s %s pen %s k=%2d leap %8.3f ms  digest %s  mean %.3f" % (os.environ.get("AB_HINT", "0"), os.environ.get("ASM_LEAP_QUAD", "default"), name, pen, k, ms,
                  hashlib.sha256(out.tobytes()).hexdigest()[:12], out.mean()), flush=True)
    del batch
eng.close()
