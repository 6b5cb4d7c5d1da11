"""Filtering-stage timings (development tool): PYTHONPATH=. python tools/bench_filter.py [C2] [n]
GPU: SIMD_ED (bit-parallel LEAP, Levenshtein, ED_GLOBAL) with/without its SHD pre-filter, and the stand-alone SHD, per
kernel with HIP events; CPU beside it on a bounded sample: the oracle (port) and, when oracle/_ref/libasm_ref_simd.so was
built, the real reference sources."""
import sys
import time

import numpy as np

import approximate_string_matching_amd as m
from tests import oracle_binding as ob

eng = m.Engine(0)
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
cfg, _, _ = m.workload(name)
batch = eng.generate(cfg, 0, n)
d = eng.malloc(4 * n)
tm = eng.timer()


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        tm.start(); fn(); tm.stop()
        best = min(best, tm.elapsed_ms())
    return best


for T in (3, 5, 10, 16):
    for shd in (True, False):
        ms = timed(lambda: eng.simd_ed_async(batch, T, d, shd, m.FILTER_CLEAN))
        ps = (eng.to_host(d, n) >= 0).mean()
        print("simd_ed T=%2d shd=%d clean      ms %.3f pairs/s %.3e pass %.4f" % (T, shd, ms, n / ms * 1e3, ps))
    t0 = time.perf_counter(); eng.simd_ed_async(batch, T, d, True, m.FILTER_SEQUENTIAL, ob.SIMD_WARM_STATE); eng.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    print("simd_ed T=%2d shd=1 sequential ms %.3f pairs/s %.3e (wall, incl. scratch allocation)" % (T, ms, n / ms * 1e3))
    ms = timed(lambda: eng.shd_filter_async(batch, T, d))
    print("shd     e=%2d                   ms %.3f pairs/s %.3e pass %.4f" % (T, ms, n / ms * 1e3, eng.to_host(d, n).mean()))

# SIMD_ED affine mode, clean (init_affine(gap, af, ED_GLOBAL, x, o, e) per pair)
for g, af, x, o, e in ((3, 60, 2, 3, 1), (3, 20, 1, 1, 1), (8, 60, 2, 3, 1), (8, 120, 4, 6, 2), (30, 60, 2, 3, 1)):
    ms = timed(lambda: eng.simd_ed_affine_async(batch, g, af, x, o, e, d))
    ps = (eng.to_host(d, n) >= 0).mean()
    print("simd_ed affine gap=%2d af=%3d pen (%d,%d,%d) ms %.3f pairs/s %.3e pass %.4f" % (g, af, x, o, e, ms, n / ms * 1e3, ps))

# CPU side, bounded sample
ns = min(n, 200_000)
hb = m.generate_pairs(cfg, 0, ns)
orc = ob.load_oracle()
for T in (3,):
    t0 = time.perf_counter(); orc.simd_ed(hb, T, True, 0, ob.SIMD_WARM_STATE); dt = time.perf_counter() - t0
    print("cpu port  simd_ed T=%d shd=1: %.3e pairs/s (1 thread, %d pairs)" % (T, ns / dt, ns))
    t0 = time.perf_counter(); orc.shd(hb, T); dt = time.perf_counter() - t0
    print("cpu port  shd e=%d:          %.3e pairs/s" % (T, ns / dt))
    if ob.have_reference_simd():
        ref = ob.load_reference_simd()
        t0 = time.perf_counter(); ref.simd_ed(hb, T, True); dt = time.perf_counter() - t0
        print("cpu reference simd_ed T=%d shd=1: %.3e pairs/s (1 thread, incl. string conversion)" % (T, ns / dt))
        t0 = time.perf_counter(); ref.shd(hb, T); dt = time.perf_counter() - t0
        print("cpu reference shd e=%d:          %.3e pairs/s (incl. string conversion)" % (T, ns / dt))
        small = m.generate_pairs(cfg, 0, 20_000)
        t0 = time.perf_counter(); ref.simd_ed_affine(small, 3, 60, 2, 3, 1); dt = time.perf_counter() - t0
        print("cpu reference simd_ed affine gap=3 af=60 (2,3,1): %.3e pairs/s (init_affine before every pair)" % (20_000 / dt))
        t0 = time.perf_counter(); orc.simd_ed_affine(small, 3, 60, 2, 3, 1); dt = time.perf_counter() - t0
        print("cpu port      simd_ed affine gap=3 af=60 (2,3,1): %.3e pairs/s" % (20_000 / dt))
