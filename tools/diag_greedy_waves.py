"""Diagnostic build only: per-wave totals of the Greedy kernel laid out by workgroup (which waves of a workgroup finish when)."""
import ctypes, os, sys
import numpy as np
os.environ["ASM_MI355X_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libasm_diag.so")
import approximate_string_matching_amd as m
eng = m.Engine(0)
cfg, _, p = m.workload("C2")
n = 1_000_000
b = eng.generate(cfg, 0, n)
d = eng.malloc(4 * n)
nw = 4096
dbg = eng.malloc(8 * 8 * nw)
eng.memset_async(dbg, 0, 8 * 8 * nw)
eng.lib.asm_diag_set_buffer.argtypes = [ctypes.c_void_p]
eng.lib.asm_diag_set_buffer(dbg)
for _ in range(2):
    eng.align_async(b, m.GREEDY, p, d)
eng.synchronize()
a = eng.to_host(dbg, 8 * nw, np.uint64).reshape(nw, 8)
wpg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
live = a[:, 2] > 0
nwv = int(live.sum())
tot = a[:nwv, 4].astype(float).reshape(-1, wpg)
its = a[:nwv, 2].astype(float).reshape(-1, wpg)
np.set_printoptions(linewidth=200, precision=0, suppress=True)
print("workgroups", tot.shape[0], "waves per workgroup", wpg)
print("mean total by wave slot (k cycles):", tot.mean(axis=0) / 1e3)
print("mean iterations by wave slot:", its.mean(axis=0))
print("first 6 workgroups:\n", tot[:6] / 1e3)
print("per-workgroup max: min %.0f median %.0f max %.0f" % (tot.max(axis=1).min(), np.median(tot.max(axis=1)), tot.max(axis=1).max()))
print("per-workgroup mean: min %.0f median %.0f max %.0f" % (tot.mean(axis=1).min(), np.median(tot.mean(axis=1)), tot.mean(axis=1).max()))
