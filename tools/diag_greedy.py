"""Diagnostic build only: where the K=3 Greedy kernel's cycles go (refill block vs step body), from s_memtime stamps."""
import ctypes, os, sys
import numpy as np
os.environ["ASM_MI355X_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libasm_diag.so")
import approximate_string_matching_amd as m
eng = m.Engine(0)
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg, _, p = m.workload(wl)
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
a = a[a[:, 2] > 0]
print("waves", len(a))
tot = a[:, 4].astype(float)
print("per wave: total cycles %.0f  refill %.0f (%.1f%%)  step %.0f (%.1f%%)  iterations %.1f  mean active lanes per iteration %.1f"
      % (tot.mean(), a[:, 0].mean(), 100 * a[:, 0].sum() / tot.sum(), a[:, 1].mean(), 100 * a[:, 1].sum() / tot.sum(),
         a[:, 2].mean(), (a[:, 3] / a[:, 2]).mean()))
print("cycles per iteration: refill %.0f step %.0f" % ((a[:, 0] / a[:, 2]).mean(), (a[:, 1] / a[:, 2]).mean()))
print("wave total cycles: min %.0f median %.0f max %.0f" % (tot.min(), np.median(tot), tot.max()))
b.free(); eng.close()
