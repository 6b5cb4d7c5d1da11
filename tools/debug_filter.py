"""Development aid: PYTHONPATH=. python tools/debug_filter.py WL N T SHD — first GPU/oracle mismatches of the SIMD_ED filter."""
import sys
import numpy as np
import approximate_string_matching_amd as asm
from tests import oracle_binding as ob

wl, n, T, shd = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "1"
orc = ob.load_oracle()
eng = asm.Engine(0)
cfg, _, _ = asm.workload(wl)
hb = asm.generate_pairs(cfg, 41, n)
b = eng.upload(hb, asm.GREEDY_CLEAN)
for mode in (1, 0):
    want, raw, ps = orc.simd_ed(hb, T, shd, mode, ob.SIMD_WARM_STATE)
    got = eng.simd_ed(b, T, shd, mode, ob.SIMD_WARM_STATE)
    bad = np.nonzero(got != want)[0]
    print("mode", mode, "bad", bad.size, bad[:8])
    for i in bad[:4]:
        lo = max(0, i - 3)
        print(" i", i, "got", got[lo:i + 1], "want", want[lo:i + 1], "raw", raw[lo:i + 1])
        a, r = hb.pair(int(i))
        print("  ", a)
        print("  ", r)
