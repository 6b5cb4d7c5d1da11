"""Every kernel family stays parity-green, not only the default dispatch: the switches the C-ABI library keeps (environment
variables read at asm_create; one fallback per kernel family, the allocator, the stream layout) select the alternative paths,
and each is checked against the oracle in a fresh process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
import approximate_string_matching_amd as asm
from tests import oracle_binding
orc = oracle_binding.load_oracle()
eng = asm.Engine(0)
def check(name, got, want):
    bad = int((got != want).sum())
    assert bad == 0, (name, bad)
for wl, n, k in (("C2", 12000, 3), ("C5", 6000, 3), ("C3", 1500, 30), ("C2", 2000, 10)):
    cfg, _, _ = asm.workload(wl)
    hb = asm.generate_pairs(cfg, 77, n)
    p = asm.Params.default(k=k)
    b = eng.upload(hb, asm.GREEDY_SEQUENTIAL)
    d = [eng.malloc(4 * n) for _ in range(3)]
    d_cnt = eng.malloc(32)
    eng.memset_async(d_cnt, 0, 32)
    eng.run_benchmark_async(b, p, d[0], d[1], d[2], d_cnt, repack=True)
    check(wl + " nw", eng.to_host(d[0], n), orc.nw(hb))
    check(wl + " leap", eng.to_host(d[1], n), orc.leap(hb, k=k))
    check(wl + " greedy", eng.to_host(d[2], n), orc.greedy(hb, k=k, mode=0))
    cost, cig, _ = eng.greedy_with_cigar(b, p, cap=96)
    want_cost, want_cig = orc.greedy(hb, k=k, mode=0, cigars=True)
    check(wl + " cigar cost", cost, want_cost)
    assert cig == want_cig, wl
    if wl in ("C2", "C5"):
        pg = asm.Params.default(k=k, x=2, o=3, e=1)
        bc = eng.upload(hb, asm.GREEDY_CLEAN)
        check(wl + " nw affine", eng.align(bc, asm.NW, pg), orc.nw(hb, 2, 3, 1))
        check(wl + " leap general", eng.align(bc, asm.LEAP, pg), orc.leap(hb, k, 2, 3, 1))
        check(wl + " greedy general", eng.align(bc, asm.GREEDY, pg), orc.greedy(hb, k, 2, 3, 1, mode=1))
        for g, af in ((3, 40), (9, 70)):
            check(wl + " simd_ed affine", eng.simd_ed_affine(bc, g, af, 2, 3, 1), orc.simd_ed_affine(hb, g, af, 2, 3, 1)[0])
print("ok")
""" % ROOT


@pytest.mark.parametrize("env", [
    {"ASM_GREEDY_FAST": "0"},                      # FP64 lane-refilling Greedy at k <= 3
    {"ASM_LEAP_HINT": "0"},                        # LEAP in input order
    {"ASM_LEAP_SORT": "0"},                        # wide-band LEAP: work sort inside workgroups only
    {"ASM_NW_BANDED": "0", "ASM_NW_WFA": "0"},     # full-height bit-parallel NW, full-matrix affine NW
    {"ASM_NW_BYLEN": "0"},                         # mixed-length NW without the length sort
    {"ASM_WAVE": "0"},                             # workgroup-per-pair fallbacks
    {"ASM_BUCKET": "0"},                           # one width class
    {"ASM_OVERLAP": "0"},                          # one stream
    {"ASM_POOL": "0"},                             # plain hipMalloc / hipFree
])
def test_alternative_kernels_match_the_oracle(env):
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=e, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (env, out.stdout[-500:], out.stderr[-1500:])
