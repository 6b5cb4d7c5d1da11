#!/usr/bin/env python3
"""Generates tests/golden/filter_*.npz from the REAL bit-parallel LEAP (SIMD_ED) and SHD sources
(oracle/_ref/libasm_ref_simd.so = /root/reference/GASMA/benchmark/LEAP_SIMD compiled in place; this container only).
Run from the repo root:  python tests/golden/make_golden_filter.py

Per case (a seeded batch of the product's own generator, inputs pinned by SHA-256): for every (ED threshold, SHD on/off)
the reference's check_pass() and get_ED() per pair, run in batch order after the harness's warm-up pair (oracle/
ref_harness_simd.cpp), and bit_vec_filter_avx's verdict for several error thresholds.  Fixtures are data only."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import approximate_string_matching_amd as asm  # noqa: E402
from tests import oracle_binding  # noqa: E402
from tests.golden.make_golden import inputs_sha  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [("filter_c1", "C1", 0, 3000), ("filter_c2", "C2", 100, 4000), ("filter_c4", "C4", 0, 3000), ("filter_c5", "C5", 200, 3000)]
SIMD_SETTINGS = [(1, 1), (2, 0), (3, 1), (3, 0), (5, 1), (8, 0), (12, 1), (16, 1), (24, 0)]  # (ED threshold, SHD enable)
SHD_ERRORS = [0, 1, 3, 5, 9, 16]
# affine mode, clean (init_affine before every pair): (gap threshold, affine threshold, x, o, e)
AFFINE_SETTINGS = [(3, 60, 2, 3, 1), (5, 40, 1, 1, 1), (8, 100, 4, 6, 2), (2, 30, 3, 5, 2), (10, 25, 1, 2, 1)]
# ... with init_affine's SHD_enable = true: (gap threshold, affine threshold, x, o, e, SHD threshold)
AFFINE_SHD_SETTINGS = [(3, 60, 2, 3, 1, 3), (6, 30, 1, 1, 1, 2), (12, 120, 4, 6, 2, 5), (16, 90, 2, 3, 1, 16)]


def main():
    ref = oracle_binding.load_reference_simd()
    index = {}
    for name, wl, first, n in CASES:
        cfg, _, _ = asm.workload(wl)
        hb = asm.generate_pairs(cfg, first, n)
        out = {}
        for t, shd in SIMD_SETTINGS:
            ed, ps = ref.simd_ed(hb, t, bool(shd))
            out[f"pass_t{t}_shd{shd}"] = ps.astype(np.uint8)
            out[f"ed_t{t}_shd{shd}"] = ed.astype(np.int32)
        for me in SHD_ERRORS:
            out[f"shd_e{me}"] = ref.shd(hb, me).astype(np.uint8)
        for g, af, x, o, e in AFFINE_SETTINGS:
            ed, ps = ref.simd_ed_affine(hb, g, af, x, o, e)
            out[f"af_pass_g{g}_a{af}_x{x}o{o}e{e}"] = ps.astype(np.uint8)
            out[f"af_ed_g{g}_a{af}_x{x}o{o}e{e}"] = ed.astype(np.int32)
        for g, af, x, o, e, st in AFFINE_SHD_SETTINGS:
            ed, ps = ref.simd_ed_affine(hb, g, af, x, o, e, shd_t=st)
            out[f"afs_pass_g{g}_a{af}_x{x}o{o}e{e}_s{st}"] = ps.astype(np.uint8)
            out[f"afs_ed_g{g}_a{af}_x{x}o{o}e{e}_s{st}"] = ed.astype(np.int32)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        index[name] = {"workload": wl, "first": first, "n": n, "inputs_sha256": inputs_sha(hb)}
        print(name, {k: float(v.mean()) for k, v in out.items() if k.startswith("pass")})
    with open(os.path.join(HERE, "filter_index.json"), "w") as fh:
        json.dump({"cases": index, "simd_settings": SIMD_SETTINGS, "shd_errors": SHD_ERRORS, "affine_settings": AFFINE_SETTINGS, "affine_shd_settings": AFFINE_SHD_SETTINGS,
                   "warm_state": list(oracle_binding.SIMD_WARM_STATE)}, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
