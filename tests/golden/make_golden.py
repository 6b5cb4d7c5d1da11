#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref/libasm_ref.so = /root/reference compiled in
place; this container only).  Run from the repo root:  python tests/golden/make_golden.py

Each fixture holds, for a seeded batch of the product's own generator: a SHA-256 of the inputs (so a generator
change is caught), and the reference's per-pair outputs: Greedy cost + CIGAR digest in both buffer-tail modes, LEAP
get_ED().  NW has no reference-side pin (parasail is absent from the reference tree): its vectors come from the
pure-Python Gotoh DP below, written independently of the oracle's C.  Fixtures are data only."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import approximate_string_matching_amd as asm  # noqa: E402
from tests import oracle_binding  # noqa: E402
from tests.util import KNOWN_PAIRS  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [  # name, workload, first pair, n, k, (x, o, e)
    ("c1_k3", "C1", 0, 3000, 3, (1, 1, 1)),
    ("c2_k3", "C2", 0, 6000, 3, (1, 1, 1)),
    ("c2_k3_late", "C2", 900_000, 2000, 3, (1, 1, 1)),
    ("c3_k30", "C3", 0, 1500, 30, (1, 1, 1)),
    ("c4_k3", "C4", 0, 4000, 3, (1, 1, 1)),
    ("c5_k3", "C5", 0, 3000, 3, (1, 1, 1)),
    ("c2_k3_x2o3e1", "C2", 50, 2000, 3, (2, 3, 1)),
    ("c2_k5_x4o6e2", "C2", 70, 2000, 5, (4, 6, 2)),
    ("c5_k10_x1o2e1", "C5", 90, 1500, 10, (1, 2, 1)),
]


def digest(strings):
    return np.array([int.from_bytes(hashlib.blake2b(s.encode(), digest_size=8).digest(), "little") for s in strings],
                    np.uint64)


def inputs_sha(hb):
    h = hashlib.sha256()
    for a in (hb.read_off, hb.reads, hb.ref_off, hb.refs):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def gotoh_py(a, b, x, o, e):
    """Global affine distance, match 0 / mismatch x / gap(L) = o + (L-1)e — plain Python, three matrices."""
    INF = 10 ** 9
    m, n = len(a), len(b)
    H = [[INF] * (n + 1) for _ in range(m + 1)]
    E = [[INF] * (n + 1) for _ in range(m + 1)]
    F = [[INF] * (n + 1) for _ in range(m + 1)]
    H[0][0] = 0
    for j in range(1, n + 1):
        E[0][j] = o + (j - 1) * e
        H[0][j] = E[0][j]
    for i in range(1, m + 1):
        F[i][0] = o + (i - 1) * e
        H[i][0] = F[i][0]
        for j in range(1, n + 1):
            E[i][j] = min(E[i][j - 1] + e, H[i][j - 1] + o)
            F[i][j] = min(F[i - 1][j] + e, H[i - 1][j] + o)
            H[i][j] = min(H[i - 1][j - 1] + (0 if a[i - 1] == b[j - 1] else x), E[i][j], F[i][j])
    return H[m][n]


def main():
    ref = oracle_binding.load_reference()
    index = {}
    for name, wl, first, n, k, (x, o, e) in CASES:
        cfg, _, _ = asm.workload(wl)
        hb = asm.generate_pairs(cfg, first, n)
        out = {"inputs_sha256": inputs_sha(hb)}
        for mode, tag in ((0, "seq"), (1, "clean")):
            cost, cig = ref.greedy(hb, k=k, x=x, o=o, e=e, mode=mode, cigars=True)
            out[f"greedy_{tag}_cost"] = cost
            out[f"greedy_{tag}_cigar"] = digest(cig)
        out["leap_ed"] = ref.leap(hb, k=k, x=x, o=o, e=e)
        nw_n = 150 if hb.lengths()[0].max() <= 160 else 60
        out["nw_first"] = np.array([gotoh_py(*hb.pair(i), x, o, e) for i in range(nw_n)], np.int32)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **{k2: v for k2, v in out.items() if k2 != "inputs_sha256"})
        index[name] = {"workload": wl, "first": first, "n": n, "k": k, "x": x, "o": o, "e": e,
                       "inputs_sha256": out["inputs_sha256"], "nw_first": nw_n}
        print(name, "greedy mean", out["greedy_clean_cost"].mean(), "leap mean", out["leap_ed"].mean())
    # known-answer vectors: literal pairs of the reference tree, outputs of the compiled reference
    ka = {}
    for key, (a, b) in KNOWN_PAIRS.items():
        hb = asm.HostBatch.from_strings([(a, b)])
        ka[key] = {"read": a, "ref": b, "nw": gotoh_py(a, b, 1, 1, 1)}
        for k in (2, 3):
            cost, cig = ref.greedy(hb, k=k, mode=1, cigars=True)
            ka[key][f"greedy_k{k}"] = int(cost[0])
            ka[key][f"greedy_cigar_k{k}"] = cig[0]
            ka[key][f"leap_k{k}"] = int(ref.leap(hb, k=k)[0])
    with open(os.path.join(HERE, "index.json"), "w") as fh:
        json.dump({"cases": index, "known_answers": ka}, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
