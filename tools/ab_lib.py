"""Same-box A/B of two builds of the library (development tool, GPU box):
    PYTHONPATH=. python tools/ab_lib.py approximate-string-matching_amd/libasm_base.so approximate-string-matching_amd/libasm_mi355x.so C3:2e6 C2:1e6
Each (library, workload) is timed in a fresh process (the library path is read at import), `rounds` times, interleaved; prints the
per-kernel medians side by side.  LEAP is timed un-hinted and hinted by the Greedy penalties (the shape of C3's step)."""
import json, os, statistics, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, sys
import approximate_string_matching_amd as m
eng = m.Engine(0)
name, n = sys.argv[1], int(float(sys.argv[2]))
cfg, _, params = m.workload(name)
batch = eng.generate(cfg, 0, n)
d = {a: eng.malloc(4 * n) for a in (m.NW, m.LEAP, m.GREEDY)}
tm = eng.timer()
out = {}
def timed(fn):
    best = 1e9
    for _ in range(4):
        tm.start(); fn(); tm.stop(); best = min(best, tm.elapsed_ms())
    return best
out["pack"] = timed(lambda: eng.pack_async(batch))
out["nw"] = timed(lambda: eng.align_async(batch, m.NW, params, d[m.NW]))
out["greedy"] = timed(lambda: eng.align_async(batch, m.GREEDY, params, d[m.GREEDY]))
out["leap"] = timed(lambda: eng.align_async(batch, m.LEAP, params, d[m.LEAP]))
out["leap_hint_nw"] = timed(lambda: eng.align_hinted_async(batch, m.LEAP, params, d[m.NW], d[m.LEAP]))
out["leap_hint_greedy"] = timed(lambda: eng.align_hinted_async(batch, m.LEAP, params, d[m.GREEDY], d[m.LEAP]))
print(json.dumps(out))
"""
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
works = [a for a in sys.argv[1:] if not a.endswith(".so")] or ["C2:1e6"]
rounds = int(os.environ.get("AB_ROUNDS", "3"))
res = {}
for r in range(rounds):
    for w in works:
        name, n = w.split(":")
        for lib in libs:
            env = dict(os.environ, ASM_MI355X_LIB=os.path.abspath(lib), PYTHONPATH=ROOT)
            p = subprocess.run([sys.executable, "-c", CHILD, name, n], env=env, capture_output=True, text=True, timeout=600)
            if p.returncode != 0:
                print("FAILED", lib, w, p.stderr[-800:], flush=True)
                continue
            d = json.loads(p.stdout.strip().splitlines()[-1])
            for k, v in d.items():
                res.setdefault((w, k), {}).setdefault(lib, []).append(v)
for (w, k), by in sorted(res.items()):
    row = "  ".join("%s %.4f" % (os.path.basename(l)[6:-3], statistics.median(v)) for l, v in by.items())
    print("%-10s %-18s %s" % (w, k, row), flush=True)
