"""A/B of library switches on the whole bench step (development tool, GPU box):
python tools/ab_step.py "ASM_GREEDY_WAVES=2" "ASM_GREEDY_WAVES=3" ... — every configuration is run `rounds` times, interleaved, each
in a fresh process; prints ms/step per run and the median."""
import json, os, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds = int(os.environ.get("AB_ROUNDS", "3"))
configs = sys.argv[1:] or [""]
res = {c: [] for c in configs}
for r in range(rounds):
    for c in configs:
        env = dict(os.environ)
        for kv in c.split():
            k, v = kv.split("=", 1)
            env[k] = v
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "5", "--no-cpu-baseline",
                              "--no-sequential", "--no-standalone", "--no-in-order"], env=env, capture_output=True, text=True, timeout=300)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        res[c].append(d["ms_per_step"])
        print(f"round {r} [{c}] {d['ms_per_step']:.4f}", flush=True)
for c in configs:
    print(f"[{c}] median {statistics.median(res[c]):.4f} ms/step  min {min(res[c]):.4f}  runs {['%.4f' % v for v in res[c]]}")
