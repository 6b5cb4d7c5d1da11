"""The README's four simulated-data result blocks re-run on the device at the README's size (10^6 pairs per error rate, Greedy in
sequential mode = the reference as run): PYTHONPATH=. python tools/readme_lines.py [n]"""
import sys
import numpy as np
import approximate_string_matching_amd as m

README = {0.05: (99.757, 92.975, 97.512), 0.10: (98.066, 78.020, 94.213), 0.15: (93.424, 57.939, 90.418), 0.20: (88.579, 46.023, 88.289)}
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
REF_STREAM = len(sys.argv) > 2 and sys.argv[2] == "ref"   # pairs drawn the reference's way (Dataset over glibc rand(); test infrastructure)
if REF_STREAM:
    from tests import oracle_binding
    orc = oracle_binding.load_oracle()
eng = m.Engine(0)
p = m.Params.default(k=3)
for err, (rl, rg, rc) in README.items():
    for seed in (2000 + int(round(err * 100)), 7000 + int(round(err * 100))):
        if REF_STREAM:
            batch = eng.upload(m.HostBatch(*orc.reference_dataset(n, 100, err, seed)), m.GREEDY_SEQUENTIAL)
        else:
            batch = eng.generate(m.GenConfig.exact(seed, 100, err), 0, n, m.GREEDY_SEQUENTIAL)
        d = [eng.malloc(4 * n) for _ in range(3)]
        d_cnt = eng.malloc(32)
        eng.memset_async(d_cnt, 0, 32)
        eng.run_benchmark_async(batch, p, d[0], d[1], d[2], d_cnt, repack=True)
        cnt = eng.to_host(d_cnt, 8).view(np.uint64)[:4].astype(np.float64)
        cov = eng.coverage(batch, p, window=64)
        se = lambda q: 100 * (q / 100 * (1 - q / 100) * (1 / n + 1e-6)) ** 0.5
        lp, gp, cp = 100 * cnt[2] / n, 100 * cnt[3] / n, 100.0 * cov["covered"] / n
        print("err %.2f seed %d  LEAP %.3f (README %.3f, %+.1f sigma)  Greedy %.3f (README %.3f, %+.1f sigma)  coverage %.3f (README %.3f, %+.3f)  undetermined %d"
              % (err, seed, lp, rl, (lp - rl) / se(rl), gp, rg, (gp - rg) / se(rg), cp, rc, cp - rc, cov["undetermined"]), flush=True)
        for x in d + [d_cnt]:
            eng.free(x)
        batch.free()
