set -e
mkdir -p gpurun_out/r3
for f in 0 1; do for w in 1 2 3; do
  echo "FAST=$f WAVES=$w"
  ASM_GREEDY_FAST=$f ASM_PERSIST_WAVES=$w PYTHONPATH=. timeout -k 10 100 python tools/bench_quick.py C2 1e6 2>&1 | grep -E "greedy"
  ASM_GREEDY_FAST=$f ASM_PERSIST_WAVES=$w timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sequential --no-standalone 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  step ms', d['ms_per_step'], 'value %.3e' % d['value'])"
done; done
