# development tool: greedy_persist work distribution sweep at C2 (static share / chunk), stand-alone kernel and whole step
for cfg in "0 128" "250 128" "500 128" "500 64" "750 128" "1000 128" "1000 256" "1000 64"; do
  set -- $cfg
  echo "dyn $1 chunk $2"
  ASM_QUEUE_DYN=$1 ASM_QUEUE_CHUNK=$2 timeout -k 10 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-sequential 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.4f'%d['ms_per_step'], d.get('kernels_ms'), d['roofline'].get('avg_launch_ms'))"
done
