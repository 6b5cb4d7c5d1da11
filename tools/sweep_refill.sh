for g in 1 4 8 16 24 32 48; do echo "refill $g"; ASM_REFILL_GREEDY=$g ASM_REFILL_LEAP=$g PYTHONPATH=. timeout -k 10 100 python tools/bench_quick.py C2 1e6 2>&1 | grep -E "leap|greedy"; done
