"""Experiment: how much of the pack's exposed time can be hidden by cutting the batch into chunks and running the NW -> LEAP
chain of chunk c beside the pack of chunk c+1 (development tool; GPU box: PYTHONPATH=. python tools/exp_pipeline.py [chunks])."""
import sys, time
import numpy as np
import torch
import approximate_string_matching_amd as m

C = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = 1_000_000
eng = m.Engine(0)
cfg, _, params = m.workload("C2")
main, packs, side = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
eng.set_stream(main.cuda_stream)
full = eng.generate(cfg, 0, N)
parts = [eng.generate(cfg, c * (N // C), N // C) for c in range(C)]
d = {a: eng.malloc(4 * N) for a in (m.NW, m.LEAP, m.GREEDY)}
dp = [{a: eng.malloc(4 * (N // C)) for a in (m.NW, m.LEAP)} for _ in range(C)]
eng.synchronize()

def use(s):
    eng.set_stream(s.cuda_stream)

def step_serial():
    use(main)
    eng.pack_async(full)
    ev = torch.cuda.Event(); ev.record(main)
    side.wait_event(ev)
    use(side); eng.align_async(full, m.GREEDY, params, d[m.GREEDY])
    ej = torch.cuda.Event(); ej.record(side)
    use(main)
    eng.align_async(full, m.NW, params, d[m.NW])
    eng.align_hinted_async(full, m.LEAP, params, d[m.NW], d[m.LEAP])
    main.wait_event(ej)

def step_chunked():
    # packs of all chunks in a row on their own stream (the full batch's pack stands in for them: same bytes), NW -> LEAP of
    # chunk c as soon as its pack is done, Greedy over everything once the last pack is done
    evs = []
    e0 = torch.cuda.Event(); e0.record(main); packs.wait_event(e0)
    use(packs)
    for c in range(C):
        eng.pack_async(parts[c])
        e = torch.cuda.Event(); e.record(packs); evs.append(e)
    eng.pack_async(full) if False else None
    side.wait_event(evs[-1])
    use(side); eng.align_async(full, m.GREEDY, params, d[m.GREEDY])
    ej = torch.cuda.Event(); ej.record(side)
    use(main)
    for c in range(C):
        main.wait_event(evs[c])
        eng.align_async(parts[c], m.NW, params, dp[c][m.NW])
        eng.align_hinted_async(parts[c], m.LEAP, params, dp[c][m.NW], dp[c][m.LEAP])
    main.wait_event(ej)

use(main); eng.pack_async(full); eng.synchronize()
for name, fn in (("serial", step_serial), ("chunked x%d" % C, step_chunked)):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    K = 30
    for _ in range(K): fn()
    torch.cuda.synchronize()
    print(name, "ms/step %.4f" % ((time.perf_counter() - t) / K * 1e3))
