"""PCIe-inclusive rate of the host-buffer boundary (development tool): PYTHONPATH=. python tools/bench_host_path.py [C2] [n]
Host ASCII + offsets in, three penalty arrays out: asm_batch_upload (H2D + pack), asm_run_benchmark_async (NW, LEAP, Greedy,
counters), three D2H copies — wall clock, best of 3.  Never the bench.py `value` (that one starts with inputs resident)."""
import sys
import time

import numpy as np

import approximate_string_matching_amd as m

eng = m.Engine(0)
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
cfg, _, p = m.workload(name)
hb = m.generate_pairs(cfg, 0, n)
d = [eng.malloc(4 * n) for _ in range(3)]
d_cnt = eng.malloc(32)
in_bytes = hb.reads.nbytes + hb.refs.nbytes + hb.read_off.nbytes + hb.ref_off.nbytes
best = None
for it in range(4):
    t0 = time.perf_counter()
    b = eng.upload(hb, m.GREEDY_CLEAN)
    eng.synchronize()
    t1 = time.perf_counter()
    eng.memset_async(d_cnt, 0, 32)
    eng.run_benchmark_async(b, p, d[0], d[1], d[2], d_cnt, repack=False)
    eng.synchronize()
    t2 = time.perf_counter()
    out = [eng.to_host(x, n) for x in d]
    t3 = time.perf_counter()
    del b
    row = (t3 - t0, t1 - t0, t2 - t1, t3 - t2)
    if it and (best is None or row[0] < best[0]):
        best = row
tot, up, run, down = best
print("%s n=%d  input %.1f MB, output %.1f MB" % (name, n, in_bytes / 1e6, 12 * n / 1e6))
print("upload+pack %.2f ms (%.1f GB/s) | NW+LEAP+Greedy+counters %.2f ms | copy back %.2f ms (%.1f GB/s)"
      % (up * 1e3, in_bytes / up / 1e9, run * 1e3, down * 1e3, 12 * n / down / 1e9))
print("PCIe-inclusive: %.3e pairs/s through all three aligners (%.2f ms per 1e6 pairs)" % (n / tot, tot * 1e3 * 1e6 / n))

# ---- the file path: read_string_file + run through asm_stream_seq_file (reader threads -> pinned buffers -> HBM, parse on
# the device, aligners overlapped with the next chunk's transfer); the file is in the page cache
import os
import tempfile

n_file = int(float(sys.argv[3])) if len(sys.argv) > 3 else 4 * n
path = os.path.join(tempfile.gettempdir(), "asm_stream_%s_%d.seq" % (name, n_file))
if not os.path.exists(path):
    t0 = time.perf_counter()
    from approximate_string_matching_amd import HostBatch
    with open(path, "wb") as fh:
        step = 500_000
        for lo in range(0, n_file, step):
            part = m.generate_pairs(cfg, lo, min(step, n_file - lo))
            ml, nl = part.lengths()
            if (ml == ml[0]).all():  # fixed-length reads: assemble the text with numpy instead of a Python loop
                L = int(ml[0])
                rows = []
                rd = part.reads.reshape(-1, L)
                for i in range(part.n):
                    rows.append(b">" + rd[i].tobytes() + b"\n<" + part.refs[part.ref_off[i]:part.ref_off[i + 1]].tobytes() + b"\n")
                fh.write(b"".join(rows))
            else:
                for i in range(part.n):
                    a, b2 = part.pair(i)
                    fh.write((">%s\n<%s\n" % (a, b2)).encode())
    print("wrote %s (%.1f MB) in %.1f s" % (path, os.path.getsize(path) / 1e6, time.perf_counter() - t0))
size = os.path.getsize(path)
with open(path, "rb") as fh:  # page cache
    while fh.read(1 << 26):
        pass
for mode, mname in ((m.GREEDY_CLEAN, "clean"), (m.GREEDY_SEQUENTIAL, "sequential")):
    for chunk_mb in (16, 64, 256):
        best = None
        for it in range(3):
            res, st = eng.stream_seq_file(path, p, mode, chunk_bytes=chunk_mb << 20, capacity=n_file)
            if it and (best is None or st.seconds < best.seconds):
                best = st
        print("stream %-10s chunk %3d MB: %d pairs, %d chunks, %.1f MB in %.2f ms = %.3e pairs/s end to end (%.1f GB/s of file; reader busy %.2f ms)"
              % (mname, chunk_mb, best.pairs, best.chunks, size / 1e6, best.seconds * 1e3, best.pairs / best.seconds, size / best.seconds / 1e9,
                 best.seconds_read * 1e3), flush=True)
