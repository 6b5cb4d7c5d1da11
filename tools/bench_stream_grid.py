"""Streaming ingest over reader-thread counts and chunk sizes (development tool; the file of tools/bench_host_path.py must exist:
PYTHONPATH=. python tools/bench_stream_grid.py /tmp/asm_stream_C2_4000000.seq).  ASM_READER_THREADS is read per call."""
import os, sys, time
import approximate_string_matching_amd as m
path = sys.argv[1]
eng = m.Engine(0)
_, _, p = m.workload("C2")
size = os.path.getsize(path)
with open(path, "rb") as fh:
    while fh.read(1 << 26):
        pass
for threads in (6, 8, 10, 12):
    os.environ["ASM_READER_THREADS"] = str(threads)
    for chunk_mb in (32, 48, 64, 96, 128):
        best = None
        for it in range(4):
            res, st = eng.stream_seq_file(path, p, m.GREEDY_CLEAN, chunk_bytes=chunk_mb << 20, capacity=4_000_000)
            if it and (best is None or st.seconds < best.seconds):
                best = st
        print("threads %2d chunk %3d MB: %.2f ms = %.3e pairs/s (reader busy %.2f ms)" % (threads, chunk_mb, best.seconds * 1e3, best.pairs / best.seconds, best.seconds_read * 1e3), flush=True)
