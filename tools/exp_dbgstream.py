import os, sys
import approximate_string_matching_amd as m
path = "/tmp/asm_stream_C2_4000000.seq"
eng = m.Engine(0)
_, _, p = m.workload("C2")
for it in range(3):
    if it == 2: os.environ["ASM_STREAM_DEBUG"] = "1"
    res, st = eng.stream_seq_file(path, p, m.GREEDY_CLEAN, chunk_bytes=64 << 20, capacity=4_000_000)
    print("total %.2f ms reader %.2f" % (st.seconds*1e3, st.seconds_read*1e3))
