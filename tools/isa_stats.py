#!/usr/bin/env python3
"""Static ISA statistics per kernel of the product library (instruction mix; no GPU needed).
usage: tools/isa_stats.py [substring-of-kernel-name ...]"""
import os, re, subprocess, sys
from collections import Counter

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "approximate-string-matching_amd")
out = "/tmp/_asm_capi.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S",
                "--cuda-device-only", "-o", out, "csrc/asm_capi.hip"], cwd=root, check=True, capture_output=True)
s = open(out).read()
want = sys.argv[1:]
for m in re.finditer(r"^(_Z[\w]+):\s*; @", s, re.M):
    name = m.group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
    if "hipcub" in dem or "rocprim" in dem:
        continue
    if want and not any(w in dem for w in want):
        continue
    end = s.find(".Lfunc_end", m.start())
    body = s[m.end():end]
    ops = [l.split()[0] for l in (x.strip() for x in body.split("\n"))
           if l and not l.startswith((".", ";", "//")) and not l.endswith(":")]
    c = Counter(ops)
    v = sum(n for k, n in c.items() if k.startswith("v_"))
    print(f"{dem:34s} total={len(ops):6d} valu={v:6d} salu={sum(n for k, n in c.items() if k.startswith('s_')):6d} "
          f"b64shift={sum(n for k, n in c.items() if 'b64' in k and 'sh' in k):5d} f64={sum(n for k, n in c.items() if 'f64' in k):4d} "
          f"cndmask={c.get('v_cndmask_b32_e32', 0) + c.get('v_cndmask_b32_e64', 0):5d} branch={sum(n for k, n in c.items() if 'branch' in k):4d}")
    if want:
        print("    ", c.most_common(25))
