#!/usr/bin/env python3
"""Prints VGPR/SGPR/scratch/LDS/occupancy per kernel of the product library (hipcc remarks; no GPU needed)."""
import os, re, subprocess, sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "approximate-string-matching_amd")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
       "-Rpass-analysis=kernel-resource-usage", "-shared", "-o", "/tmp/_asm_res.so", "csrc/asm_capi.hip"]
out = subprocess.run(cmd, cwd=root, capture_output=True, text=True).stderr
cur, rows = None, []
for line in out.splitlines():
    m = re.search(r"remark:\s*(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|"
                  r"Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" ")[0]] = v
for r in rows:
    if "hipcub" in r["name"] or "rocprim" in r["name"]:
        continue
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    print("%-42s vgpr=%-4s sgpr=%-4s scratch=%-5s occ=%-2s lds=%s" % (
        name.replace("void ", ""), r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")))
