#!/usr/bin/env python3
"""Collects every counter bench.py's `roofline` quotes and WRITES profiles/r04_pmc.json itself (run on the GPU box from the
repo root: `python3 tools/pmc_collect.py --head <git sha> [--workload C2]`).

Passes, each its own rocprofv3 run over bench.py (counters never share a run with tracing, and FETCH_SIZE / WRITE_SIZE do not
fit one pass: MI355X_MICROARCH.md §rocprofv3 PMC slots):
  1. --pmc FETCH_SIZE                    2. --pmc WRITE_SIZE
  3. --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  4. --pmc SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
plus tools/ubench/valu_rates (issue cost per instruction with FP32 control rows, shader clock measured in-kernel) and a static
instruction histogram of the dominant kernel (hipcc -S) to weight those costs into one cycles-per-instruction figure.
HBM bytes follow the guide's gfx950 corrections: FETCH_SIZE (KiB) x 2 for wide coalesced reads, WRITE_SIZE (KiB) as read.
The file is keyed by the digest of the kernel sources (bench.kernel_source_digest): bench.py refuses numbers measured on other
code.  Everything is also copied to gpurun_out/final/ so that it comes back from the box."""
import argparse
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_digest, PMC_FILE)

FAMILIES = (("greedy", "greedy_"), ("leap", "leap_"), ("nw", "nw_"), ("pack", "pack_kernel"))
PASSES = {
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
    "sq_a": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY",
             "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"],
    "sq_b": ["SQ_BUSY_CYCLES", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE"],
}


def run_pass(name, counters, out_dir, bench_args, reuse=False):
    d = os.path.join(out_dir, name)
    if not reuse:
        shutil.rmtree(d, ignore_errors=True)
        env = dict(os.environ, TMPDIR="/tmp")
        cmd = ["rocprofv3", "--pmc", *counters, "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.join(ROOT, "bench.py"), *bench_args]
        with open(os.path.join(out_dir, name + ".log"), "w") as log:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, timeout=1500)
        if r.returncode != 0:
            raise SystemExit(f"pass {name} failed (rc {r.returncode}); see {out_dir}/{name}.log")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(f"pass {name}: {len(agg)} kernels", flush=True)
    return agg


_ASM_TEXT = None


def product_isa():
    global _ASM_TEXT
    if _ASM_TEXT is None:
        pkg = os.path.join(ROOT, "approximate-string-matching_amd")
        out = "/tmp/_asm_capi.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S",
                        "--cuda-device-only", "-o", out, "csrc/asm_capi.hip"], cwd=pkg, check=True, capture_output=True)
        _ASM_TEXT = open(out).read()
    return _ASM_TEXT


def isa_mix(kernel_name, rows):
    """Static instruction histogram of the named kernel instantiation (hipcc -S of the product source) weighted with the ubench
    rows."""
    s = product_isa()
    best = None
    want = kernel_name.replace(" ", "")
    for m in re.finditer(r"^(_Z[\w]+):\s*; @", s, re.M):
        dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        if dem.replace("void ", "").replace(" ", "").startswith(want + "("):
            end = s.find(".Lfunc_end", m.start())
            best = s[m.end():end]
            break
    if best is None:
        return None
    ops = [ln.split()[0] for ln in (x.strip() for x in best.split("\n"))
           if ln and not ln.startswith((".", ";", "//")) and not ln.endswith(":")]
    valu = collections.Counter(o for o in ops if o.startswith("v_"))

    def cyc(name):
        # "cycles_span": first wave's loop start to last wave's loop end over (waves per SIMD x instructions per wave) — what the
        # SIMD needs per instruction with 8 waves resident.  (The per-wave figure "cycles" undercounts: issue arbitration is
        # oldest-first, so the waves of a SIMD finish one after the other, not together.)
        r = rows.get(name + "@8")
        return r["cycles_span"] if r else None

    table = {"v_xor_b32": "k_xor", "v_add_u32": "k_add", "v_sub_u32": "k_add", "v_subrev_u32": "k_add", "v_and_b32": "k_and",
             "v_or_b32": "k_and", "v_not_b32": "k_not", "v_mov_b32": "c_mov", "v_lshlrev_b32": "k_shl", "v_lshrrev_b32": "k_shr_v",
             "v_ashrrev_i32": "k_shr_v", "v_and_or_b32": "k_and_or", "v_or3_b32": "k_or3", "v_lshl_or_b32": "k_lshl_or",
             "v_add3_u32": "k_add3", "v_bfe_u32": "k_bfe", "v_bfi_b32": "k_bfi", "v_alignbit_b32": "k_alignbit",
             "v_ffbl_b32": "k_ffbl", "v_ffbh_u32": "k_ffbh", "v_bcnt_u32_b32": "k_bcnt", "v_min_u32": "k_min", "v_min_i32": "k_min",
             "v_max_i32": "k_max_i", "v_max_u32": "k_max_i", "v_med3_i32": "k_med3", "v_max3_i32": "k_max3", "v_mul_lo_u32": "k_mul_lo",
             "v_cndmask_b32": "k_cmp_cnd", "v_perm_b32": "k_perm", "v_dot4_u32_u8": "k_dot4", "v_bitop3_b32": "k_bitop3",
             "v_lshlrev_b64": "k_shl64", "v_lshrrev_b64": "k_shr64v", "v_lshl_add_u64": "k_add64", "v_fma_f64": "k_fma64",
             "v_mul_f64": "k_mul64", "v_cvt_f64_i32": "k_cvt", "v_readlane_b32": "k_readlane", "v_mbcnt_lo_u32_b32": "k_mbcnt",
             "v_mbcnt_hi_u32_b32": "k_mbcnt"}
    total = weighted = 0.0
    unknown = collections.Counter()
    for op, cnt in valu.items():
        base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
        row = table.get(base)
        if row is None:
            if base.startswith("v_cmp"):
                row = "k_cmp_cnd"
            elif "b64" in base or "u64" in base or "f64" in base:
                row = "k_shr64v"
            else:
                row = "k_alignbit"  # generic 32-bit VOP3
                unknown[base] += cnt
        c = cyc(row)
        if c is None:
            continue
        total += cnt
        weighted += cnt * c
    return {"cycles_per_inst_mix": weighted / total if total else None, "static_valu_instructions": int(total),
            "unlisted_ops_priced_as_generic_vop3": dict(unknown.most_common(12)),
            "note": "static histogram of the kernel's ISA, each opcode priced with its ubench row (cycles_span, 8 waves/SIMD)"}


def add_mixes(kernels, ubench):
    """Every measured kernel gets the issue cost of ITS instruction mix (static histogram x ubench rows); a family of width-class
    kernels the instruction-weighted mean of its members' mixes."""
    for short, e in kernels.items():
        members = e.get("members")
        if members:
            tot = w = 0.0
            for m in members:
                mix = isa_mix(m["kernel_name"], ubench["rows"])
                if mix and mix.get("cycles_per_inst_mix") and m.get("insts_valu"):
                    m["cycles_per_inst_mix"] = mix["cycles_per_inst_mix"]
                    tot += m["insts_valu"]
                    w += m["insts_valu"] * mix["cycles_per_inst_mix"]
            if tot:
                e["valu_mix"] = {"cycles_per_inst_mix": w / tot,
                                 "note": "instruction-weighted mean over the family's width-class kernels (`members`), each a static "
                                         "histogram of its ISA priced with the ubench rows"}
            continue
        mix = isa_mix(e["kernel_name"], ubench["rows"])
        if mix:
            e["valu_mix"] = mix


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--remix", action="store_true",
                    help="no GPU: recompute the per-kernel instruction-mix costs of an existing profiles/r04_pmc.json (same sources)")
    ap.add_argument("--from-raw", action="store_true",
                    help="no GPU: rebuild the json from the raw rocprofv3 output of the last run (gpurun_out/final/pmc) and the ubench "
                         "rows of the existing json (same workload, same sources)")
    ap.add_argument("--head", default="unknown", help="git HEAD of the tree being measured (the box has no .git)")
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=0, help="pairs per launch (default: the workload's batch capped at 1e6)")
    args = ap.parse_args()
    if args.remix:
        with open(bench.pmc_path(args.workload)) as fh:
            doc = json.load(fh)
        if doc.get("source_digest") != bench.kernel_source_digest():
            raise SystemExit("the kernel sources changed since the counters were collected: re-run on the GPU box")
        for k in ("cycles_per_inst_mix", "static_valu_instructions", "unlisted_ops_priced_as_generic_vop3", "note", "mix_kernel"):
            doc["ubench"].pop(k, None)
        add_mixes(doc["kernels"], doc["ubench"])
        with open(bench.pmc_path(args.workload), "w") as fh:
            json.dump(doc, fh, indent=1)
        for short, e in doc["kernels"].items():
            print(short, e["kernel_name"], e.get("valu_mix", {}).get("cycles_per_inst_mix"))
        return
    out_dir = os.path.join(ROOT, "gpurun_out", "final", "pmc")
    os.makedirs(out_dir, exist_ok=True)
    bench_args = ["--workload", args.workload, "--steps", str(args.steps), "--warmup", "1", "--no-cpu-baseline",
                  "--no-sequential", "--no-standalone", "--no-in-order"] + (["--pairs", str(args.pairs)] if args.pairs else [])
    data = {name: run_pass(name, ctrs, out_dir, bench_args, reuse=args.from_raw) for name, ctrs in PASSES.items()}

    import approximate_string_matching_amd as asm

    _, n_default, _ = asm.workload(args.workload)
    pairs = args.pairs or min(n_default, 1_000_000)
    keys = {"insts_valu": ("sq_a", "SQ_INSTS_VALU"), "insts_salu": ("sq_a", "SQ_INSTS_SALU"),
            "active_inst_valu": ("sq_a", "SQ_ACTIVE_INST_VALU"), "wave_cycles": ("sq_a", "SQ_WAVE_CYCLES"),
            "waves": ("sq_a", "SQ_WAVES"), "wait_any": ("sq_a", "SQ_WAIT_ANY"),
            "wait_inst_any": ("sq_a", "SQ_WAIT_INST_ANY"), "active_inst_any": ("sq_a", "SQ_ACTIVE_INST_ANY"),
            "thread_cycles_valu": ("sq_b", "SQ_THREAD_CYCLES_VALU"), "busy_cycles": ("sq_b", "SQ_BUSY_CYCLES"),
            "insts_vmem_rd": ("sq_b", "SQ_INSTS_VMEM_RD"), "insts_lds": ("sq_b", "SQ_INSTS_LDS"),
            "grbm_gui_active": ("sq_b", "GRBM_GUI_ACTIVE")}
    kernels = {}
    for short, prefix in FAMILIES:
        # The family's kernels of the timed region = those launched most often.  A mixed-length batch launches one kernel per
        # width class, all equally often: their counters are SUMMED (the bench line divides them by the family's time), the
        # member with the most VALU instructions lends its name, and `members` lists every one of them.
        cand = [(len(v.get("SQ_WAVES", [])), k) for k, v in data["sq_a"].items() if prefix in k.split("(")[0]]
        if not cand:
            continue
        top = max(c for c, _ in cand)
        names = sorted(k for c, k in cand if c == top)
        mean = lambda p, c, name: (sum(data[p][name][c]) / len(data[p][name][c])) if data[p].get(name, {}).get(c) else None  # noqa: E731
        member = {}
        for name in names:
            fetch_kib, write_kib = mean("fetch", "FETCH_SIZE", name), mean("write", "WRITE_SIZE", name)
            m = {"kernel_name": name.split("(")[0].replace("void ", ""), "launches": top, "fetch_kib_raw": fetch_kib, "write_kib": write_kib,
                 "fetch_bytes": fetch_kib * 1024 * 2 if fetch_kib is not None else None,
                 "write_bytes": write_kib * 1024 if write_kib is not None else None}
            if fetch_kib is not None and write_kib is not None:
                m["traffic_bytes"] = m["fetch_bytes"] + m["write_bytes"]
            for key, (p, c) in keys.items():
                m[key] = mean(p, c, name)
            member[name] = m
        lead = max(names, key=lambda k: member[k].get("insts_valu") or 0)
        e = dict(member[lead])
        if len(names) > 1:
            for key in list(e):
                if key in ("kernel_name", "launches"):
                    continue
                vals = [member[k].get(key) for k in names]
                e[key] = sum(vals) if all(v is not None for v in vals) else None
            e["members"] = [{"kernel_name": member[k]["kernel_name"], "insts_valu": member[k].get("insts_valu"),
                             "traffic_bytes": member[k].get("traffic_bytes")} for k in names]
        if e.get("thread_cycles_valu") and e.get("active_inst_valu"):
            e["lane_util"] = e["thread_cycles_valu"] / (64.0 * e["active_inst_valu"])
        kernels[short] = e
        print(short, json.dumps(e), flush=True)

    # issue costs + clock
    if args.from_raw:
        with open(bench.pmc_path(args.workload)) as fh:
            prev = json.load(fh)
        ubench = prev["ubench"]
        args.head = prev.get("git_head", args.head)
    else:
        ub_dir = os.path.join(ROOT, "tools", "ubench")
        exe = os.path.join(ub_dir, "valu_rates")
        if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(exe + ".hip"):
            subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-Wno-unused-value", "-o", exe, exe + ".hip"], check=True)
        txt = subprocess.run([exe], capture_output=True, text=True, check=True, timeout=600).stdout
        with open(os.path.join(ROOT, "gpurun_out", "final", "valu_rates.txt"), "w") as fh:
            fh.write(txt)
        ub = json.loads(txt.strip().splitlines()[-1][len("JSON "):])
        ubench = {"sclk_hz": ub["sclk_hz_median"], "rows": ub["rows"], "source": "tools/ubench/valu_rates.hip (profiles/r04_valu_rates.txt)"}
    add_mixes(kernels, ubench)
    doc = {"_comment": "written by tools/pmc_collect.py; HBM bytes = FETCH_SIZE KiB x 2 (gfx950 tallies 128-B requests at 64 B for "
                       "wide coalesced reads) + WRITE_SIZE KiB, per launch, mean over the launches of the run",
           "git_head": args.head, "source_digest": bench.kernel_source_digest(), "workload": args.workload, "pairs": pairs,
           "simd_count": 1024, "kernels": kernels, "ubench": ubench}
    target = bench.pmc_path(args.workload)
    for path in (target, os.path.join(ROOT, "gpurun_out", "final", os.path.basename(target))):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            json.dump(doc, fh, indent=1)
    print("wrote", target)


if __name__ == "__main__":
    main()
