#!/usr/bin/env python3
"""Instruction mix of one kernel, priced with measured issue costs.

    tools/kernel_mix.py <file.hip> <kernel-name-substring> <ubench.txt> [out.json]

Compiles the translation unit to gfx950 assembly with the library's flags, takes the body of the kernel whose mangled
name contains the substring, and counts its vector instructions per opcode.  The count is static: the kernels this is used
on are one unrolled field-arithmetic body inside a loop, with a few dozen instructions of set-up and search around it
(LOOP_TRIPS > 1 weights each instruction by that many trips per enclosing backward branch instead, which over-weights
short inner search loops).  The opcode counts are then priced with the cycles per wave-instruction per SIMD that
tools/ubench_valu.hip measured at eight wavefronts per SIMD (<ubench.txt> is that program's output); opcodes the
micro-benchmark does not cover are priced like v_add_u32 (plain 32-bit VALU).  The result is the kernel's mean cost per
vector wave-instruction: the VALU-issue bound bench.py prices the measured instruction count at.
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "xelis-he_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-Wno-unused-result", "-Wno-pass-failed"]
LOOP_TRIPS = 1


def assembly(src):
    out = subprocess.run(["hipcc", *FLAGS, "--cuda-device-only", "-S", "-I", CSRC, src, "-o", "-"], check=True, capture_output=True, text=True)
    return out.stdout


def kernel_body(asm, needle):
    lines = asm.splitlines()
    start = None
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m and needle in m.group(1) and start is None:
            start, name = i + 1, m.group(1)
    if start is None:
        raise SystemExit("no kernel matching %r" % needle)
    body = []
    for ln in lines[start:]:
        if ln.startswith(".Lfunc_end"):
            break
        body.append(ln)
    return name, body


def ubench_costs(path):
    """cycles per wave-instruction per SIMD at 8 waves/SIMD, by instruction name"""
    costs = {}
    for ln in open(path):
        m = re.match(r"^(\S+).*waves/SIMD=8 .* ([0-9.]+) cyc/wave-instr/SIMD", ln)
        if m:
            costs[m.group(1)] = float(m.group(2))
    return costs


def main():
    src, needle, ubench = sys.argv[1:4]
    out_path = sys.argv[4] if len(sys.argv) > 4 else None
    name, body = kernel_body(assembly(src), needle)
    # loop depth per line: a backward branch to label L closes a loop that opened at L
    label_at = {}
    for i, ln in enumerate(body):
        m = re.match(r"^(\.LBB\w+):", ln)
        if m:
            label_at[m.group(1)] = i
    depth = [0] * len(body)
    for i, ln in enumerate(body):
        m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\w+)", ln)
        if m and m.group(1) in label_at and label_at[m.group(1)] <= i:
            for j in range(label_at[m.group(1)], i + 1):
                depth[j] += 1
    counts, static = {}, {}
    for i, ln in enumerate(body):
        m = re.match(r"^\s+(v_\w+)", ln)
        if not m:
            continue
        op = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", m.group(1))
        counts[op] = counts.get(op, 0.0) + float(LOOP_TRIPS) ** depth[i]
        static[op] = static.get(op, 0) + 1
    costs = ubench_costs(ubench)
    plain = costs["v_add_u32"]
    alias = {"v_mad_co_u64_u32": "v_mad_u64_u32", "v_add_co_u32": "v_add_u32", "v_addc_co_u32": "v_add_u32", "v_add_co_ci_u32": "v_add_u32",
             "v_lshlrev_b64": "v_lshrrev_b64", "v_ashrrev_i64": "v_lshrrev_b64", "v_mul_u32_u24": "v_mad_u32_u24"}
    total = sum(counts.values())
    cycles = 0.0
    rows = []
    for op, w in sorted(counts.items(), key=lambda kv: -kv[1]):
        cost = costs.get(op, costs.get(alias.get(op, ""), plain))
        cycles += w * cost
        rows.append({"opcode": op, "static": static[op], "weight_frac": round(w / total, 5), "cycles": cost, "measured": op in costs or alias.get(op, "") in costs})
    result = {"kernel": name, "source": os.path.relpath(src, ROOT), "loop_trips_assumed": LOOP_TRIPS, "vector_instructions_static": sum(static.values()),
              "cycles_per_wave_instruction": round(cycles / total, 3),
              "costs_from": os.path.relpath(ubench, ROOT), "costs_note": "cycles per wave-instruction per SIMD at 8 wavefronts per SIMD, 2.4 GHz; unlisted opcodes priced as v_add_u32",
              "mix": [r for r in rows if r["weight_frac"] >= 0.002]}
    text = json.dumps(result, indent=1)
    if out_path:
        open(out_path, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
