#!/usr/bin/env python3
"""Static opcode-class histogram of one kernel of the gfx950 assembly (hipcc -S --cuda-device-only output).

    python tools/isa_histogram.py /tmp/rtc_kernels.s rtc_render_kernel_simple

Classes are chosen to line up with the SQ_INSTS_* PMC counters (VALU split into FP64 arithmetic by kind, FP32,
integer / logic, moves and selects, compares, conversions; SALU; LDS; VMEM; SMEM; waits and branches), so that the
static mix can be set against the executed counts of profiles/<round>/pmc_summary.txt."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_"):
        if re.match(r"v_(add|sub)_f64", op): return "valu_f64_add"
        if op.startswith("v_mul_f64"): return "valu_f64_mul"
        if re.match(r"v_(fma|div_fmas)_f64", op): return "valu_f64_fma"
        if re.match(r"v_(rcp|rsq|sqrt)_f64", op): return "valu_f64_trans"
        if re.match(r"v_(div_scale|div_fixup|ldexp|frexp_\w+|trunc|floor|ceil|rndne|fract)_f64", op): return "valu_f64_other"
        if re.match(r"v_(max|min)_f64", op): return "valu_f64_minmax"
        if re.match(r"v_cmpx?_\w+_f64", op) or op.startswith("v_cmp_class_f64"): return "valu_cmp_f64"
        if re.match(r"v_cmpx?_", op): return "valu_cmp_other"
        if re.match(r"v_cvt_", op): return "valu_cvt"
        if re.match(r"v_pk_\w+_f32", op): return "valu_f32_packed"
        if re.search(r"_f32", op): return "valu_f32"
        if re.match(r"v_(mov|accvgpr|swap)", op): return "valu_mov"
        if op.startswith("v_cndmask"): return "valu_cndmask"
        if re.match(r"v_(readlane|readfirstlane|writelane|permlane|bpermute|mbcnt)", op): return "valu_lane"
        return "valu_int_logic"
    if op.startswith("ds_"): return "lds"
    if re.match(r"(global|flat|buffer)_atomic", op): return "vmem_atomic"
    if re.match(r"(global|flat|buffer|scratch)_load", op): return "vmem_load" if not op.startswith("scratch") else "scratch_load"
    if re.match(r"(global|flat|buffer|scratch)_store", op): return "vmem_store" if not op.startswith("scratch") else "scratch_store"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memtime"): return "smem"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if re.match(r"s_(cbranch|branch|setpc|swappc|call)", op): return "s_branch"
    if op.startswith("s_nop") or op.startswith("s_sleep"): return "s_nop"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    counts = collections.Counter()
    ops = collections.Counter()
    inside = False
    for line in open(path):
        if line.startswith(kernel + ":"):
            inside = True
            continue
        if inside and line.startswith(".Lfunc_end"):
            break
        if not inside:
            continue
        m = re.match(r"\s+([a-z_0-9]+)\b", line)
        if not m or line.strip().startswith((".", ";")):
            continue
        op = m.group(1)
        counts[classify(op)] += 1
        ops[op] += 1
    total = sum(counts.values())
    print(f"# {kernel}: {total} instructions (static)")
    for k, v in counts.most_common():
        print(f"{k:20s} {v:7d} {100.0 * v / total:5.1f} %")
    valu = sum(v for k, v in counts.items() if k.startswith("valu"))
    f64 = sum(v for k, v in counts.items() if k in ("valu_f64_add", "valu_f64_mul", "valu_f64_fma", "valu_f64_trans"))
    print(f"# VALU {valu}, of which FP64 add/mul/fma/trans {f64} ({100.0 * f64 / max(valu, 1):.1f} %)")
    if len(sys.argv) > 3:
        print("# top opcodes")
        for k, v in ops.most_common(int(sys.argv[3])):
            print(f"{k:28s} {v:7d}")


if __name__ == "__main__":
    main()
