#!/usr/bin/env python3
"""Where in a kernel's loops an opcode pattern sits: python tools/asm_loops.py kernel.s [regex]
Loop nesting is read off the back edges of the assembly (a branch to an earlier label); prints each matching
instruction with its depth and the innermost loop's span."""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"scratch_")
labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB[0-9_]+):", l))}
loops = []
for j, l in enumerate(lines):
    m = re.search(r"\s(s_cbranch_\w+|s_branch)\s+(\.LBB[0-9_]+)", l)
    if m and m.group(2) in labels and labels[m.group(2)] < j:
        loops.append((labels[m.group(2)], j))
for i, l in enumerate(lines):
    if pat.search(l):
        inside = sorted([lp for lp in loops if lp[0] <= i <= lp[1]], key=lambda lp: lp[1] - lp[0])
        inner = inside[0] if inside else None
        print(f"{i:6d} depth {len(inside)} inner {inner} {l.strip()[:70]}")
print("loops:", sorted(set(loops), key=lambda lp: lp[0]))
