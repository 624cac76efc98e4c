#!/usr/bin/env python3
"""Formats the passes of tools/pmc_mem.sh: per-launch means of the render kernel's vector-memory counters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarize
d, n = sys.argv[1], int(sys.argv[2])
c = {}
for i in range(1, n + 1):
    try:
        c.update(summarize([os.path.join(d, f"p{i}")]))
    except Exception as e:  # a pass whose counters this device does not have
        print(f"# pass {i}: {e}")
print("# per-launch means of the render kernel (rocprofv3 --pmc, one pass per line of tools/pmc_mem.sh)")
for k in sorted(c):
    print(f"{k:44s} {c[k]:18.1f}")
if "SQ_BUSY_CYCLES" in c:
    cycles = c["SQ_BUSY_CYCLES"] / 32.0
    print(f"# kernel cycles (SQ_BUSY_CYCLES / 32 SEs) {cycles/1e6:.3f} M; x 256 CUs = {cycles*256/1e6:.1f} M CU-cycles")
    for k in ("TA_TA_BUSY_sum", "TCP_GATE_EN1_sum", "TCP_GATE_EN2_sum", "TD_TD_BUSY_sum", "TCP_PENDING_STALL_CYCLES_sum",
              "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum"):
        if k in c:
            print(f"# {k} / CU-cycles = {c[k] / (cycles * 256):.3f}")
