#!/usr/bin/env python3
"""Formats the passes of tools/pmc_mix.sh: wave-level instructions per launch by class, and what is left over."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarize
d = sys.argv[1]
c = summarize([os.path.join(d, x) for x in ("a", "b", "c")])
valu = c["SQ_INSTS_VALU"]
f64 = [("FP64 add (incl. compares / min / max?)", "SQ_INSTS_VALU_ADD_F64"), ("FP64 mul", "SQ_INSTS_VALU_MUL_F64"), ("FP64 fma", "SQ_INSTS_VALU_FMA_F64"),
       ("FP64 transcendental (rcp, rsq, sqrt)", "SQ_INSTS_VALU_TRANS_F64")]
f32 = [("FP32 add", "SQ_INSTS_VALU_ADD_F32"), ("FP32 mul", "SQ_INSTS_VALU_MUL_F32"), ("FP32 fma", "SQ_INSTS_VALU_FMA_F32"), ("FP32 transcendental", "SQ_INSTS_VALU_TRANS_F32")]
oth = [("INT32", "SQ_INSTS_VALU_INT32"), ("INT64", "SQ_INSTS_VALU_INT64"), ("conversions", "SQ_INSTS_VALU_CVT")]
print("# per-launch means of the render kernel (rocprofv3 --pmc, three passes); wave-level instruction counts")
print(f"{'VALU instructions':44s} {valu/1e6:10.2f} M   lanes active {c['SQ_THREAD_CYCLES_VALU'] / 64.0 / c['SQ_ACTIVE_INST_VALU']:.3f}")
known = 0.0
for name, key in f64 + f32 + oth:
    v = c.get(key, 0.0); known += v
    print(f"  {name:42s} {v/1e6:10.2f} M  {100.0*v/valu:5.1f} %")
print(f"  {'not in any class above (moves, selects, compares, bit ops, div_scale / div_fmas / div_fixup, ldexp, frexp ...)':42s} {(valu-known)/1e6:10.2f} M  {100.0*(valu-known)/valu:5.1f} %")
for name, key in (("SALU", "SQ_INSTS_SALU"), ("branches", "SQ_INSTS_BRANCH"), ("SMEM", "SQ_INSTS_SMEM"), ("VMEM reads", "SQ_INSTS_VMEM_RD"), ("VMEM writes", "SQ_INSTS_VMEM_WR"), ("LDS", "SQ_INSTS_LDS")):
    print(f"{name:44s} {c.get(key, 0.0)/1e6:10.2f} M")
cycles = c["SQ_BUSY_CYCLES"] / 32.0
print(f"{'kernel cycles (SQ_BUSY_CYCLES / 32 SEs)':44s} {cycles/1e6:10.3f} M   vector pipe issuing {c['SQ_ACTIVE_INST_VALU'] * 4.0 / 1024.0 / cycles:.3f}   wave cycles waiting {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.3f}")
