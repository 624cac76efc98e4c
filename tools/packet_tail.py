#!/usr/bin/env python3
"""Which packets of the steady-state schedule are long, and where they sit in the hand-out order.
Run with RTC_TIME_ALWAYS=1 RTC_TIME_DUMP=<file> RTC_PROFILE_DUMP=1 (tools/packet_times.py writes the file)."""
import sys
import numpy as np
rows = np.loadtxt(sys.argv[1])
idx, items, now, pred = rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3]
n = len(rows)
fair = now.sum() / 2048.0
print(f"{n} packets, sum {now.sum():.0f} units, fair share of one of 2048 waves {fair:.0f}")
for lo, hi in ((0, 0.01), (0.01, 0.05), (0.05, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 0.9), (0.9, 0.95), (0.95, 0.98), (0.98, 1.0)):
    a, b = int(lo * n), int(hi * n)
    s = slice(a, b)
    print(f"order {lo:4.2f}-{hi:4.2f}: {b - a:5d} packets, items/packet {items[s].mean():5.2f}, now mean {now[s].mean():8.0f} max {now[s].max():8.0f}"
          f" | predicted mean {pred[s].mean():8.0f} | now/pred {now[s].sum() / max(pred[s].sum(), 1):5.2f}")
o = np.argsort(-now)[:12]
print("longest packets (order index, items, now, predicted, now as a share of the fair share):")
for i in o:
    print(f"  {int(idx[i]):6d} {int(items[i]):3d} {now[i]:9.0f} {pred[i]:9.0f} {now[i] / fair:5.2f}")
late = np.argsort(-(now * (idx > 0.9 * n)))[:8]
print("longest packets of the last 10 % of the order:")
for i in late:
    print(f"  {int(idx[i]):6d} {int(items[i]):3d} {now[i]:9.0f} {pred[i]:9.0f} {now[i] / fair:5.2f}")
if rows.shape[1] >= 5 and len(sys.argv) > 2:
    chunks_x = int(sys.argv[2])   # chunks per image row (width / 8): where the slow packets of the cheap end are
    print("slowest packets of the last 10 %: chunk column, row (of the first item), now, predicted")
    for i in late:
        c = int(rows[i, 4]); print(f"  x {c % chunks_x:4d} y {c // chunks_x:4d}  {now[i]:8.0f} {pred[i]:8.0f}")
