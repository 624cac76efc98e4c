#!/usr/bin/env python3
"""When do the waves of a steady frame begin, fetch their last packet and end?  Needs the light diagnostic build:

    python tools/variants.py "lite=-DRTC_PROFILE -DRTC_PROFILE_LITE" -- python tools/wave_ends.py [scene W H depth]

(no section stamps: the frame runs at nearly its product speed).  Prints, as fractions of the frame (first wave's begin to
last wave's end): percentiles of the waves' begin, last fetch and end, the machine's occupancy over time (waves alive in
every tenth of the frame), and the mean of (wave lifetime / frame) - what a perfectly balanced frame would gain."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
log = "gpurun_out/wave_ends_log.txt"
os.environ["RTC_PROFILE_DUMP"] = "1"; os.environ["RTC_PROFILE_LOG"] = log
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
name, w, h, depth = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else ("cover", 1920, 1080, 5)
hs = rtc.HostScene.from_file(name + ".json"); cam = hs.camera(w, h)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
canvas = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
g = rtc.GpuScene(hs.desc)
for i in range(30):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream); g.render_device(cam, canvas.data_ptr(), depth, None, stream.cuda_stream); b.record(stream); torch.cuda.synchronize()
ms = a.elapsed_time(b)
g.stats()                      # (dumps the log of the last launch)
rows = np.loadtxt(log)
units = rows[:, 3]
begin, end, fetch = rows[:, 6], rows[:, 7], rows[:, 8]     # s_memrealtime: 100 MHz, one clock for all XCDs
t0, t1 = begin.min(), end.max()
T = t1 - t0
pct = lambda v: " ".join("%.3f" % np.percentile(v, p) for p in (0, 10, 50, 90, 100))
print("%s %dx%d depth %d: %s, frame %.3f ms by events, %.3f ms first begin to last end; %d waves, %.1f packets per wave" % (
    name, w, h, depth, g.last_kernel_name(), ms, T / 1e5, len(rows), units.mean()))
print("fractions of the frame, percentiles 0 / 10 / 50 / 90 / 100 over the waves")
print("  begin          ", pct((begin - t0) / T))
print("  last fetch     ", pct((fetch - t0) / T))
print("  end            ", pct((end - t0) / T))
print("  last packet    ", pct((end - fetch) / T), "(end - last fetch)")
alive = [(int(((begin <= t0 + T * (k + 0.5) / 20) & (end > t0 + T * (k + 0.5) / 20)).sum())) for k in range(20)]
print("waves alive at the middle of every twentieth of the frame:", alive)
print("mean wave lifetime / frame = %.3f" % ((end - begin).mean() / T))
