#!/usr/bin/env python3
"""Frame after frame on fresh scene handles: does the schedule settle, and where?

    python tools/frame_series.py reflection_and_refraction 1920 1080 8 [--handles 4] [--frames 60] [--option name=value]

Per handle one line: the GPU time of every frame (HIP events around each launch, launches enqueued back to back), then
min / median / last.  A handle re-packs its schedule from the frame before while the view stands still for the first
frames only (rtc_capi.hip launch(): the schedule is kept once two packings in a row measured the same within 2 %)."""
import argparse, importlib, os, statistics, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
ap = argparse.ArgumentParser()
ap.add_argument("scene"); ap.add_argument("width", type=int); ap.add_argument("height", type=int); ap.add_argument("depth", type=int, nargs="?", default=5)
ap.add_argument("--handles", type=int, default=4)
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--option", action="append", default=[])
ap.add_argument("--busy-ms", type=float, default=0.0, help="keep the GPU busy this long before the first handle (clocks up)")
ap.add_argument("--wiggle", type=float, default=0.0, help="rotate the camera by +- this angle on alternate frames: every frame is measured and re-packed")
ap.add_argument("--remeasure-at", type=int, default=-1, help="frame at which one frame of another depth forces a new measurement")
a = ap.parse_args()
for opt in a.option:
    n, v = opt.split("=")
    rtc.set_option(n, float(v))
hs = rtc.HostScene.from_file(a.scene if a.scene.endswith(".json") else a.scene + ".json")
cam = hs.camera(a.width, a.height)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
canvas = torch.empty((a.height, a.width, 3), dtype=torch.float64, device="cuda")
if a.busy_ms > 0:
    import time
    x = torch.rand((4096, 4096), device="cuda"); t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < a.busy_ms:
        x = (x @ x).clamp_(0, 1); torch.cuda.synchronize()
for h in range(a.handles):
    g = rtc.GpuScene(hs.desc)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.frames + 1)]
    ev[0].record(stream)
    cams = [cam, cam]
    if a.wiggle != 0.0:   # (the two views are made before the frames are enqueued: the host's share is one call per frame)
        hs.rotate_camera(a.wiggle); cams[0] = hs.camera(a.width, a.height)
        hs.rotate_camera(-a.wiggle); cams[1] = hs.camera(a.width, a.height)
    import time
    host_t = [time.perf_counter()]
    for f in range(a.frames):
        cam = cams[f % 2]
        host_t.append(time.perf_counter())
        g.render_device(cam, canvas.data_ptr(), a.depth - (1 if f == a.remeasure_at else 0), None, stream.cuda_stream)
        ev[f + 1].record(stream)
    stream.synchronize()
    t = [ev[f].elapsed_time(ev[f + 1]) for f in range(a.frames)]
    import numpy as np
    sch = g.schedule()
    items = sch[sch != 0xFFFFFFFF]
    pixels = ((items >> 26) & 63) + 1
    est, got = g.chunk_times(cam)
    print("  schedule: %d packets, %d items, %d of them part of a cut chunk (%d chunks cut), first packets' items %s; measured chunk ticks: total %d max %d" % (
        len(sch), len(items), int((pixels < 64).sum()), len(np.unique((items & 0xFFFFF)[pixels < 64])),
        [int((r != 0xFFFFFFFF).sum()) for r in sch[:8]], int(got.sum()), int(got.max())), flush=True)
    if a.wiggle != 0.0:
        print("  host, us between enqueues: " + " ".join("%.0f" % ((host_t[i + 1] - host_t[i]) * 1e6) for i in range(1, a.frames)))
    print("handle %d %s | %s | min %.3f median-of-last-20 %.3f last %.3f" % (
        h, g.last_kernel_name(), " ".join("%.3f" % x for x in t), min(t), statistics.median(t[-20:]), t[-1]), flush=True)
