#!/usr/bin/env python3
"""Interactive mode: per-call host time of rtc_render_device while the camera orbits (the re-pack every 64 frames)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "cover.json"
hs = rtc.HostScene.from_file(name); gpu = rtc.GpuScene(hs.desc)
canvas = torch.empty((1080, 1920, 3), dtype=torch.float64, device="cuda")
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
times = []
t_all = time.perf_counter()
for i in range(200):
    cam = hs.camera(1920, 1080)
    t0 = time.perf_counter()
    gpu.render_device(cam, canvas.data_ptr(), 5, None, stream.cuda_stream)
    times.append((time.perf_counter() - t0) * 1e3)
    hs.rotate_camera(0.01)
torch.cuda.synchronize()
total = (time.perf_counter() - t_all) * 1e3
big = [(i, round(t, 2)) for i, t in enumerate(times) if t > 0.5]
print("calls slower than 0.5 ms (index, ms):", big)
print("median call ms", sorted(times)[100], "total ms for 200 frames", round(total, 1), "=> ms/frame", round(total / 200, 3))
