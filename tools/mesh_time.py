#!/usr/bin/env python3
"""GPU-only frame time of a few scenes at full size (no oracle): for BVH / traversal experiments."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
CASES = [("teapot.json", 1920, 1080, 5), ("dragons.json", 3840, 2160, 5), ("nefertiti.json", 1080, 1800, 5)]
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
for name, w, h, depth in CASES:
    t0 = time.time(); hs = rtc.HostScene.from_file(name); t1 = time.time()
    gpu = rtc.GpuScene(hs.desc); t2 = time.time()
    cam = hs.camera(w, h)
    canvas = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
    for _ in range(4): gpu.render_device(cam, canvas.data_ptr(), depth, None, stream.cuda_stream)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    a.record(stream)
    for _ in range(n): gpu.render_device(cam, canvas.data_ptr(), depth, None, stream.cuda_stream)
    b.record(stream); torch.cuda.synchronize()
    print(f"{name:16s} {a.elapsed_time(b)/n:8.3f} ms   (load {t1-t0:.2f} s, scene create {t2-t1:.2f} s)", flush=True)
