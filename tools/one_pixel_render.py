#!/usr/bin/env python3
"""Renders one small rectangle a few times (for rocprofv3 passes over a single wave's work):
    python tools/one_pixel_render.py cover.json 1920 1080 X Y W H [depth] [launches]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
scene, w, h, x, y, rw, rh = sys.argv[1], *[int(a) for a in sys.argv[2:8]]
depth = int(sys.argv[8]) if len(sys.argv) > 8 else 5
n = int(sys.argv[9]) if len(sys.argv) > 9 else 8
hs = rtc.HostScene.from_file(scene); cam = hs.camera(w, h)
g = rtc.GpuScene(hs.desc)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
buf = torch.empty((rh, rw, 3), dtype=torch.float64, device="cuda")
for i in range(n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream); g.render_device(cam, buf.data_ptr(), depth, (x, y, rw, rh), stream.cuda_stream); b.record(stream); torch.cuda.synchronize()
    print("launch", i, "ms", round(a.elapsed_time(b), 4), g.last_kernel_name(), flush=True)
print(g.stats())
