#!/usr/bin/env python3
"""GPU time of consecutive frames of one handle, launch by launch (python tools/frame_series.py [scene] [frames] [sync 0|1])."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "cover"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 24
sync = len(sys.argv) <= 3 or sys.argv[3] != "0"
depth = 8 if name.startswith("reflection") else 5
w, h = (3840, 2160) if name == "dragons" else (1920, 1080)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
hs = rtc.HostScene.from_file(name + ".json"); cam = hs.camera(w, h)
canvas = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
gpu = rtc.GpuScene(hs.desc)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(frames)]
for i in range(frames):
    ev[i][0].record(stream)
    gpu.render_device(cam, canvas.data_ptr(), depth, None, stream.cuda_stream)
    ev[i][1].record(stream)
    if sync or i < 12: torch.cuda.synchronize()
torch.cuda.synchronize()
print(name, "sync" if sync else "back to back after frame 12", " ".join(f"{a.elapsed_time(b):.3f}" for a, b in ev), flush=True)
