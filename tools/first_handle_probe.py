#!/usr/bin/env python3
"""Diagnostic: is the first scene handle of a process slower, or the first hundred milliseconds of GPU work?  One handle,
frame time (mean of 12 frames) after 6, 50, 200, 600 and 1500 frames; then a second handle after 6 frames."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
hs = rtc.HostScene.from_file("cover.json"); cam = hs.camera(1920, 1080)
canvas = torch.empty((1080, 1920, 3), dtype=torch.float64, device="cuda")
def timed(gpu):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(12): gpu.render_device(cam, canvas.data_ptr(), 5, None, stream.cuda_stream)
    b.record(stream); torch.cuda.synchronize()
    return a.elapsed_time(b) / 12
gpu = rtc.GpuScene(hs.desc)
done = 0
out = []
for upto in (6, 50, 200, 600, 1500):
    while done < upto:
        gpu.render_device(cam, canvas.data_ptr(), 5, None, stream.cuda_stream); done += 1
    torch.cuda.synchronize()
    out.append(f"after {upto}: {timed(gpu):.3f}"); done += 12
gpu.close()
g2 = rtc.GpuScene(hs.desc)
for _ in range(6): g2.render_device(cam, canvas.data_ptr(), 5, None, stream.cuda_stream)
torch.cuda.synchronize()
out.append(f"second handle after 6: {timed(g2):.3f}")
print(" | ".join(out))
