#!/usr/bin/env python3
"""One fresh scene handle, three launches (for a kernel trace of the first frame): tools/first_frame_one.py scene w h depth"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
name, w, h, depth = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
hs = rtc.HostScene.from_file(name); cam = hs.camera(w, h)
canvas = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
gpu = rtc.GpuScene(hs.desc)
torch.cuda.synchronize()
for i in range(3):
    gpu.render_device(cam, canvas.data_ptr(), depth, None, stream.cuda_stream)
    torch.cuda.synchronize()
