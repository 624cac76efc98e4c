#!/usr/bin/env python3
"""How long does the heaviest 8x8 chunk of a frame take when it has the GPU to itself?  (lower bound of any launch)"""
import argparse, importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cover.json")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--depth", type=int, default=5)
args = ap.parse_args()
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
hs = rtc.HostScene.from_file(args.scene)
cam = hs.camera(args.width, args.height)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); sptr = stream.cuda_stream
gpu = rtc.GpuScene(hs.desc)
buf = torch.empty((args.height, args.width, 3), dtype=torch.float64, device="cuda")
def timed(rect, reps=5):
    for _ in range(2): gpu.render_device(cam, buf.data_ptr(), args.depth, rect, sptr)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(reps): gpu.render_device(cam, buf.data_ptr(), args.depth, rect, sptr)
    b.record(stream); torch.cuda.synchronize()
    st = gpu.stats()
    return a.elapsed_time(b) / reps, st["primary"] + st["secondary"], st["shadow_traced"]
res = []
for y in range(0, args.height - 8, 40):
    for x in range(0, args.width - 8, 40):
        t, rays, sh = timed((x, y, 8, 8))
        res.append((t, x, y, rays, sh))
res.sort(reverse=True)
print("empty-ish launch (cheapest 8x8):", res[-1])
print("heaviest 8x8 chunks:", res[:5])
t, x, y, _, _ = res[0]
for w, h in ((8, 8), (8, 1), (1, 1), (16, 16), (64, 64), (256, 256)):
    x0, y0 = min(x, args.width - w), min(y, args.height - h)   # keep the rectangle inside the image
    print((w, h), timed((x0, y0, w, h), 10))
