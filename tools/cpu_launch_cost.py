#!/usr/bin/env python3
"""CPU time of one rtc_render_device call (enqueue only, no sync) for a few rectangle sizes."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
hs = rtc.HostScene.from_file(sys.argv[1] if len(sys.argv) > 1 else "cover.json"); cam = hs.camera(1920, 1080); gpu = rtc.GpuScene(hs.desc)
canvas = torch.empty((1080, 1920, 3), dtype=torch.float64, device="cuda")
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
for rect in ((0, 0, 8, 8), (0, 0, 256, 256), (0, 0, 1920, 1080)):
    for _ in range(5): gpu.render_device(cam, canvas.data_ptr(), 5, rect, stream.cuda_stream)
    torch.cuda.synchronize()
    n = 100
    t0 = time.perf_counter()
    for _ in range(n): gpu.render_device(cam, canvas.data_ptr(), 5, rect, stream.cuda_stream)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(rect, "cpu us/call", round((t1 - t0) / n * 1e6, 1), "wall us/call", round((t2 - t0) / n * 1e6, 1))
T = 64
tx, ty = rtc.tile_grid(1920, 1080, T, T)
for world in (1, 8):
    first, stride, count, padded = rtc.tiles_of_rank(tx * ty, 0, world)
    buf = torch.zeros((padded, T, T, 3), dtype=torch.float64, device="cuda")
    for depth in (5, 0):
        for _ in range(5): gpu.render_tiles_device(cam, buf.data_ptr(), T, T, first, stride, count, depth, stream.cuda_stream)
        torch.cuda.synchronize()
        n = 100
        t0 = time.perf_counter()
        for _ in range(n): gpu.render_tiles_device(cam, buf.data_ptr(), T, T, first, stride, count, depth, stream.cuda_stream)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("tiles world", world, "depth", depth, "cpu us/call", round((t1 - t0) / n * 1e6, 1), "wall us/call", round((t2 - t0) / n * 1e6, 1))
