#!/usr/bin/env python3
"""One rank's share of a frame split N ways, M frames in flight, for profiling (rocprofv3 ... -- python3 tools/share_probe.py):
    share_probe.py [--scene cover.json] [--world 8] [--rank 0] [--inflight 3] [--frames 90] [--tile 64]
Tiles dealt round-robin (first = rank, stride = world); a scene handle and M - 1 clones on M streams."""
import argparse, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cover.json"); ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--depth", type=int, default=5); ap.add_argument("--world", type=int, default=8); ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--option", action="append", default=[])
ap.add_argument("--inflight", type=int, default=3); ap.add_argument("--frames", type=int, default=90); ap.add_argument("--tile", type=int, default=64)
a = ap.parse_args()
for opt in a.option:
    n, v = opt.split("="); rtc.set_option(n, float(v))
hs = rtc.HostScene.from_file(a.scene); cam = hs.camera(a.width, a.height)
tx, ty = rtc.tile_grid(cam.hsize, cam.vsize, a.tile, a.tile)
first, stride, count, padded = rtc.tiles_of_rank(tx * ty, a.rank, a.world)
M = max(1, a.inflight)
streams = [torch.cuda.Stream() for _ in range(M)]
g0 = rtc.GpuScene(hs.desc)
handles = [g0] + [g0.clone() for _ in range(M - 1)]
bufs = [torch.zeros((padded, a.tile, a.tile, 3), dtype=torch.float64, device="cuda") for _ in range(M)]
def frame(i):
    handles[i % M].render_tiles_device(cam, bufs[i % M].data_ptr(), a.tile, a.tile, first, stride, count, a.depth, streams[i % M].cuda_stream)
for i in range(8 * M):
    frame(i); torch.cuda.synchronize()
s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s0.record(streams[0])
for st in streams[1:]: st.wait_event(s0)
for i in range(a.frames): frame(i)
for st in streams[1:]: streams[0].wait_stream(st)
s1.record(streams[0]); torch.cuda.synchronize()
print("%s: rank %d of %d, %d tiles, %d frames in flight: %.4f ms per frame, %s" % (a.scene, a.rank, a.world, count, M, s0.elapsed_time(s1) / a.frames, handles[0].last_kernel_name()))
