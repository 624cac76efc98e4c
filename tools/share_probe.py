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
ap.add_argument("--handles-first", action="store_true", help="create the scene handles before the streams")
ap.add_argument("--extra-streams", type=int, default=0, help="streams created (and left unused) before the M that are used")
ap.add_argument("--set-current", action="store_true", help="make streams[0] torch's current stream")
ap.add_argument("--list", action="store_true", help="the same tiles as a tile LIST (rtc_render_tile_list_device)")
ap.add_argument("--balanced", action="store_true", help="with --all-ranks: the cost-balanced tile lists of rtc_assign_tiles instead of strided tiles")
ap.add_argument("--all-ranks", action="store_true", help="every rank's share one after the other in this process (handles made and closed per rank, the streams stay)")
ap.add_argument("--inflight", type=int, default=3); ap.add_argument("--frames", type=int, default=90); ap.add_argument("--tile", type=int, default=64)
a = ap.parse_args()
for opt in a.option:
    n, v = opt.split("="); rtc.set_option(n, float(v))
hs = rtc.HostScene.from_file(a.scene); cam = hs.camera(a.width, a.height)
tx, ty = rtc.tile_grid(cam.hsize, cam.vsize, a.tile, a.tile)
M = max(1, a.inflight)
if a.all_ranks:
    streams = [torch.cuda.Stream() for _ in range(M)]
    if os.environ.get("TOUCH_STREAMS"):
        for st in streams:
            with torch.cuda.stream(st): torch.zeros(1, device="cuda")
        torch.cuda.synchronize()
    import numpy as np
    rank_of = None
    if a.balanced:
        cost = np.zeros(tx * ty)
        for rank in range(a.world):
            first, stride, count, padded = rtc.tiles_of_rank(tx * ty, rank, a.world)
            g = rtc.GpuScene(hs.desc); buf = torch.zeros((padded, a.tile, a.tile, 3), dtype=torch.float64, device="cuda")
            g.render_tiles_device(cam, buf.data_ptr(), a.tile, a.tile, first, stride, count, a.depth, streams[0].cuda_stream)
            cost[first::stride] = g.tile_costs(count); g.close()
        rank_of, _ = rtc.assign_tiles(cost, a.world)
    order = list(range(a.world))
    if os.environ.get("RANKS_REVERSED"): order.reverse()
    for rank in order:
        first, stride, count, padded = rtc.tiles_of_rank(tx * ty, rank, a.world)
        g0 = rtc.GpuScene(hs.desc); handles = [g0] + [g0.clone() for _ in range(M - 1)]
        padded = (tx * ty + a.world - 1) // a.world
        bufs = [torch.zeros((padded, a.tile, a.tile, 3), dtype=torch.float64, device="cuda") for _ in range(M)]
        mine = np.flatnonzero(rank_of == rank).astype(np.uint32) if rank_of is not None else None
        def frame(i):
            if mine is not None:
                handles[i % M].render_tile_list_device(cam, bufs[i % M].data_ptr(), a.tile, a.tile, mine, a.depth, streams[i % M].cuda_stream)
                return
            handles[i % M].render_tiles_device(cam, bufs[i % M].data_ptr(), a.tile, a.tile, first, stride, count, a.depth, streams[i % M].cuda_stream)
        for i in range(8 * M):
            frame(i); torch.cuda.synchronize()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record(streams[0])
        for st in streams[1:]: st.wait_event(s0)
        for i in range(a.frames): frame(i)
        for st in streams[1:]: streams[0].wait_stream(st)
        s1.record(streams[0]); torch.cuda.synchronize()
        alone = []
        for k in range(M):   # every handle by itself, frame after frame: its own frames' latency
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(streams[k])
            for _ in range(20): frame(k)
            e1.record(streams[k]); torch.cuda.synchronize()
            alone.append(round(e0.elapsed_time(e1) / 20, 4))
        print("   handles alone:", alone, end=" ")
        sch = [h.schedule() for h in handles]
        print("rank %d: %.4f ms per frame" % (rank, s0.elapsed_time(s1) / a.frames), "| tiles", len(mine) if mine is not None else count,
              "| packets per handle", [len(x) for x in sch], "| cut items", [int(((((x[x != 0xFFFFFFFF]) >> 26) & 63) < 63).sum()) for x in sch],
              "|", handles[0].last_kernel_name(), flush=True)
        for h in handles: h.close()
    sys.exit(0)
first, stride, count, padded = rtc.tiles_of_rank(tx * ty, a.rank, a.world)
if a.handles_first:
    g0 = rtc.GpuScene(hs.desc)
    handles = [g0] + [g0.clone() for _ in range(M - 1)]
unused = [torch.cuda.Stream() for _ in range(a.extra_streams)]
streams = [torch.cuda.Stream() for _ in range(M)]
if a.set_current: torch.cuda.set_stream(streams[0])
if not a.handles_first:
    g0 = rtc.GpuScene(hs.desc)
    handles = [g0] + [g0.clone() for _ in range(M - 1)]
bufs = [torch.zeros((padded, a.tile, a.tile, 3), dtype=torch.float64, device="cuda") for _ in range(M)]
import numpy as np
mine = np.arange(first, tx * ty, stride, dtype=np.uint32)[:count]
def frame(i):
    if a.list:
        handles[i % M].render_tile_list_device(cam, bufs[i % M].data_ptr(), a.tile, a.tile, mine, a.depth, streams[i % M].cuda_stream)
        return
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
