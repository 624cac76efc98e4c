#!/usr/bin/env python3
"""Diagnostic: the per-tile costs four ranks measure (rtc_get_tile_costs) for their round-robin shares of cover 400x230
when their launches are enqueued together on four streams of ONE device (what librtc_multi's virtual ranks do)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
hs = rtc.HostScene.from_file("cover.json"); cam = hs.camera(400, 230)
T, world = 64, 4
tx, ty = rtc.tile_grid(400, 230, T, T)
streams = [torch.cuda.Stream() for _ in range(world)]
gs = [rtc.GpuScene(hs.desc) for _ in range(world)]
shares = [rtc.tiles_of_rank(tx * ty, r, world) for r in range(world)]
bufs = [torch.zeros((shares[r][3], T, T, 3), dtype=torch.float64, device="cuda") for r in range(world)]
for frame in range(2):
    for r in range(world):
        first, stride, count, padded = shares[r]
        tiles = np.arange(first, tx * ty, stride, dtype=np.uint32)
        gs[r].render_tile_list_device(cam, bufs[r].data_ptr(), T, T, tiles, 5, streams[r].cuda_stream)
    torch.cuda.synchronize()
    cost = np.zeros(tx * ty)
    for r in range(world):
        first, stride, count, padded = shares[r]
        c = gs[r].tile_costs(count)
        cost[first::stride] = c
        print("frame", frame, "rank", r, np.round(c).astype(int).tolist(), gs[r].last_kernel_name())
    rank_of, _ = rtc.assign_tiles(cost, world)
    load = np.array([cost[rank_of == r].sum() for r in range(world)])
    print("frame", frame, "max/mean after LPT", load.max() / load.mean())
