#!/usr/bin/env python3
"""Per-wave timeline of one simulated rank's share (needs a -DRTC_PROFILE build: python tools/variants.py "prof=-DRTC_PROFILE" -- python tools/wave_log.py ...)."""
import argparse, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cover.json")
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--tile", type=int, default=64)
ap.add_argument("--frames", type=int, default=4)
ap.add_argument("--log", default="gpurun_out/wave_log.txt")
args = ap.parse_args()
os.environ["RTC_PROFILE_DUMP"] = "1"
os.environ["RTC_PROFILE_LOG"] = args.log
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
hs = rtc.HostScene.from_file(args.scene)
cam = hs.camera(1920, 1080)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); sptr = stream.cuda_stream
g = rtc.GpuScene(hs.desc)
tx, ty = rtc.tile_grid(cam.hsize, cam.vsize, args.tile, args.tile)
first, stride, count, padded = rtc.tiles_of_rank(tx * ty, args.rank, args.world)
buf = torch.zeros((padded, args.tile, args.tile, 3), dtype=torch.float64, device="cuda")
for i in range(args.frames):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    g.render_tiles_device(cam, buf.data_ptr(), args.tile, args.tile, first, stride, count, 5, sptr)
    b.record(stream); torch.cuda.synchronize()
    print("frame", i, "ms", a.elapsed_time(b))
st = g.stats()
rows = np.loadtxt(args.log)
life = rows[:, 1] / 100.0   # s_memtime ticks at 100 MHz -> us
iters, units, last_fetch = rows[:, 2], rows[:, 3], rows[:, 4] * 256 / 100.0
print("waves", len(rows), "rays", st["primary"] + st["secondary"])
for name, v in (("lifetime_us", life), ("iterations", iters), ("packets", units), ("last_fetch_us", last_fetch)):
    print(f"{name:14s} min {v.min():8.1f} p10 {np.percentile(v,10):8.1f} med {np.median(v):8.1f} p90 {np.percentile(v,90):8.1f} max {v.max():8.1f} sum {v.sum():12.0f}")
print("us per iteration (lifetime/iterations): med", np.median(life / np.maximum(iters, 1)))
o = np.argsort(-life)[:8]
print("longest waves: life, iters, packets, last_fetch_us, last packet index")
for i in o:
    print("  ", life[i], iters[i], units[i], last_fetch[i], int(rows[i, 5]))
print("last packet index of all waves: min", int(rows[:, 5].min()), "median", int(np.median(rows[:, 5])), "max", int(rows[:, 5].max()))
if rows.shape[1] >= 22:
    # sections of each wave's LAST packet (prof[] deltas since its last fetch; slot 7 = iterations)
    last = rows[:, 6:22] / 100.0
    names = ["pop/store", "closest", "after-closest", "shadow", "lighting", "behind", "after-behind", "iters*100", "share", "deal/pull",
             "precomp", "pattern", "lights-setup", "spawn", "counter", "items"]
    tail = life - last_fetch
    for label, sel in (("all waves", np.arange(len(rows))), ("the 5 % of waves that end last", np.argsort(-life)[:len(rows) // 20])):
        print(f"last packet of {label}: time after the last fetch mean {tail[sel].mean():.0f}, iterations {last[sel, 7].mean() * 100:.1f}")
        print("   " + " | ".join(f"{n} {last[sel, k].mean():.0f}" for k, n in enumerate(names) if n != "-" and k != 7))
