#!/usr/bin/env python3
"""What the packets of a schedule cost, in schedule order: python tools/schedule_profile.py [scene W H depth].
A view that wiggles by 1e-7 rad is measured every frame, so the chunk times the handle holds are those of a frame that ran
on a measured schedule; the schedule read back is the one packed from them.  Prints, per twentieth of the schedule
(by packet index): packets, chunks per packet, mean and max packet time (the chunks' measured times added up, in the
packer's units: s_memtime ticks / 16) and the share of the frame's total time in that twentieth."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
name, w, h, depth = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else ("cover", 1920, 1080, 5)
hs = rtc.HostScene.from_file(name + ".json")
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
canvas = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
g = rtc.GpuScene(hs.desc)
hs.rotate_camera(1e-7); cams = [hs.camera(w, h)]; hs.rotate_camera(-1e-7); cams.append(hs.camera(w, h))
for f in range(12):
    g.render_device(cams[f % 2], canvas.data_ptr(), depth, None, stream.cuda_stream)
torch.cuda.synchronize()
sch = g.schedule()                     # packed from frame 11's measurement: what frame 12 runs
est, got = g.chunk_times(cams[1])      # ... and that measurement
g.render_device(cams[0], canvas.data_ptr(), depth, None, stream.cuda_stream)   # frame 12, measured as well
torch.cuda.synchronize()
est2, got2 = g.chunk_times(cams[0])    # what the packets of `sch` really took (a packet's time, shared among its items)
valid = sch != 0xFFFFFFFF
chunk = sch & 0xFFFFF
pixels = ((sch >> 26) & 63) + 1
whole = (pixels == 64)
t_item = np.where(valid, got[np.minimum(chunk, len(got) - 1)] * (pixels / 64.0), 0.0)
t_packet = t_item.sum(axis=1)
t_actual = np.where(valid & whole, got2[np.minimum(chunk, len(got2) - 1)], 0.0).sum(axis=1)   # (packets of whole chunks only)
n_items = valid.sum(axis=1)
total = t_packet.sum()
waves = 3072 if "3" in g.last_kernel_name() else 2048
print("%s %dx%d: %s, %d packets, %d chunks; total %.0f units, a wave's share %.0f (%d waves); heaviest packet %.0f = %.2f shares" % (
    name, w, h, g.last_kernel_name(), len(sch), len(got), total, total / waves, waves, t_packet.max(), t_packet.max() / (total / waves)))
n = len(sch)
for k in range(20):
    a, b = n * k // 20, n * (k + 1) // 20
    tp = t_packet[a:b]
    ok = (valid[a:b] & ~whole[a:b]).sum(axis=1) == 0      # packets without runs of cut chunks
    ratio = t_actual[a:b][ok].sum() / max(1.0, tp[ok].sum())
    print("  packets %6d-%6d: items per packet %.2f, packet time mean %8.0f max %8.0f, %.1f %% of the frame's time; the frame after: x %.2f" % (
        a, b, n_items[a:b].mean(), tp.mean(), tp.max(), 100.0 * tp.sum() / total, ratio))
