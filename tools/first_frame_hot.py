#!/usr/bin/env python3
"""Is the first frame of a fresh scene handle slow because of its schedule, or because the GPU is cold?

    python tools/first_frame_hot.py [scene W H depth]

Handle B is created first and left alone; handle A then renders 60 frames (clocks up, caches warm); B's FIRST frame - on
the estimate's schedule, measuring, with the packer behind it - follows A's last frame on the same stream without a gap.
Beside it: the first frame of a handle on a GPU that idled while the handle was created (what tools/first_frame.py and
bench.py's first_frame_ms report), and A's steady frame."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
cases = [("cover", 1920, 1080, 5), ("teapot", 1920, 1080, 5), ("dragons", 3840, 2160, 5), ("reflection_and_refraction", 1920, 1080, 8)]
if len(sys.argv) > 4:
    cases = [(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))]
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); sp = stream.cuda_stream
for name, w, h, depth in cases:
    hs = rtc.HostScene.from_file(name + ".json"); cam = hs.camera(w, h)
    canvas = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
    def timed(g, n=1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(n): g.render_device(cam, canvas.data_ptr(), depth, None, sp)
        b.record(stream)
        return a, b
    cold = []
    for _ in range(3):
        g = rtc.GpuScene(hs.desc); torch.cuda.synchronize()
        a, b = timed(g); torch.cuda.synchronize(); cold.append(a.elapsed_time(b)); g.close()
    hot = []
    for _ in range(3):
        B = rtc.GpuScene(hs.desc); A = rtc.GpuScene(hs.desc); torch.cuda.synchronize()
        timed(A, 60)
        a, b = timed(B)                 # enqueued behind A's frames: no idle gap in front of it
        s0, s1 = timed(A, 10)
        torch.cuda.synchronize()
        hot.append(a.elapsed_time(b)); steady = s0.elapsed_time(s1) / 10
        A.close(); B.close()
    print("%-12s first frame, GPU idle before it %.3f ms | first frame behind 60 frames of another handle %.3f ms | steady %.3f ms" % (
        name[:12], sorted(cold)[1], sorted(hot)[1], steady), flush=True)
