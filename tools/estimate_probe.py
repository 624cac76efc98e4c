#!/usr/bin/env python3
"""How good is the first frame's estimate?  Per 8x8 chunk: what rtc_estimate_kernel says it costs against what the first
(measuring) launch measured (rtc_get_chunk_times), per BASELINE config: totals, correlation, and the measured time of the
chunks by decile of their estimate - what a calibration of the roots' weights (rtc_capi.hip, buildRootTables) goes by.
    python tools/estimate_probe.py [scene w h depth]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
cases = [("cover.json", 1920, 1080, 5), ("reflection_and_refraction.json", 1920, 1080, 8), ("teapot.json", 1920, 1080, 5),
         ("dragons.json", 3840, 2160, 5), ("fresnel.json", 300, 300, 5)]
if len(sys.argv) > 4: cases = [(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))]
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
for name, w, h, depth in cases:
    hs = rtc.HostScene.from_file(name); cam = hs.camera(w, h)
    canvas = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
    gpu = rtc.GpuScene(hs.desc)
    gpu.render_device(cam, canvas.data_ptr(), depth, None, stream.cuda_stream); torch.cuda.synchronize()
    est, got = gpu.chunk_times(cam)
    est, got = est.astype(np.float64), got.astype(np.float64)
    r = np.corrcoef(est, got)[0, 1]
    scale = got.sum() / est.sum()
    order = np.argsort(est)
    dec = np.array_split(order, 10)
    print(f"{name} {w}x{h}: {len(est)} chunks, estimate total {est.sum() / 1e6:.1f} M ticks, measured {got.sum() / 1e6:.1f} M (x{scale:.2f}), correlation {r:.3f}")
    print("  decile of estimate: mean estimate -> mean measured (ticks):", " | ".join(f"{est[d].mean():.0f} -> {got[d].mean():.0f}" for d in dec))
    top = np.argsort(got)[-len(got) // 100:]
    print(f"  the heaviest 1 % of chunks by measurement: measured mean {got[top].mean():.0f}, their estimate mean {est[top].mean():.0f}; "
          f"heaviest chunk measured {got.max():.0f}, its estimate {est[np.argmax(got)]:.0f}; largest estimate {est.max():.0f}")
    gpu.close()
