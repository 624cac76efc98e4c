#!/usr/bin/env python3
"""Two- against three-waves-per-SIMD form of the simple kernel over image sizes (the crossover behind usesSimple3() in
csrc/rtc_capi.hip): python tools/simple3_sweep.py [scene ...]; steady-state ms per frame, best of two handles."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
SIZES = [(300, 300), (640, 360), (960, 540), (1280, 720), (1600, 900), (1920, 1080), (2560, 1440), (3840, 2160)]
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)


def frame_ms(hs, w, h, depth):
    cam = hs.camera(w, h)
    canvas = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
    best = 1e9
    for _ in range(2):
        gpu = rtc.GpuScene(hs.desc)
        for i in range(6):
            gpu.render_device(cam, canvas.data_ptr(), depth, None, stream.cuda_stream); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(12): gpu.render_device(cam, canvas.data_ptr(), depth, None, stream.cuda_stream)
        b.record(stream); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 12)
        gpu.close()
    return best


for name in sys.argv[1:] or ["cover", "reflection_and_refraction", "fresnel", "cubes"]:
    hs = rtc.HostScene.from_file(name + ".json")
    depth = 8 if name.startswith("reflection") else 5
    for w, h in SIZES:
        rtc.set_option("simple3_min_chunks", 1e9)
        t2 = frame_ms(hs, w, h, depth)
        rtc.set_option("simple3_min_chunks", 0)
        t3 = frame_ms(hs, w, h, depth)
        print(f"{name[:10]:10s} {w}x{h} chunks {((w + 7) // 8) * ((h + 7) // 8):6d}  two waves {t2:.3f}  three waves {t3:.3f}  ratio {t3 / t2:.3f}", flush=True)
