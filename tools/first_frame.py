#!/usr/bin/env python3
"""One-shot use (the reference CLI renders every scene once, main.zig:52-99): GPU and wall time of the FIRST launch on a
fresh scene handle (schedule from rtc_estimate_kernel; measures), of the second (schedule packed from the first's
measurements; a static view measures nothing more) and the median of launches 4-11, per BASELINE config; three handles,
the median handle per launch.  Options as name=value (e.g. cut_above=0.5)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
label = ""
for opt in sys.argv[1:]:
    n, v = opt.split("="); rtc.set_option(n, float(v)); label += opt + " "
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
for name, w, h, depth in (("cover.json", 1920, 1080, 5), ("reflection_and_refraction.json", 1920, 1080, 8), ("teapot.json", 1920, 1080, 5),
                          ("dragons.json", 3840, 2160, 5), ("nefertiti.json", 1080, 1800, 5), ("fresnel.json", 300, 300, 5)):
    hs = rtc.HostScene.from_file(name); cam = hs.camera(w, h)
    canvas = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
    res = []
    for rep in range(3):
        gpu = rtc.GpuScene(hs.desc)
        torch.cuda.synchronize()
        row = []
        for i in range(12):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            a.record(stream)
            gpu.render_device(cam, canvas.data_ptr(), depth, None, stream.cuda_stream)
            b.record(stream); torch.cuda.synchronize()
            row.append((a.elapsed_time(b), (time.perf_counter() - t0) * 1e3))
        res.append(row)
        gpu.close()
    med = [sorted(r[i] for r in res)[len(res) // 2] for i in range(12)]   # per launch: the median handle
    steady = sorted(m[0] for m in med[4:])[4]
    print(f"{label}{name.split('.')[0][:12]:12s} first gpu {med[0][0]:.3f} wall {med[0][1]:.3f} | second gpu {med[1][0]:.3f} | steady gpu {steady:.3f} | first / steady {med[0][0] / steady:.2f}", flush=True)
