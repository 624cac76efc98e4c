#!/usr/bin/env python3
"""GPU-only frame time at 1920x1080 of the scenes beyond the five BASELINE configs (csg / texture-map `_ext` kernels,
perturbed patterns, cones, cylinders): steady state, three handles each."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
names = sys.argv[1:] or ["csg_demo", "csg", "texture_demo", "earth", "skybox_demo", "nefertiti", "groups", "cubes", "cylinders", "xyz", "perturb_demo"]
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
out = []
for name in names:
    hs = rtc.HostScene.from_file(name + ".json"); cam = hs.camera(1920, 1080)
    canvas = torch.empty((1080, 1920, 3), dtype=torch.float64, device="cuda")
    ts = []
    for rep in range(2):
        gpu = rtc.GpuScene(hs.desc)
        for i in range(6):
            gpu.render_device(cam, canvas.data_ptr(), 5, None, stream.cuda_stream); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(8): gpu.render_device(cam, canvas.data_ptr(), 5, None, stream.cuda_stream)
        b.record(stream); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 8)
        gpu.close()
    out.append(f"{name[:12]} {min(ts):.3f}")
print(" | ".join(out), flush=True)
