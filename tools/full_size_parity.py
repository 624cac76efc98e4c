#!/usr/bin/env python3
"""Every golden scene at 1920x1080 (dragons at 3840x2160), launches 1-3 and one after a camera move, every 60th row
against the oracle; prints the kernel, the frame time and the largest difference.  A one-off widening of
tests/test_parity_gpu.py::test_full_size_configs to all scenes."""
import glob, importlib, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
rtc = importlib.import_module("ray-tracer-challenge_amd")
import oracle_binding as ob
bad = 0
for path in sorted(glob.glob(os.path.join(REPO, "tests/golden/scenes/*.json"))):
    name = os.path.basename(path)
    w, h = (3840, 2160) if name == "dragons.json" else (1920, 1080)
    depth = 8 if name.startswith("reflection") else 5
    hs = rtc.HostScene.from_file(name); gpu = rtc.GpuScene(hs.desc); osc = ob.OracleScene(hs.desc)
    step = 60 if h == 1080 else 120
    worst = 0.0
    for launch in range(4):
        if launch == 3: hs.rotate_camera(0.2)
        cam = hs.camera(w, h)
        if launch in (0, 3): want, counters = osc.render(cam, depth, row_step=step)
        got = gpu.render(cam, depth); st = gpu.stats()
        rows = np.arange(0, h, step)
        d = float(np.abs(got[rows] - want[rows]).max())
        worst = max(worst, d)
        ok = d < 1e-9 and st["overflow"] == 0 and st["primary"] == w * h and np.isfinite(got).all()
        if not ok:
            bad += 1
            print("MISMATCH", name, launch, d, st)
    print(f"{name:34s} {gpu.last_kernel_name():30s} worst |delta| {worst:.2e}", flush=True)
    gpu.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
