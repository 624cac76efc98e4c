#!/usr/bin/env python3
"""GPU-only steady-state frame time of named scenes (kernel side, HIP events on the render's stream; no oracle).

    python tools/time_scenes.py [--set configs|mesh|misc|all] [--scenes a,b,..] [--size WxH] [--depth N]
                                [--handles 3] [--settle 50] [--frames 20] [--option name=value ...] [--check]

  configs  the five BASELINE configs at their own sizes and depths (fresnel 300x300, cover / teapot 1080p, r&r 1080p
           depth 8, dragons 4K)                                                       [default]
  mesh     teapot 1080p, dragons 4K, nefertiti 1080x1800 (the BVH walk)
  misc     the scenes beyond the configs at 1080p (csg, texture maps, cones, cylinders, perturbed patterns)
Each scene: `--handles` fresh scene handles (every handle measures its own first frame and packs its own schedule),
`--settle` untimed frames, then `--frames` timed back to back; prints min [mean max] over the handles and the kernel
that ran.  --check: also compares every 24th row with the CPU oracle (needs oracle/build/liboracle.so).
The library directory is the package's, or $RTC_LIB_DIR (tools/variants.py points it at a variant build)."""
import argparse, importlib, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")

SETS = {
    "configs": [("fresnel", 300, 300, 5), ("cover", 1920, 1080, 5), ("reflection_and_refraction", 1920, 1080, 8),
                ("teapot", 1920, 1080, 5), ("dragons", 3840, 2160, 5)],
    "mesh": [("teapot", 1920, 1080, 5), ("dragons", 3840, 2160, 5), ("nefertiti", 1080, 1800, 5)],
    "misc": [(n, 1920, 1080, 5) for n in ("csg_demo", "csg", "texture_demo", "earth", "skybox_demo", "nefertiti", "groups",
                                          "cubes", "cylinders", "xyz", "perturb_demo")],
}
SETS["all"] = SETS["configs"] + [c for c in SETS["misc"]]

ap = argparse.ArgumentParser()
ap.add_argument("--set", default="configs", choices=sorted(SETS))
ap.add_argument("--scenes", default="")
ap.add_argument("--size", default="")
ap.add_argument("--depth", type=int, default=0)
ap.add_argument("--handles", type=int, default=3)
ap.add_argument("--settle", type=int, default=50)
ap.add_argument("--frames", type=int, default=20)
ap.add_argument("--option", action="append", default=[])
ap.add_argument("--check", action="store_true")
ap.add_argument("--label", default="")
args = ap.parse_args()
cases = SETS[args.set]
if args.scenes:
    known = {c[0]: c for s in SETS.values() for c in s}
    cases = [known.get(n, (n, 1920, 1080, 5)) for n in args.scenes.split(",")]
if args.size:
    w, h = (int(v) for v in args.size.split("x"))
    cases = [(c[0], w, h, c[3]) for c in cases]
if args.depth:
    cases = [(c[0], c[1], c[2], args.depth) for c in cases]
for opt in args.option:
    n, v = opt.split("=")
    rtc.set_option(n, float(v))
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
out = []
for name, w, h, depth in cases:
    hs = rtc.HostScene.from_file(name + ".json"); cam = hs.camera(w, h)
    canvas = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
    ts, kernel, delta = [], "", None
    for rep in range(args.handles):
        gpu = rtc.GpuScene(hs.desc)
        for i in range(args.settle):
            gpu.render_device(cam, canvas.data_ptr(), depth, None, stream.cuda_stream)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(args.frames): gpu.render_device(cam, canvas.data_ptr(), depth, None, stream.cuda_stream)
        b.record(stream); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / args.frames)
        kernel = gpu.last_kernel_name()
        st = gpu.stats()
        if st["overflow"]: kernel += " OVERFLOW"
        if args.check and rep == 0:
            import numpy as np, oracle_binding as ob
            step = max(1, h // 24)
            want, c = ob.OracleScene(hs.desc).render(cam, depth, row_step=step, threads=os.cpu_count() and 16)
            rows = np.arange(0, h, step)
            delta = float(np.abs(canvas.cpu().numpy()[rows] - want[rows]).max())
        gpu.close()
    line = f"{name[:14]} {min(ts):.4f} [{sum(ts) / len(ts):.4f} {max(ts):.4f}] {kernel.replace('rtc_render_kernel', 'k')}"
    if delta is not None: line += f" maxdelta {delta:.2e}"
    out.append(line)
    print((args.label + " " if args.label else "") + line, flush=True)
