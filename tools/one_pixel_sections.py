#!/usr/bin/env python3
"""Where ONE wave's time goes when it renders one pixel (or a small rectangle) with the GPU to itself - the dependent
chain that floors every small launch.  Needs a -DRTC_PROFILE build (python tools/variants.py "prof=-DRTC_PROFILE" -- python tools/one_pixel_sections.py ...):
    python tools/one_pixel_sections.py cover.json 1920 1080 X Y W H [depth]"""
import importlib, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NAMES = ["pop/store", "closest", "after-closest", "shadow", "lighting", "behind", "after-behind/spawn", "(iterations)", "share", "deal",
         "record", "normal", "pattern", "spawn", "counter", "items"]
if os.environ.get("RTC_PROF_CHILD") != "1":
    out = subprocess.run([sys.executable] + sys.argv, env=dict(os.environ, RTC_PROF_CHILD="1", RTC_PROFILE_DUMP="1"), capture_output=True, text=True)
    text = out.stdout + out.stderr
    for l in text.splitlines():
        if l.startswith("rect") or l.startswith("stats"): print(l)
    last = [l for l in text.splitlines() if l.startswith("rtc prof:")]
    if last:
        v = [int(x) for x in last[-1].split("|")[0].split()[2:18]]
        total = sum(v[i] for i in range(16) if i != 7)
        print("wave ticks (100 MHz) in sections, all waves summed:", total, "iterations", v[7])
        print(" | ".join(f"{n} {v[i]} ({100.0 * v[i] / total:.0f}%)" for i, n in enumerate(NAMES) if i != 7 and v[i]))
        print(last[-1].split("|", 1)[1])
    else:
        print(text[-2000:])
    sys.exit(0)
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
scene, w, h, x, y, rw, rh = sys.argv[1], *[int(a) for a in sys.argv[2:8]]
depth = int(sys.argv[8]) if len(sys.argv) > 8 else 5
hs = rtc.HostScene.from_file(scene); cam = hs.camera(w, h)
g = rtc.GpuScene(hs.desc)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
buf = torch.empty((rh, rw, 3), dtype=torch.float64, device="cuda")
for i in range(4):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream); g.render_device(cam, buf.data_ptr(), depth, (x, y, rw, rh), stream.cuda_stream); b.record(stream); torch.cuda.synchronize()
    st = g.stats()
    print("rect", (x, y, rw, rh), "launch", i, "ms", round(a.elapsed_time(b), 4), flush=True)
print("stats", st)
