#!/usr/bin/env python3
"""Where a wave's time goes, and what the group walks did (needs a -DRTC_PROFILE build: python tools/variants.py "prof=-DRTC_PROFILE" -- python tools/prof_sections.py ...):
prof_sections.py dragons.json 3840 2160 [depth].  Shares only - the stamps slow the kernel down."""
import importlib, os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("RTC_PROF_CHILD") != "1":
    env = dict(os.environ, RTC_PROF_CHILD="1", RTC_PROFILE_DUMP="1")
    out = subprocess.run([sys.executable] + sys.argv, env=env, capture_output=True, text=True)
    text = out.stdout + out.stderr
    names = ["pop/store", "closest", "after-closest", "shadow", "lighting", "behind", "after-behind/spawn", "(iterations)", "share", "deal",
             "record", "normal", "pattern", "spawn", "counter", "items"]
    last = [l for l in text.splitlines() if l.startswith("rtc prof:")]
    for l in text.splitlines():
        if l.startswith("frame") or l.startswith("rtc walks") or l.startswith("rtc trace") or l.startswith("rtc occluder"): print(l)
    if last:
        v = [int(x) for x in last[-1].split("|")[0].split()[2:18]]
        total = sum(v[i] for i in range(16) if i != 7)
        print(" | ".join(f"{n} {100.0 * v[i] / total:.1f}%" for i, n in enumerate(names) if i != 7 and v[i]))
        print(f"total wave cycles {total} ({total / 1e6:.0f} M), wave-level iterations {v[7]}")
    else:
        print(text[-3000:])
    sys.exit(0)
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
scene, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 5
hs = rtc.HostScene.from_file(scene); cam = hs.camera(w, h)
g = rtc.GpuScene(hs.desc)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
canvas = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
for i in range(4):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream); g.render_device(cam, canvas.data_ptr(), depth, None, stream.cuda_stream); b.record(stream); torch.cuda.synchronize()
    print("frame", i, "ms", round(a.elapsed_time(b), 3), flush=True)
    st = g.stats()
print("stats", st)
