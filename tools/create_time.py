#!/usr/bin/env python3
"""Host-side cost of getting a scene onto the GPU: load (JSON + OBJ + PNG parse, flatten), rtc_scene_create (validation,
candidate BVH build, uploads), rtc_scene_clone, first frame.  python tools/create_time.py [scene ...] (GPU box)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
torch.cuda.init(); torch.zeros(1, device="cuda")
for name in sys.argv[1:] or ["cover.json", "teapot.json", "nefertiti.json", "dragons.json"]:
    t0 = time.perf_counter(); hs = rtc.HostScene.from_file(name); t1 = time.perf_counter()
    g = rtc.GpuScene(hs.desc); t2 = time.perf_counter()
    g2 = rtc.GpuScene(hs.desc); t3 = time.perf_counter()
    c = g.clone(); t4 = time.perf_counter()
    cam = hs.camera(); img = g.render(cam, 5); t5 = time.perf_counter()
    print(f"{name}: load {1e3*(t1-t0):.1f} ms, create {1e3*(t2-t1):.1f} ms (again {1e3*(t3-t2):.1f}), clone {1e3*(t4-t3):.2f} ms, first render to host {1e3*(t5-t4):.1f} ms; "
          f"{hs.desc.n_leaves} leaves", flush=True)
    g.close(); g2.close(); c.close()
