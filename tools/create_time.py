#!/usr/bin/env python3
"""What rtc_scene_create costs for the mesh scenes: the host half by itself (rtc_diag_build_tables: validation,
depth-first order, bounds, the candidate BVHs on `build_threads` threads, the eight-wide collapse) and the whole create
(+ leaf records, upload), single-threaded and with the library's own thread count; before them what the host library's
loader takes to get from the scene file to the description (librtc_host: the entries of "objects" one after the other,
and on the loader's own thread count).  python tools/create_time.py [scene ...]"""
import importlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
rtc = importlib.import_module("ray-tracer-challenge_amd")
scenes = sys.argv[1:] or ["dragons.json", "nefertiti.json", "teapot.json", "cover.json"]
warm = rtc.GpuScene(rtc.HostScene.from_file("fresnel.json").desc)  # (the process's first create carries the HIP runtime's start-up)
for scene in scenes:
    for loader_threads in (1, 0):
        rtc.set_loader_threads(loader_threads)
        loads = []
        for _ in range(3):
            t = time.perf_counter()
            hs = rtc.HostScene.from_file(scene)
            loads.append((time.perf_counter() - t) * 1e3)
        print(f"{scene:16s} HostScene.from_file (JSON + OBJ -> World -> divide(8) -> rtc_scene_desc), loader threads {loader_threads or 'default':>7}: "
              f"{min(loads):7.1f} ms (min of 3; first {loads[0]:.1f})", flush=True)
    for threads in (1, 0):
        rtc.set_option("build_threads", threads)
        tables = min(rtc.build_tables_digest(hs.desc)[1] for _ in range(5))
        creates = []
        for _ in range(5):
            t = time.perf_counter()
            g = rtc.GpuScene(hs.desc)
            creates.append((time.perf_counter() - t) * 1e3)
            del g
        print(f"{scene:16s} build_threads {threads or 'default':>7}: tables {tables:7.1f} ms, rtc_scene_create {min(creates):7.1f} ms (min of 5)", flush=True)
rtc.set_option("build_threads", 0)
