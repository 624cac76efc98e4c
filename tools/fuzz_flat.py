#!/usr/bin/env python3
"""The no-group random scenes of tests/test_parity_gpu.py (_random_flat_scene) over many seeds: GPU vs oracle,
two launches each (estimate-scheduled, device-packed).  tools/fuzz_flat.py [first] [count] [max_objects]
(rtc.set_option('simple3_min_chunks', 0) with max_objects 7 puts the simple scenes on the three-wave kernel)"""
import importlib, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
rtc = importlib.import_module("ray-tracer-challenge_amd")
import oracle_binding as ob
import test_parity_gpu as t
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
max_objects = int(sys.argv[3]) if len(sys.argv) > 3 else 60
for opt in sys.argv[4:]:   # name=value options, e.g. box_cull=1 simple3_min_chunks=0
    rtc.set_option(opt.split("=")[0], float(opt.split("=")[1]))
kernels = {}
bad, worst = [], 0.0
for seed in range(first, first + count):
    for simple in (True, False):
        hs = rtc.HostScene(t._random_flat_scene(seed, simple, max_objects))
        cam = hs.camera()
        gpu = rtc.GpuScene(hs.desc)
        want, counters = ob.OracleScene(hs.desc).render(cam, 5)
        for launch in range(2):
            got = gpu.render(cam, 5)
            d = float(np.abs(got - want).max())
            st = gpu.stats()
            kernels[gpu.last_kernel_name()] = kernels.get(gpu.last_kernel_name(), 0) + 1
            ok = d < 1e-5 and st["secondary"] == counters["secondary"] and st["shadow_calls"] == counters["shadow"] and st["overflow"] == 0
            worst = max(worst, d)
            if not ok:
                bad.append((seed, simple, launch, d))
        gpu.close()
    if (seed - first) % 50 == 49:
        print("...", seed, "worst so far", worst, flush=True)
print(kernels)
print(f"{count} seeds from {first} (simple and flat): {len(bad)} mismatches {bad[:5]}, worst |delta| {worst:.3e}")
sys.exit(1 if bad else 0)
