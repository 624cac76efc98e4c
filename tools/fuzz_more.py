#!/usr/bin/env python3
"""One-off widening of tests/test_parity_gpu.py::test_random_scenes: the same generator over many more seeds
(python tools/fuzz_more.py [first] [count] [depth] [name=value option ...], e.g. waves3=1 for the three-wave general
kernel); prints the seeds that break parity or the ray counters."""
import importlib, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import torch
torch.zeros(1, device="cuda")  # torch's HIP runtime first (see _one_hip_runtime)
rtc = importlib.import_module("ray-tracer-challenge_amd")
import oracle_binding as ob
from test_parity_gpu import _random_scene, TOL
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 5
for opt in sys.argv[4:]:
    rtc.set_option(opt.split("=")[0], float(opt.split("=")[1]))
bad, worst = [], 0.0
for seed in range(first, first + count):
    hs = rtc.HostScene(_random_scene(seed)); cam = hs.camera()
    gpu = rtc.GpuScene(hs.desc)
    got = gpu.render(cam, depth)
    want, counters = ob.OracleScene(hs.desc).render(cam, depth)
    st = gpu.stats()
    d = float(np.abs(got - want).max())
    worst = max(worst, d)
    if not (d < TOL and st["secondary"] == counters["secondary"] and st["shadow_calls"] == counters["shadow"] and st["overflow"] == 0):
        bad.append((seed, d))
        print("MISMATCH seed", seed, d, flush=True)
    gpu.close()
    if (seed - first) % 50 == 49: print("...", seed, "worst so far", worst, flush=True)
print(f"{count} seeds from {first}: {len(bad)} mismatches, worst |delta| {worst:.3e}")
sys.exit(1 if bad else 0)
