#!/usr/bin/env python3
"""Leave-one-out over the top-level objects of a random fuzz scene (tests/test_parity_gpu.py::_random_scene):
which objects a GPU / oracle mismatch needs.  python tools/fuzz_leave_one_out.py <seed>"""
import importlib, os, sys, json, copy
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
rtc = importlib.import_module("ray-tracer-challenge_amd")
import oracle_binding as ob
import test_parity_gpu as t
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 410
base = json.loads(t._random_scene(seed))
def kind(o):
    ty = o["type"]; return list(ty.keys())[0] if isinstance(ty, dict) else ty
def run(name, sc, depth=5):
    hs = rtc.HostScene(json.dumps(sc)); cam = hs.camera(); gpu = rtc.GpuScene(hs.desc)
    got = gpu.render(cam, depth); want, c = ob.OracleScene(hs.desc).render(cam, depth)
    d = np.abs(got - want).max(axis=2); bad = np.argwhere(d > 1e-5); st = gpu.stats()
    print(f"{name:28s} depth {depth}: bad {len(bad)} max {d.max():.3e} sec {st['secondary']}/{c['secondary']} first bad {bad[:2].tolist()}", flush=True)
for depth in (2, 3, 4, 5): run("base", base, depth)
objs = base["objects"]
for i, o in enumerate(objs):
    sc = copy.deepcopy(base); del sc["objects"][i]
    run(f"without #{i} {kind(o)}", sc)
