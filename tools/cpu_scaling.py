#!/usr/bin/env python3
"""Host CPU environment of the GPU box and the oracle's thread-scaling curve on one fixed sample of rows."""
import importlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
rtc = importlib.import_module("ray-tracer-challenge_amd")
import oracle_binding as ob
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(f, open(f).read().strip())
    except OSError:
        pass
os.system("lscpu | grep -E 'Model name|Socket|Thread|Core|NUMA node\\(s\\)'")
hs = rtc.HostScene.from_file("cover.json")
cam = hs.camera(1920, 1080)
osc = ob.OracleScene(hs.desc)
for th in (1, 2, 4, 8, 16, 32, 48, 64, 96, 128, 192, 256):
    best = None
    for rep in range(2):
        t0 = time.perf_counter(); _, c = osc.render(cam, 5, row_step=8, threads=th); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    r = (c["primary"] + c["secondary"]) / best / 1e6
    print(f"{th:4d} threads {best:7.3f} s {r:8.3f} Mrays/s {r / th:6.3f} per thread", flush=True)
