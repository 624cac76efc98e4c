#!/usr/bin/env python3
"""What the FIRST device-to-host copy of a process costs, apart from this library: plain hipMemcpy of a 1080p f64 canvas
(49.8 MB) from a torch device buffer into (a) a touched pageable buffer, first copy of the process, (b) the same again,
(c) a fresh untouched buffer, (d) the same again; then what touching a fresh buffer costs alone.

    python tools/first_copy_probe.py [warm_bytes]

With `warm_bytes` a copy of that many bytes (into its own touched buffer) is made and timed FIRST: does a small copy pay
the one-time set-up for the large one?"""
import ctypes, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
importlib.import_module("ray-tracer-challenge_amd").hip_lib()   # (maps the one HIP runtime of the process into the global scope)
hip = ctypes.CDLL(None)
n = 1920 * 1080 * 3
d = torch.zeros(n, dtype=torch.float64, device="cuda"); torch.cuda.synchronize()
def copy(dst):
    t0 = time.perf_counter()
    assert hip.hipMemcpy(ctypes.c_void_p(dst.ctypes.data), ctypes.c_void_p(d.data_ptr()), ctypes.c_size_t(dst.nbytes), 2) == 0
    return (time.perf_counter() - t0) * 1e3
if len(sys.argv) > 1:
    w = np.empty(max(1, int(sys.argv[1]) // 8)); w[:] = 1.0
    print("warm copy of %d bytes first: %.2f ms, again %.2f ms" % (w.nbytes, copy(w), copy(w)))
a = np.empty(n); a[:] = 1.0
print("touched buffer, first copy of the process %.2f ms, again %.2f ms" % (copy(a), copy(a)))
b = np.empty(n)
print("fresh buffer %.2f ms, again %.2f ms" % (copy(b), copy(b)))
c = np.empty(n); t0 = time.perf_counter(); c[::512] = 0.0
print("touching a fresh buffer alone %.2f ms" % ((time.perf_counter() - t0) * 1e3))
e = np.empty(n)
print("another fresh buffer %.2f ms" % copy(e))
