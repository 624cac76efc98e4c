#!/usr/bin/env python3
"""Builds variants of librtc_hip.so with extra compiler flags and runs a command against each (GPU box).

    python tools/variants.py [--keep] "name=-DFLAG -DOTHER=2" "base=" ... -- <command ...> [-- <another command ...>]
    e.g.  python tools/variants.py "base=" "nofuse=-DRTC_FUSED_SHADOWS=0" -- python tools/time_scenes.py --set configs

Every variant is built into gpurun_out/variants/<name>/ (rtc_kernels.o, rtc_capi.o, librtc_hip.so, with copies of the host and
multi libraries beside it) - never into ray-tracer-challenge_amd/lib/, so the product build the tests and bench.py load stays
what `make` made.  The command runs with RTC_LIB_DIR pointing at the variant (the Python binding honours it) and its
output is prefixed with the variant's name.  The compiler's warnings and errors are shown.  Replaces the
*_variants.sh / try_*.sh family of rounds 2-3."""
import os, shutil, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "ray-tracer-challenge_amd")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
BASE = ["--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-result"]

argv = sys.argv[1:]
keep = "--keep" in argv
argv = [a for a in argv if a != "--keep"]
if "--" not in argv:
    sys.exit(__doc__)
cut = argv.index("--")
variants, rest = argv[:cut], argv[cut + 1:]
commands = [[]]
for a in rest:
    if a == "--":
        commands.append([])
    else:
        commands[-1].append(a)
rc = 0
for v in variants:
    name, _, flags = v.partition("=")
    out = os.path.join(REPO, "gpurun_out", "variants", name)
    os.makedirs(out, exist_ok=True)
    objs = []
    for f in ("rtc_kernels", "rtc_capi"):
        o = os.path.join(out, f + ".o")
        subprocess.check_call([HIPCC] + BASE + flags.split() + ["-c", "-o", o, os.path.join(PKG, "csrc", f + ".hip")], cwd=REPO)
        objs.append(o)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "librtc_hip.so")] + objs)
    for lib in ("librtc_host.so", "librtc_multi.so"):
        src = os.path.join(PKG, "lib", lib)
        if os.path.exists(src): shutil.copy(src, out)
    env = dict(os.environ, RTC_LIB_DIR=out)
    for command in commands:
        p = subprocess.run(command, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        for line in p.stdout.splitlines():
            print(f"[{name}] {line}", flush=True)
        rc = rc or p.returncode
    if not keep:
        for f in os.listdir(out):
            if f.endswith(".o"): os.remove(os.path.join(out, f))
sys.exit(rc)
