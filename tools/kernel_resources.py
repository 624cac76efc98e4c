#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every render kernel, from the compiler's own resource-usage remarks
(CPU only: cross-compiles rtc_kernels.hip for gfx950 into /tmp).  Extra hipcc flags may follow, e.g. -DRTC_LB2=1."""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, "ray-tracer-challenge_amd", "csrc", "rtc_kernels.hip")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fPIC",
       "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/tmp/rtc_kernels_res.o", src] + sys.argv[1:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: [^ ]*?:?\s*([A-Za-z][\w \[\]/]*?): (.+?) \[-Rpass", line)
    if not m:
        if "error" in line: print(line)
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name" or k == "Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k] = v
print(f"{'kernel':34s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s}")
for r in rows:
    if not r["name"].startswith("rtc_render"): continue
    print(f"{r['name']:34s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} {r.get('VGPRs Spill','?'):>6s} "
          f"{r.get('SGPRs Spill','?'):>6s} {r.get('ScratchSize [bytes/lane]','?'):>8s} {r.get('LDS Size [bytes/block]','?'):>7s} {r.get('Occupancy [waves/SIMD]','?'):>4s}")
