#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every render kernel, from the compiler's own resource-usage remarks.

    python tools/kernel_resources.py [-DFLAGS ...]                     cross-compiles rtc_kernels.hip for gfx950 into /tmp (CPU only) and prints the table
    python tools/kernel_resources.py --from-remarks FILE --json OUT    the Makefile's form: the remarks of the PRODUCT build (hipcc -Rpass-analysis=
                                                                       kernel-resource-usage ... 2> FILE) as lib/kernel_resources.json, which bench.py
                                                                       reads for `roofline.scratch_bytes_per_lane` of the kernel that ran"""
import json, os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(remarks):
    rows, cur = [], None
    for line in remarks.splitlines():
        m = re.search(r"remark: [^ ]*?:?\s*([A-Za-z][\w \[\]/]*?): (.+?) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name" or k == "Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    return rows


def as_json(rows):
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "TotalSGPRs": "sgprs", "VGPRs Spill": "vgprs_spilled", "SGPRs Spill": "sgprs_spilled",
            "ScratchSize [bytes/lane]": "scratch_bytes_per_lane", "LDS Size [bytes/block]": "lds_bytes_per_block",
            "Occupancy [waves/SIMD]": "waves_per_simd"}
    return {r["name"]: {keys[k]: int(v) for k, v in r.items() if k in keys and v.isdigit()} for r in rows}


if __name__ == "__main__":
    argv = sys.argv[1:]
    if "--from-remarks" in argv:
        rows = parse(open(argv[argv.index("--from-remarks") + 1]).read())
        out = argv[argv.index("--json") + 1]
        json.dump(as_json(rows), open(out, "w"), indent=1, sort_keys=True)
        sys.exit(0 if rows else 1)
    src = os.path.join(REPO, "ray-tracer-challenge_amd", "csrc", "rtc_kernels.hip")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fPIC",
           "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/tmp/rtc_kernels_res.o", src] + argv
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    for line in err.splitlines():
        if "error" in line: print(line)
    rows = parse(err)
    print(f"{'kernel':34s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s}")
    for r in rows:
        if not r["name"].startswith("rtc_render"): continue
        print(f"{r['name']:34s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} {r.get('VGPRs Spill','?'):>6s} "
              f"{r.get('SGPRs Spill','?'):>6s} {r.get('ScratchSize [bytes/lane]','?'):>8s} {r.get('LDS Size [bytes/block]','?'):>7s} {r.get('Occupancy [waves/SIMD]','?'):>4s}")
