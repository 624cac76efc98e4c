#!/usr/bin/env python3
"""Copies the summaries of one tools/profile_round.sh run (gpurun_out/profiles_<tag>/) into profiles/<round>/ and
refreshes profiles/traffic.json (what bench.py reports as roofline.traffic / roofline_valu.executed_*), one entry per
profiled workload.  Run again after tools/profile_round_bench.sh to pick the bench lines up.

    python tools/collect_profiles.py gpurun_out/profiles_r03 r03
"""
import glob, json, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarize

src, rnd = sys.argv[1], sys.argv[2]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(repo, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
WORKLOADS = [("", "cover", {"scene": "cover.json", "width": 1920, "height": 1080, "depth": 5}),
             ("d_", "dragons", {"scene": "dragons.json", "width": 3840, "height": 2160, "depth": 5}),
             ("t_", "teapot", {"scene": "teapot.json", "width": 1920, "height": 1080, "depth": 5})]
PASSES = ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2", "pmc_sq3", "pmc_tcc")
entries = []
for prefix, name, key in WORKLOADS:
    stats = glob.glob(os.path.join(src, prefix + "stats", "**", "*_kernel_stats.csv"), recursive=True)
    if not stats:
        continue
    shutil.copy(stats[0], os.path.join(dst, f"kernel_stats_{name}.csv"))
    trace = glob.glob(os.path.join(src, prefix + "stats", "**", "*_kernel_trace.csv"), recursive=True)[0]
    with open(trace) as f, open(os.path.join(dst, f"kernel_trace_head_{name}.csv"), "w") as g:
        rows = f.readlines()
        g.writelines([rows[0]] + [r for r in rows[1:] if "rtc_render_kernel" in r][-3:])
    # the render kernel's launch durations from the trace: all dispatches (what --stats averages: it includes the one
    # no-work dispatch a handle's first launch sends ahead, and the frames of the two- / three-wave trial) and the steady
    # state (the kernel with the most dispatches, durations within a factor of two of their median)
    import csv, statistics
    durs = {}
    for r in csv.DictReader(open(trace)):
        if "rtc_render_kernel" in r["Kernel_Name"]:
            durs.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    main = max(durs, key=lambda n: len(durs[n]))
    med = statistics.median(durs[main])
    steady = [d for d in durs[main] if 0.5 * med <= d <= 2.0 * med]
    key["kernel_ms"] = {"kernel": main, "steady_mean": sum(steady) / len(steady), "steady_dispatches": len(steady),
                        "all_mean": sum(durs[main]) / len(durs[main]), "all_dispatches": len(durs[main]),
                        "other_kernels": {n: len(v) for n, v in durs.items() if n != main}}
    rep = {}
    pmc = summarize([os.path.join(src, prefix + d) for d in PASSES], report=rep)
    with open(os.path.join(dst, f"pmc_{name}.txt"), "w") as g:
        g.write(f"# {key['scene']} {key['width']}x{key['height']} depth {key['depth']}: per-launch means over the steady-state dispatches of {rep.get('kernel')} (tools/pmc_summary.py); separate rocprofv3 --pmc passes (tools/profile_round.sh)\n")
        for k, v in sorted(pmc.items()):
            g.write(f"{k:32s} {v:18.1f}\n")
        valu = pmc.get("SQ_INSTS_VALU", 0.0)
        if valu and "SQ_BUSY_CYCLES" in pmc:
            cycles = pmc["SQ_BUSY_CYCLES"] / 32.0
            g.write(f"# lanes active per VALU instruction      {pmc['SQ_THREAD_CYCLES_VALU'] / 64.0 / pmc['SQ_ACTIVE_INST_VALU']:.3f}\n")
            g.write(f"# wave cycles waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES) {pmc['SQ_WAIT_ANY'] / pmc['SQ_WAVE_CYCLES']:.3f}\n")
            g.write(f"# vector pipe issuing (SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / (SQ_BUSY_CYCLES / 32 SEs)) {pmc['SQ_ACTIVE_INST_VALU'] * 4.0 / 1024.0 / cycles:.3f}\n")
            classes = ["ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "INT64", "CVT"]
            known = sum(pmc.get("SQ_INSTS_VALU_" + c, 0.0) for c in classes)
            g.write(f"# VALU instructions in no arithmetic class (moves, selects, compares, bit ops, div_scale / div_fixup ...) {(valu - known) / 1e6:.1f} M of {valu / 1e6:.1f} M\n")
            g.write(f"# HBM traffic FETCH_SIZE x 2 + WRITE_SIZE = {(2 * pmc['FETCH_SIZE'] + pmc['WRITE_SIZE']) * 1024 / 1e6:.1f} MB; canvas {24 * key['width'] * key['height'] / 1e6:.1f} MB\n")
    entries.append(dict(key, fetch_size_kb=round(pmc["FETCH_SIZE"], 1), write_size_kb=round(pmc["WRITE_SIZE"], 1),
                        pmc={k: round(v, 1) for k, v in sorted(pmc.items())},
                        source=f"profiles/{rnd}/pmc_{name}.txt (rocprofv3 --pmc, separate passes, per-launch mean)"))
    ks = open(os.path.join(dst, f"kernel_stats_{name}.csv")).read().splitlines()
    row = [r for r in ks if key["kernel_ms"]["kernel"] + '"' in r or key["kernel_ms"]["kernel"] + "," in r][0].replace('"', "").split(",")
    print(name, "kernel", row[0], "avg ns", row[3], "calls", row[1], "| VALU", round(pmc.get("SQ_INSTS_VALU", 0) / 1e6, 1), "M",
          "FETCH", round(pmc["FETCH_SIZE"] / 1024, 1), "MiB WRITE", round(pmc["WRITE_SIZE"] / 1024, 1), "MiB")
json.dump({"workloads": entries}, open(os.path.join(repo, "profiles", "traffic.json"), "w"), indent=1)
for name in ("bench.json", "bench_dragons.json", "bench_teapot.json", "bench_rr.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, "bench_n1.json" if name == "bench.json" else name))
if os.path.exists(os.path.join(src, "configs.txt")):
    with open(os.path.join(src, "configs.txt")) as f, open(os.path.join(dst, "configs_all.jsonl"), "w") as g:
        g.writelines(l for l in f if l.startswith("{"))
if os.path.exists(os.path.join(src, "scale_sim.txt")):
    shutil.copy(os.path.join(src, "scale_sim.txt"), os.path.join(dst, "scale_sim_one_gpu.jsonl"))
if os.path.exists(os.path.join(src, "scale_sim_inflight3.txt")):
    shutil.copy(os.path.join(src, "scale_sim_inflight3.txt"), os.path.join(dst, "scale_sim_frames_in_flight.jsonl"))
