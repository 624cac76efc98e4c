#!/usr/bin/env python3
"""Copies the summaries of one tools/profile_round.sh run (gpurun_out/profiles_<tag>/) into profiles/<round>/ and
refreshes profiles/traffic.json (what bench.py reports as roofline.traffic).

    python tools/collect_profiles.py gpurun_out/profiles_r01d r01
"""
import glob, json, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarize

src, rnd = sys.argv[1], sys.argv[2]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(repo, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats, os.path.join(dst, "kernel_stats.csv"))
trace = glob.glob(os.path.join(src, "stats", "**", "*_kernel_trace.csv"), recursive=True)[0]
with open(trace) as f, open(os.path.join(dst, "kernel_trace_head.csv"), "w") as g:
    rows = f.readlines()
    g.writelines([rows[0]] + [r for r in rows[1:] if "rtc_render_kernel" in r][:3])
pmc = summarize([os.path.join(src, d) for d in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2", "pmc_tcc")])
with open(os.path.join(dst, "pmc_summary.txt"), "w") as g:
    g.write("# per-launch means over the dispatches of rtc_render_kernel; separate rocprofv3 --pmc passes (tools/profile_round.sh)\n")
    for k, v in sorted(pmc.items()):
        g.write(f"{k:32s} {v:18.1f}\n")
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "bench_n1.json"))
with open(os.path.join(src, "configs.txt")) as f, open(os.path.join(dst, "configs_all.jsonl"), "w") as g:
    g.writelines(l for l in f if l.startswith("{"))
if os.path.exists(os.path.join(src, "scale_sim.txt")):
    shutil.copy(os.path.join(src, "scale_sim.txt"), os.path.join(dst, "scale_sim_one_gpu.jsonl"))
dstats = glob.glob(os.path.join(src, "d_stats", "**", "*_kernel_stats.csv"), recursive=True)
if dstats:
    shutil.copy(dstats[0], os.path.join(dst, "kernel_stats_dragons.csv"))
    dpmc = summarize([os.path.join(src, d) for d in ("d_pmc_fetch", "d_pmc_write", "d_pmc_sq1", "d_pmc_tcc")])
    with open(os.path.join(dst, "pmc_dragons.txt"), "w") as g:
        g.write("# dragons.json 3840x2160 depth 5 (rtc_render_kernel): per-launch means, separate rocprofv3 --pmc passes (tools/profile_round.sh)\n")
        for k, v in sorted(dpmc.items()):
            g.write(f"{k:32s} {v:18.1f}\n")
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
traffic = {"scene": "cover.json", "width": 1920, "height": 1080, "depth": 5,
           "fetch_size_kb": round(pmc["FETCH_SIZE"], 1), "write_size_kb": round(pmc["WRITE_SIZE"], 1),
           "pmc": {k: round(v, 1) for k, v in sorted(pmc.items())},
           "source": f"profiles/{rnd}/pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, per-launch mean)"}
json.dump(traffic, open(os.path.join(repo, "profiles", "traffic.json"), "w"), indent=1)
ks = open(os.path.join(dst, "kernel_stats.csv")).read().splitlines()
row = [r for r in ks if "rtc_render_kernel" in r][0].replace('"', "").split(",")
print("kernel avg ns", row[3], "calls", row[1], "| bench kernel_ms", bench["roofline"]["kernel_ms"], "ms_per_step", bench["ms_per_step"])
print({k: pmc[k] for k in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY") if k in pmc})
