#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X render path.

Metric (BASELINE.json): Mrays/s (primary+secondary) and ms/frame on scenes/cover.json at
1920x1080, recursion depth 5.  One "step" = one frame: Camera.render of the whole image with the
flat scene already resident in HBM and the canvas left in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scene cover.json] [--width 1920] [--height 1080]

N == 1: one rtc_render_device launch per step.
N  > 1: (launched by torch.distributed.run, one rank per GPU, backend nccl == RCCL)  the image is cut
        into 64x64 tiles, dealt to the ranks by the tiles' MEASURED cost (one round-robin frame measures
        them during setup); every rank renders its tiles into a compact buffer, ONE gather per frame brings
        them to rank 0 over xGMI, rank 0 un-permutes them into the row-major canvas.  The gather of frame i runs on a
        side stream under the render of frame i+1, and a rank keeps THREE frames in flight (--inflight: three scene
        handles, streams and tile buffers, frames dealt round-robin): an eighth of a 0.5 ms frame is a handful of
        dependent iterations per wave, too short to fill a GPU by itself - the next frame's work-groups start on the
        CUs the last one has left (one-GPU rehearsal, tools/scale_sim.py --inflight: the slowest 8-way share of
        dragons 4K 0.66 -> 0.31 ms per frame, teapot 0.18 -> 0.06, cover 0.18 -> 0.11).  Total work per step is
        fixed -> "scaling": "strong".  config.frames_in_flight says what a line was measured with; the one-GPU
        headline is 1 (one frame after the other), its figure with 2 and 3 is in frames_in_flight_ms_per_frame.

Rank 0 prints ONE JSON line (see the keys at the bottom).  `roofline` describes the dominant (only)
kernel, rtc_render_kernel; `cpu_baseline` times the CPU oracle (a C++ restatement of the reference's
CPU path: the Zig reference itself cannot be built here) on a bounded sample of the same frame.
"""
import argparse
import importlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TF = 78.6   # half the 157.3 TF FP32 vector peak of the same guide (AMD spec sheet value)
TILE = 64


def algorithmic_bytes(desc, width, height):
    """Compulsory HBM bytes of one frame: the W*H*3 f64 canvas written once plus every scene table read
    once (DESIGN.md "Algorithmic bytes").  Everything else the reference's traversal touches per ray is
    re-use of those tables: for cover.json 19 leaves x 128 B per ray out of a ~10 KB scene that the
    kernel keeps in LDS."""
    scene = (desc.n_xforms * 96 + desc.n_leaves * 16 + desc.n_tris * 144 + desc.n_nodes * 56 + desc.n_children * 4 +
             desc.n_materials * 64 + desc.n_patterns * 144 + desc.n_lights * 48 + desc.n_roots * 176)
    return 24 * width * height + scene


def scene_bytes_touched(desc, stats):
    """SURVEY 8(d) per-ray figure for scenes without groups: every ray (incl. every isShadowed ray) of the
    REFERENCE traversal reads every leaf's 128-B inverse.  Cache-level re-use, not HBM traffic."""
    rays = stats["primary"] + stats["secondary"] + stats["shadow_calls"]
    return 128 * desc.n_leaves * rays if desc.n_nodes == 0 else None


def _profiled(scene, width, height, depth):
    """The committed PMC passes of this workload, if there are any (profiles/traffic.json, written by
    tools/collect_profiles.py: one entry per profiled workload)."""
    try:
        t = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))
    except OSError:
        return None
    for w in t.get("workloads", [t]):
        if [w.get("scene"), w.get("width"), w.get("height"), w.get("depth")] == [scene, width, height, depth]:
            return w
    return None


def measured_traffic(scene, width, height, depth):
    """HBM bytes per launch from the PMC passes of the latest committed profile (FETCH_SIZE x2 as the
    MI355X guide prescribes for gfx950, WRITE_SIZE exact), if that profile is of this workload."""
    t = _profiled(scene, width, height, depth)
    return None if t is None else 2 * 1024 * t["fetch_size_kb"] + 1024 * t["write_size_kb"]


def executed_flops(scene, width, height, depth):
    """FP64 flops the kernel EXECUTES per launch, from the committed SQ PMC pass of this workload (profiles/traffic.json,
    "pmc"): wave-level instruction counts x 64 lanes x the mean fraction of active lanes, an FMA counted as 2."""
    t = _profiled(scene, width, height, depth)
    c = (t or {}).get("pmc") or {}
    need = ["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64",
            "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU"]
    if any(k not in c for k in need):
        return None
    lanes = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
    wave_instr = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + 2 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_TRANS_F64"]
    out = {"flops": wave_instr * 64.0 * lanes, "lanes_active": lanes, "valu_instructions": c["SQ_INSTS_VALU"],
           "fp64_instructions": c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_TRANS_F64"],
           "source": t.get("source")}
    if "SQ_BUSY_CYCLES" in c and "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
        # a SIMD's vector pipe issuing, with the profile's OWN clock: SQ_BUSY_CYCLES counts per shader engine (32 of them)
        cycles = c["SQ_BUSY_CYCLES"] / 32.0
        out["valu_pipe_busy"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cycles
        out["wave_cycles_waiting"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        out["clock_source"] = "SQ_BUSY_CYCLES / 32 shader engines (the profiled launch's own cycles, no assumed clock)"
    return out


def kernel_resources(kernel_name):
    """Registers, spills, scratch bytes per lane and LDS of the kernel that ran, from the compiler's resource-usage remarks
    of the build that is loaded (lib/kernel_resources.json, written by the Makefile beside librtc_hip.so)."""
    try:
        lib_dir = os.environ.get("RTC_LIB_DIR", os.path.join(REPO, "ray-tracer-challenge_amd", "lib"))
        return json.load(open(os.path.join(lib_dir, "kernel_resources.json"))).get(kernel_name)
    except (OSError, ValueError):
        return None


def what_binds(ex, res):
    """The keys VERDICT r04 item 5 asks for INSIDE `roofline` (the driver keeps that object): what binds the kernel - not
    HBM - by the committed counters (`ex`: executed_flops()) and the compiler's figures for the kernel that ran (`res`)."""
    out = {"binding": "valu_issue"}
    if ex is not None:
        out["lanes_active"] = ex["lanes_active"]
        if "valu_pipe_busy" in ex:
            out["valu_issuing"] = ex["valu_pipe_busy"]
            out["waves_waiting"] = ex["wave_cycles_waiting"]
            # issue-bound when a SIMD's vector pipe issues in most of the kernel's cycles; else the waves wait (dependent
            # fetches of the BVH walk, spill reloads) more than they issue
            out["binding"] = "valu_issue" if ex["valu_pipe_busy"] >= 0.75 else "valu_issue+latency"
        out["counters_source"] = ex.get("source")
    if res is not None:
        out["scratch_bytes_per_lane"] = res.get("scratch_bytes_per_lane")
        out["vgprs_spilled"] = res.get("vgprs_spilled")
        out["waves_per_simd"] = res.get("waves_per_simd")
    return out


def reference_traversal_bytes(width, height, counters, rows_sampled, rows_total):
    """24 W H + sum over rays [56 bbox tests + 72 triangle tests + 72 smooth-triangle hits + 128 transforms
    applied], from the ORACLE's counters of the reference's own traversal (the F7 tree, every isShadowed ray a full
    intersect) on the sampled rows, scaled to the frame.  Implementation-independent; mostly cache-level re-use."""
    if not counters or "bbox_tests" not in counters:
        return None
    scale = rows_total / float(max(rows_sampled, 1))
    per_sample = (56 * counters["bbox_tests"] + 72 * counters["tri_tests"] + 72 * counters["smooth_hits"] + 128 * counters["xforms"])
    return {"bytes": 24 * width * height + per_sample * scale,
            "bbox_tests": counters["bbox_tests"] * scale, "tri_tests": counters["tri_tests"] * scale,
            "smooth_hits": counters["smooth_hits"] * scale, "xforms": counters["xforms"] * scale,
            "source": "oracle counters of %d of %d rows, scaled" % (rows_sampled, rows_total)}


def one_shot_and_moving_view(rtc, torch, hs, args, stream):
    """What the steady-state figure leaves out (the reference renders a scene ONCE, main.zig:92, and its interactive mode
    moves the camera every frame, lib.zig:166-190): the first frame of a fresh scene handle (heuristic schedule + the
    host's share), the second (re-pack), a frame with host output (rtc_render: + D2H), and an orbit of 64 frames at
    0.01 rad per frame (enqueued back to back, as an interactive host would)."""
    import numpy as np
    W, H = args.width, args.height
    sptr = stream.cuda_stream
    cam = hs.camera(W, H)
    canvas = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
    t0 = time.perf_counter()
    g = rtc.GpuScene(hs.desc)
    torch.cuda.synchronize()
    out = {"scene_create_ms": (time.perf_counter() - t0) * 1e3}   # (not the process's first)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for name in ("first_frame", "second_frame"):
        t0 = time.perf_counter()
        ev[0].record(stream)
        g.render_device(cam, canvas.data_ptr(), args.depth, None, sptr)
        ev[1].record(stream)
        stream.synchronize()
        out[name + "_ms"] = (time.perf_counter() - t0) * 1e3
        out[name + "_kernel_ms"] = ev[0].elapsed_time(ev[1])
    # What the pages of a fresh canvas cost by themselves on this host (the floor under a one-shot render with host output,
    # whatever the library does): a canvas-sized allocation touched once per page, nothing else.
    probe = np.empty((H, W, 3), dtype=np.float64)
    t0 = time.perf_counter()
    probe.reshape(-1)[::512] = 0.0
    out["host_first_touch_alone_ms"] = (time.perf_counter() - t0) * 1e3
    del probe
    host_canvas = np.empty((H, W, 3), dtype=np.float64)   # the caller's Canvas.pixels: pageable host memory
    times = []
    for _ in range(4):
        t0 = time.perf_counter()
        g.render_into(cam, host_canvas, args.depth)
        times.append((time.perf_counter() - t0) * 1e3)
    out["host_output_first_ms"] = times[0]                 # a one-shot render: the 24 B/pixel copy into pageable memory
    out["host_output_pageable_ms"] = sorted(times[1:])[1]  # ... and again into the same pageable canvas
    # (times[0] is the process's first LARGE device-to-host copy.  The HIP runtime's one-time set-up of its pageable-copy
    # path, ~7 ms - tools/first_copy_probe.py, profiles/r04/first_copy_probe.txt - is not in it any more: the first
    # rtc_scene_create on a device pays it with a 64 KB copy of its own, `scene_create_first_ms` / `scene_create_ms`.)
    # A fresh canvas pays its pages, and those are populated while the kernel runs (rtc_render's prefaultCanvas):
    fresh = np.empty((H, W, 3), dtype=np.float64)
    t0 = time.perf_counter()
    g.render_into(cam, fresh, args.depth)
    out["host_output_fresh_canvas_ms"] = (time.perf_counter() - t0) * 1e3
    del fresh
    t0 = time.perf_counter()
    rtc.canvas_register(host_canvas)                       # an interactive host pins its canvas once (rtc_canvas_register)
    out["canvas_register_ms"] = (time.perf_counter() - t0) * 1e3
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        g.render_into(cam, host_canvas, args.depth)
        times.append((time.perf_counter() - t0) * 1e3)
    rtc.canvas_unregister(host_canvas)
    out["host_output_ms"] = sorted(times)[len(times) // 2]  # kernel + copy at link speed
    host_rgba = np.empty((H, W, 4), dtype=np.uint8)        # lib.zig's RGBA8 framebuffer, clamped on the device
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        g.render_rgba8(cam, args.depth, out=host_rgba)
        times.append((time.perf_counter() - t0) * 1e3)
    out["host_output_rgba8_ms"] = sorted(times)[len(times) // 2]
    frames, angle = 64, 0.01
    for rep in range(2):                                  # the second orbit goes on from where the first stopped
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            hs.rotate_camera(angle)
            g.render_device(hs.camera(W, H), canvas.data_ptr(), args.depth, None, sptr)
        stream.synchronize()
        out["orbit_ms" if rep == 1 else "orbit_first_pass_ms"] = (time.perf_counter() - t0) * 1e3 / frames
    hs.rotate_camera(-2 * frames * angle)
    # Throughput with several frames IN FLIGHT: independent frames (an orbit's, an animation's) on a scene handle and its
    # clones (rtc_scene_clone), each on a stream of its own, so that the work-groups of frame i + 1 start on the CUs frame i's last waves have left.  Not the headline
    # (`value` is one frame after the other on one handle); what an N-way share of a frame gains from it is in
    # tools/scale_sim.py --inflight.
    g.close()
    out["frames_in_flight_ms_per_frame"] = {str(m): frames_in_flight_ms(rtc, torch, hs, args, stream, m) for m in (1, 2)}
    out["orbit"] = "%d frames, %.2f rad per frame, one handle, frames enqueued back to back; wall time / frames" % (frames, angle)
    return out


def frames_in_flight_ms(rtc, torch, hs, args, stream, m, k=30):
    """ms per frame with m independent frames IN FLIGHT: a scene handle and m - 1 clones (rtc_scene_clone), each on a stream
    of its own, frames dealt round-robin - the work-groups of frame i + 1 start on the CUs frame i's last waves have left.
    What the N > 1 lines are measured with (a rank keeps three frames in flight): timed here on ONE
    GPU so that a scaling efficiency divides like by like (value_same_inflight).  Best of three passes of k frames."""
    W, H = args.width, args.height
    cam = hs.camera(W, H)
    streams = [torch.cuda.Stream() for _ in range(m)]
    gs = [rtc.GpuScene(hs.desc)]
    gs += [gs[0].clone() for _ in range(m - 1)]
    cv = [torch.empty((H, W, 3), dtype=torch.float64, device="cuda") for _ in range(m)]
    for i in range(24 * m):
        gs[i % m].render_device(cam, cv[i % m].data_ptr(), args.depth, None, streams[i % m].cuda_stream)
    torch.cuda.synchronize()
    best = None
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for st in streams:
            st.wait_event(a)
        for i in range(k):
            gs[i % m].render_device(cam, cv[i % m].data_ptr(), args.depth, None, streams[i % m].cuda_stream)
        for st in streams:
            stream.wait_stream(st)
        b.record(stream)
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / k
        best = t if best is None else min(best, t)
    for x in gs:
        x.close()
    return best


def algorithmic_flops(desc, hs, stats):
    """SURVEY §8(d) flop table, reference arithmetic: per ray, per leaf: 56 (ray->object) + test."""
    import numpy as np
    kinds = hs.array("leaf_kind", desc.n_leaves)
    per_kind = {0: 45, 1: 3, 2: 26, 3: 52, 4: 48, 5: 48, 6: 52}
    per_ray = sum(56 + per_kind[int(k)] for k in kinds)
    rays = stats["primary"] + stats["secondary"] + stats["shadow_calls"]
    hits = stats["primary"] + stats["secondary"]  # upper bound on shaded hits
    return rays * per_ray + hits * 120 + stats["shadow_calls"] * 120 if desc.n_nodes == 0 else None


def cpu_quota():
    """CPUs this process may actually use: the cgroup's CFS quota (cpu.max / cfs_quota_us) caps a container far below
    os.cpu_count() - the GPU boxes show 256 logical CPUs and grant 16 (cpu.max = "1600000 100000"); threads beyond the
    quota only get throttled (measured there: 16 threads 14.6 Mrays/s, 256 threads 6-10)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    cores = n if quota is None else max(1, min(n, int(quota + 0.5)))
    return cores, n, quota


def cpu_baseline(rtc, hs, cam, depth, target_seconds=12.0):
    """Times the oracle (C++ restatement of the reference's CPU path, one job per row on a thread pool like
    camera.zig:88-97, per-thread arena like camera.zig:113-120) on every k-th row of the frame, k chosen from a short
    calibration so that a pass takes a few seconds on the cores this process may use."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_binding as ob
    osc = ob.OracleScene(hs.desc)
    cores, logical, quota = cpu_quota()
    t0 = time.perf_counter()
    _, c = osc.render(cam, depth, row_step=90, threads=cores)   # calibration: 12 rows
    cal = time.perf_counter() - t0
    rows_cal = (cam.vsize + 89) // 90
    per_row = cal / rows_cal
    rows_target = max(rows_cal, min(cam.vsize, int(target_seconds / 3.0 / max(per_row, 1e-9))))
    step = max(1, cam.vsize // rows_target)
    rows = (cam.vsize + step - 1) // step
    times = []
    spent = 0.0
    while len(times) < 3 or (spent < target_seconds and len(times) < 5):   # median of >= 3 passes
        t0 = time.perf_counter()
        _, c = osc.render(cam, depth, row_step=step, threads=cores)
        times.append(time.perf_counter() - t0)
        spent += times[-1]
    dt = sorted(times)[len(times) // 2]
    rays = c["primary"] + c["secondary"]
    # Thread scaling on ONE fixed sample (every 8th row), so that the figures compare: with the per-thread arena the
    # oracle scales with the cores it is given; beyond the cgroup quota more threads only get throttled.
    scaling = {}
    for th in sorted({1, 4, cores, min(logical, 4 * cores), logical}):
        t0 = time.perf_counter()
        _, cs = osc.render(cam, depth, row_step=8, threads=th)
        dts = time.perf_counter() - t0
        scaling[str(th)] = {"mrays_per_s": (cs["primary"] + cs["secondary"]) / dts / 1e6, "seconds": dts}
    one = scaling["1"]["mrays_per_s"]
    return {
        "value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample_counters": dict(c), "sample_rows": rows,
        "sample": f"every {step}th row ({rows} of {cam.vsize} rows) of the same frame, median of {len(times)} "
                  f"passes of {dt:.2f} s, {cores} threads, one job per row (camera.zig:88-97)",
        "ms_per_frame_extrapolated": dt * 1e3 * cam.vsize / rows,
        "host": {"logical_cpus": logical, "cgroup_cpu_quota": quota,
                 "note": "threads = the CPUs the container may use (cgroup quota), not os.cpu_count(): threads beyond the "
                         "quota are throttled and lower the throughput (thread_scaling; samples shorter than one 100 ms "
                         "quota period can burst above it)"},
        "thread_scaling": scaling,
        "parallel_efficiency": scaling[str(cores)]["mrays_per_s"] / (one * cores),
        "build": "oracle/ (C++ restatement of the reference's CPU path), -O3 -ffp-contract=off, per-thread bump arena "
                 "reset after every pixel like camera.zig:113-120",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--settle-frames", type=int, default=48,
                    help="untimed frames of the workload rendered during setup so that the timed steps run at the device's sustained clock")
    ap.add_argument("--scene", default="cover.json")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the first-frame / host-output / orbit timings")
    ap.add_argument("--tile-path", action="store_true",
                    help="run the multi-GPU code path (tiles + gather + un-permute) even with one rank")
    ap.add_argument("--inflight", type=int, default=0,
                    help="frames in flight per rank (that many scene handles, each with a stream and an output buffer of its "
                         "own, frames dealt round-robin); 0 = 1 on one GPU (the headline: one frame after the other), 3 when "
                         "the frame is split over ranks (a share is too short to fill a GPU by itself, DESIGN.md section 8)")
    ap.add_argument("--output", choices=("f64", "rgba8"), default="f64",
                    help="what a split frame is gathered as: the f64 Canvas (canvas.zig, the default), or the RGBA8 framebuffer of "
                         "lib.zig:146-153 - every rank clamps the tiles it rendered and 4 bytes per pixel cross xGMI instead of 24")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="rtc_set_option before anything is created (tuning / test options; no result depends on one) - the "
                         "profiling scripts pin the kernel a handle's own trial would choose (waves3) so that every PMC pass "
                         "counts the same kernel")
    ap.add_argument("--check", action="store_true", help="after timing, compare the last frame with a plain render")
    ap.add_argument("--rehearse", action="store_true",
                    help="N ranks on ONE GPU over gloo (tiles staged through host memory): exercises the N > 1 code "
                         "path where only one GPU exists; the numbers mean nothing")
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON): libraries that chat on fd 1 (RCCL prints a version banner
    # there at communicator creation) are sent to stderr for the duration of the run.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    rtc = importlib.import_module("ray-tracer-challenge_amd")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU implementation")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.tile_path:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:   # single-process rehearsal of the tile path
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    for opt in args.option:
        name, _, value = opt.partition("=")
        rtc.set_option(name, float(value))
    hs = rtc.HostScene.from_file(args.scene)
    cam = hs.camera(args.width, args.height)
    W, H = cam.hsize, cam.vsize
    t_create = time.perf_counter()
    gpu = rtc.GpuScene(hs.desc)                      # scene uploaded to HBM once (outside the timed region)
    create_first_ms = (time.perf_counter() - t_create) * 1e3   # (the process's first: + the runtime's D2H set-up, ~7 ms)
    # A non-default torch stream: the C ABI treats a NULL stream as "the handle's own stream", and
    # torch's default stream IS the NULL stream; HIP events must sit on the stream the kernel runs on.
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    sptr = stream.cuda_stream

    # Frames in flight: handle k renders frames k, k + M, ... on stream k into buffer k; the work-groups of a frame start on
    # the CUs the frame before has left.  Handle 0 / stream 0 are the ones above.
    M = args.inflight if args.inflight > 0 else (1 if world == 1 and not args.tile_path else 3)
    gpus = [gpu] + [gpu.clone() for _ in range(M - 1)]   # (rtc_scene_clone: the same device copy of the scene)
    rstreams = [stream] + [torch.cuda.Stream() for _ in range(M - 1)]

    timing = [False]   # (set for the K timed steps)
    if world == 1 and not args.tile_path:
        canvases = [torch.empty((H, W, 3), dtype=torch.float64, device="cuda") for _ in range(M)]
        canvas = canvases[0]

        def step(i):
            gpus[i % M].render_device(cam, canvases[i % M].data_ptr(), args.depth, None, rstreams[i % M].cuda_stream)

        def finish():
            for st in rstreams[1:]:
                stream.wait_stream(st)
    else:
        tx, ty = rtc.tile_grid(W, H, TILE, TILE)
        n_tiles = tx * ty
        first, stride, count, padded = rtc.tiles_of_rank(n_tiles, rank, world)
        SLOTS = max(2, M)   # output buffers (one handle: two, so that frame i + 1 renders while frame i is gathered)
        bufs = [torch.zeros((padded, TILE, TILE, 3), dtype=torch.float64, device="cuda") for _ in range(SLOTS)]
        # Scene setup, part 1: one frame with the tiles dealt round-robin MEASURES what every tile costs
        # (rtc_get_tile_costs); the ranks exchange those few hundred numbers once (not in the timed loop) and every rank
        # computes the same cost-balanced split (rtc_assign_tiles: longest tile first onto the least-loaded rank, equal
        # buffer sizes).  From here on a rank renders its tile LIST.
        cost = torch.zeros(n_tiles, dtype=torch.float64)
        if count:
            gpu.render_tiles_device(cam, bufs[0].data_ptr(), TILE, TILE, first, stride, count, args.depth, sptr)
            cost[first::stride] = torch.from_numpy(gpu.tile_costs(count))
        if world > 1:
            c = cost if args.rehearse else cost.cuda()
            dist.all_reduce(c)
            cost = c.cpu()
        rank_of, slot_of = rtc.assign_tiles(cost.numpy(), world)
        my_tiles = np.flatnonzero(rank_of == rank).astype(np.uint32)
        count = len(my_tiles)
        load = np.bincount(rank_of, weights=cost.numpy(), minlength=world)
        split_note = "%dx%d tiles dealt by measured cost (busiest rank %.2f x the mean; round-robin would be %.2f x)" % (
            TILE, TILE, load.max() / max(load.mean(), 1e-30),
            np.bincount(np.arange(n_tiles) % world, weights=cost.numpy(), minlength=world).max() / max(load.mean(), 1e-30))
        d_slot = torch.from_numpy(slot_of.astype(np.int32)).cuda() if rank == 0 else None
        # rank 0 receives straight into [world][padded][T][T][3]: the gather list is that buffer's rows
        gathered = ([torch.empty((world,) + tuple(bufs[0].shape), dtype=torch.float64, device="cuda") for _ in range(SLOTS)]
                    if rank == 0 else [None] * SLOTS)
        gather_list = [[g[r] for r in range(world)] for g in gathered] if rank == 0 else [None] * SLOTS
        canvas = torch.empty((H, W, 3), dtype=torch.float64, device="cuda") if rank == 0 else None
        RGBA = args.output == "rgba8"
        if RGBA:   # the clamped tiles (one 32-bit word per pixel), their gathered form and the framebuffer
            rgba_bufs = [torch.zeros((padded, TILE, TILE), dtype=torch.int32, device="cuda") for _ in range(SLOTS)]
            gathered_rgba = ([torch.empty((world, padded, TILE, TILE), dtype=torch.int32, device="cuda") for _ in range(SLOTS)]
                             if rank == 0 else [None] * SLOTS)
            gather_list_rgba = [[g[r] for r in range(world)] for g in gathered_rgba] if rank == 0 else [None] * SLOTS
            framebuffer = torch.empty((H, W), dtype=torch.int32, device="cuda") if rank == 0 else None
        comm = torch.cuda.Stream()
        rendered = [torch.cuda.Event() for _ in range(SLOTS)]
        gathered_ev = [torch.cuda.Event() for _ in range(SLOTS)]
        for e in gathered_ev:
            e.record(comm)
        comm_ev = []   # (timed steps only: HIP events on the side stream around the gather + un-permute of a frame)

        def step(i):
            b = i % SLOTS
            rs = rstreams[i % M]
            rs.wait_event(gathered_ev[b])            # buffer b is free again (the frame that last used it has been sent)
            if count:
                gpus[i % M].render_tile_list_device(cam, bufs[b].data_ptr(), TILE, TILE, my_tiles, args.depth, rs.cuda_stream)
            if RGBA:
                rtc.rgba8_device(bufs[b].data_ptr(), padded * TILE * TILE, rgba_bufs[b].data_ptr(), rs.cuda_stream)
            rendered[b].record(rs)
            with torch.cuda.stream(comm):            # gather + un-permute of frame i under the render of frame i+1
                comm.wait_event(rendered[b])
                if timing[0]:
                    comm_ev.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
                    comm_ev[-1][0].record(comm)
                src = rgba_bufs[b] if RGBA else bufs[b]
                dst = (gather_list_rgba if RGBA else gather_list)[b]
                if args.rehearse:                    # gloo has no device gather: through host memory
                    comm.synchronize()
                    host = [torch.empty(src.shape, dtype=src.dtype) for _ in range(world)] if rank == 0 else None
                    dist.gather(src.cpu(), host, dst=0)
                    if rank == 0:
                        for r in range(world):
                            dst[r].copy_(host[r])
                else:
                    dist.gather(src, dst, dst=0)
                if rank == 0 and RGBA:               # one un-permute kernel: tiles -> row-major framebuffer
                    rtc.assemble_tile_list_rgba8_device(gathered_rgba[b].data_ptr(), d_slot.data_ptr(), TILE, TILE, W, H,
                                                        framebuffer.data_ptr(), comm.cuda_stream)
                elif rank == 0:                      # ... -> row-major canvas
                    rtc.assemble_tile_list_device(gathered[b].data_ptr(), d_slot.data_ptr(), TILE, TILE, W, H,
                                                  canvas.data_ptr(), comm.cuda_stream)
                if timing[0]:
                    comm_ev[-1][1].record(comm)
                gathered_ev[b].record(comm)

        def finish():
            for st in rstreams[1:]:
                stream.wait_stream(st)
            stream.wait_stream(comm)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Scene setup, like the upload and the BVH build above: the first launch with a pixel map runs a schedule packed from
    # an estimate and measures (per-pixel ray counts, per-packet times); the packer behind it makes the schedule a static
    # view renders from after that (rtc_capi.hip, launch()).  Done here so that --warmup 0 still times steady-state frames.
    for i in range(2 * M):   # (every handle's two)
        step(i)
        barrier()
    # ... and the device out of its idle state: the first 20-30 ms of GPU work after start-up run about 5 % slower than
    # what follows (same handle, same schedule: 0.54 ms per frame after 6 frames, 0.516 after 50 and after 1500;
    # tools/first_handle_probe.py), and --warmup 5 --steps 20 is 14 ms of work in all.  Untimed, like the rest of the
    # setup; reported as config.settle_frames.
    cold_ms = None
    t_cold = time.perf_counter()
    for i in range(args.settle_frames):
        step(i)
        if i + 1 == min(args.steps, args.settle_frames):   # what the same K steps take right after start-up, for the record
            finish()
            barrier()
            cold_ms = (time.perf_counter() - t_cold) * 1e3 / (i + 1)
    finish()
    barrier()
    for i in range(args.warmup):
        step(i)
    finish()
    barrier()
    kernel_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    timing[0] = True
    t0 = time.perf_counter()
    for i in range(args.steps):
        kernel_ev[i][0].record(rstreams[i % M])
        step(i)
        kernel_ev[i][1].record(rstreams[i % M])
    finish()
    barrier()
    elapsed = time.perf_counter() - t0
    timing[0] = False
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    stats = gpu.stats()                              # counters of the last launch on this rank
    if dist is not None:
        v = torch.tensor([stats["primary"], stats["secondary"], stats["shadow_calls"], stats["shadow_traced"]],
                         dtype=torch.int64, device="cpu" if args.rehearse else "cuda")
        dist.all_reduce(v)
        stats = dict(zip(["primary", "secondary", "shadow_calls", "shadow_traced"], [int(x) for x in v.tolist()]))
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in kernel_ev]))   # HIP events on the launch stream
    per_rank = None
    if dist is not None:   # what every rank's share took, so that a scaling curve can be read: who was the slowest, and why
        mine = {"rank": rank, "render_ms": kernel_ms, "tiles": int(count),
                "gather_unpermute_ms": float(np.mean([a.elapsed_time(b) for a, b in comm_ev])) if comm_ev else None}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if args.check and rank == 0:
        ref = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
        gpu.render_device(cam, ref.data_ptr(), args.depth, None, sptr)
        torch.cuda.synchronize()
        # shares of one pixel's ray tree are summed in arrival order: equal up to the last bits
        if dist is not None and args.output == "rgba8":
            got = framebuffer.cpu().numpy().view(np.uint8).reshape(H, W, 4).astype(np.int64)
            want = rtc.canvas_rgba8(ref.cpu().numpy()).astype(np.int64)
            # (a channel within an ulp of k + 0.5 may round either way when its pixel's shares are summed in another order)
            worst = float(np.abs(got - want).max())
            if not worst <= 1:
                raise SystemExit(f"tile path framebuffer differs from the plain render's: max |delta| = {worst} of 255")
            print(f"check ok: assembled RGBA8 framebuffer vs plain render, max |delta| = {worst:.3g} of 255", file=sys.stderr)
        else:
            worst = float((ref - canvas).abs().max().item())
            if not worst < 1e-12:
                raise SystemExit(f"tile path result differs from the plain render: max |delta| = {worst}")
            print(f"check ok: assembled canvas vs plain render, max |delta| = {worst:.3g}", file=sys.stderr)
    if rank == 0:
        rays = stats["primary"] + stats["secondary"]
        ms_per_step = elapsed * 1e3 / args.steps
        result = {
            "metric": "Mrays/sec (primary+secondary), scenes/%s %dx%d depth %d" % (args.scene, W, H, args.depth),
            "value": rays * args.steps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "reference scene file tests/golden/scenes/%s (copy of the reference's scenes/), camera %dx%d" % (args.scene, W, H),
            "config": {"workload": "%s %dx%d depth %d, %d leaves, %d lights; tile-split %s" %
                                   (args.scene, W, H, args.depth, hs.desc.n_leaves, hs.desc.n_lights,
                                    "none (1 GPU)" if world == 1 and not args.tile_path else f"{split_note} over {world} GPUs + 1 RCCL gather/frame"),
                       "rays_per_frame": {"primary": stats["primary"], "secondary": stats["secondary"],
                                          "shadow_calls": stats["shadow_calls"], "shadow_traced": stats["shadow_traced"]},
                       "mrays_per_s_incl_shadow_traced": (rays + stats["shadow_traced"]) * args.steps / elapsed / 1e6,
                       "frames_in_flight": M,
                       "options": args.option,
                       "output": "f64 canvas" if dist is None or args.output == "f64" else "RGBA8 framebuffer (clamped by the rank that rendered the tile, 4 B/pixel gathered)",
                       "settle_frames": args.settle_frames,
                       "ms_per_step_right_after_startup": cold_ms,
                       "timed_frames": "steady state of a STATIC view: the schedule was measured on this same frame by the two "
                                       "untimed setup launches; first_frame_ms / orbit_ms below are the other cases"},
        }
        # Like for like across N: the N > 1 lines keep SPLIT_INFLIGHT frames in flight per rank (a share of a sub-millisecond
        # frame cannot fill a GPU by itself), the one-GPU headline `value` is one frame after the other.  A scaling
        # efficiency t(1) / (N t(N)) must divide figures of one kind: value_same_inflight is this N's throughput WITH
        # SPLIT_INFLIGHT frames in flight - at N > 1 that is `value` itself (unless --inflight says otherwise), at N = 1 it
        # is measured beside the headline (same scene handle pattern as the split path: a handle and its clones).
        SPLIT_INFLIGHT = 3
        if world == 1 and not args.tile_path:
            # (--no-extras - the profiling passes - skips it: three concurrent frames would mix into the per-dispatch counters
            # and durations of the kernel being profiled)
            same_ms = ms_per_step if M == SPLIT_INFLIGHT else (None if args.no_extras else frames_in_flight_ms(rtc, torch, hs, args, stream, SPLIT_INFLIGHT))
            result["value_same_inflight"] = None if same_ms is None else rays / (same_ms * 1e-3) / 1e6
            result["config"]["same_inflight"] = {"frames_in_flight": SPLIT_INFLIGHT, "ms_per_frame": same_ms,
                                                 "note": "throughput of this N with the frames in flight the N > 1 lines use; "
                                                         "efficiency = value_same_inflight(N) / (N x value_same_inflight(1))"}
        else:
            result["value_same_inflight"] = result["value"] if M == SPLIT_INFLIGHT else None
        if world == 1 and not args.tile_path:
            ab = algorithmic_bytes(hs.desc, W, H)
            fl = algorithmic_flops(hs.desc, hs, stats)
            gbs = ab / (kernel_ms * 1e-3) / 1e9
            result["roofline"] = {
                "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                "traffic": measured_traffic(args.scene, W, H, args.depth),
                "traffic_ratio": (measured_traffic(args.scene, W, H, args.depth) or 0) / ab or None,  # traffic / algorithmic bytes: > 1 is spills and pending-ray records, not canvas
                "traffic_source": "committed profile (profiles/traffic.json), not measured in this run",
                "kernel": gpu.last_kernel_name(), "kernel_ms": kernel_ms, "algorithmic_bytes": ab,
                "scene_bytes_touched_by_reference_traversal": scene_bytes_touched(hs.desc, stats),
                "note": "HBM is NOT what binds this kernel (`binding`, `executed_fp64_frac`, `lanes_active`, `valu_issuing`, "
                        "`scratch_bytes_per_lane` in this object say what does): the compulsory traffic is the canvas (24*W*H B) plus "
                        "a few KB of scene tables that live in LDS; the binding limit is per-wave FP64 issue "
                        "latency, see roofline_valu.  `traffic` is FETCH_SIZE*2 + WRITE_SIZE of profiles/ (separate "
                        "PMC passes): fabric-side requests of the L2s, mostly the lanes' pending-ray stacks and "
                        "register spills cycling through L2 into the Infinity Cache (DESIGN.md section 5), not canvas "
                        "bytes.",
            }
            ex_any = executed_flops(args.scene, W, H, args.depth)
            result["roofline"].update(what_binds(ex_any, kernel_resources(gpu.last_kernel_name())))
            if ex_any is not None:  # (the hardware figure, in the object the driver keeps: executed FP64 flops over THIS run's kernel time)
                result["roofline"]["executed_fp64_frac"] = ex_any["flops"] / (kernel_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TF
            ex_only = ex_any if fl is None else None
            if ex_only is not None:   # a scene with groups: no per-ray flop table applies; the executed figure alone
                etf = ex_only["flops"] / (kernel_ms * 1e-3) / 1e12
                result["roofline_valu"] = {"bound": "valu_fp64", "peak": FP64_VECTOR_PEAK_TF, "unit": "TFLOP/s",
                                           "executed_flops": ex_only["flops"], "executed_tflops": etf,
                                           "executed_frac": etf / FP64_VECTOR_PEAK_TF, "lanes_active": ex_only["lanes_active"],
                                           "valu_instructions": ex_only["valu_instructions"],
                                           "fp64_instructions": ex_only["fp64_instructions"], "source": ex_only["source"]}
                for k in ("valu_pipe_busy", "wave_cycles_waiting", "clock_source"):
                    if k in ex_only:
                        result["roofline_valu"][k] = ex_only[k]
            if fl is not None:
                tf = fl / (kernel_ms * 1e-3) / 1e12
                result["roofline_valu"] = {
                    "bound": "valu_fp64", "reference_tflops": tf, "peak": FP64_VECTOR_PEAK_TF, "unit": "TFLOP/s",
                    "frac_of_reference_flops": tf / FP64_VECTOR_PEAK_TF, "algorithmic_flops": fl,
                    "note": "reference_tflops / frac_of_reference_flops price the flops of the REFERENCE algorithm (SURVEY 8(d) "
                            "table: every ray tests every leaf, every isShadowed call is a full intersect) over this kernel's "
                            "time: a statement about the algorithm replaced, NOT a utilisation - the kernel skips most of those "
                            "flops by bounding-sphere rejection.  The hardware figure is executed_frac.  Peak counts an FMA as 2 "
                            "flops; the path runs with FMA contraction OFF to round like the reference.",
                }
                ex = executed_flops(args.scene, W, H, args.depth)
                if ex is not None:
                    etf = ex["flops"] / (kernel_ms * 1e-3) / 1e12
                    result["roofline_valu"].update({
                        "executed_flops": ex["flops"], "executed_tflops": etf, "executed_frac": etf / FP64_VECTOR_PEAK_TF,
                        "executed_note": "FP64 flops the kernel executes per launch (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 x 64 "
                                         "lanes x %.2f lanes active; %d of %d VALU wave-instructions are FP64 arithmetic), from "
                                         "the committed PMC pass %s over this run's kernel time: THIS is the hardware "
                                         "utilisation, frac_of_reference_flops above prices the reference's brute-force flop count"
                                         % (ex["lanes_active"], ex["fp64_instructions"], ex["valu_instructions"], ex["source"]),
                    })
                    for k in ("valu_pipe_busy", "wave_cycles_waiting", "clock_source"):
                        if k in ex:
                            result["roofline_valu"][k] = ex[k]
            if not args.no_extras:
                result["config"].update(one_shot_and_moving_view(rtc, torch, hs, args, stream))
                result["config"]["scene_create_first_ms"] = create_first_ms
                result["config"]["frames_in_flight_ms_per_frame"][str(SPLIT_INFLIGHT)] = same_ms
            if not args.no_cpu_baseline:
                result["cpu_baseline"] = cpu_baseline(rtc, hs, cam, args.depth)
                result["config"]["gpu_vs_cpu_frame_time"] = result["cpu_baseline"]["ms_per_frame_extrapolated"] / ms_per_step
                ref = reference_traversal_bytes(W, H, result["cpu_baseline"].pop("sample_counters"),
                                                result["cpu_baseline"]["sample_rows"], H)
                if ref is not None:
                    result["roofline"]["algorithmic_bytes_reference_traversal"] = ref["bytes"]
                    result["roofline"]["reference_traversal"] = dict(ref, gbs=ref["bytes"] / (kernel_ms * 1e-3) / 1e9)
        else:
            # rank 0's share of the frame: its tiles' canvas bytes over its own kernel time (HIP events on its stream)
            ab = 24 * count * TILE * TILE
            gbs = ab / (kernel_ms * 1e-3) / 1e9
            renders = [r["render_ms"] for r in per_rank]
            result["ranks"] = {
                "render_ms": renders, "render_ms_max": max(renders), "render_ms_mean": float(np.mean(renders)),
                "tiles": [r["tiles"] for r in per_rank],
                "gather_unpermute_ms_rank0": per_rank[0]["gather_unpermute_ms"],
                "note": "HIP events per rank: render_ms = a rank's render launch on its stream (incl. waiting for its "
                        "output buffer; with frames in flight the kernels of up to M frames share the GPU, so this is "
                        "longer than ms_per_step), gather_unpermute_ms = rank 0's side stream from `frame rendered` to `canvas "
                        "assembled` (the gather waits for the slowest rank); ms_per_step is max over ranks of the wall "
                        "time of K pipelined frames",
            }
            result["roofline"] = {
                "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                "traffic": None, "kernel": gpu.last_kernel_name(), "kernel_ms": kernel_ms, "algorithmic_bytes": ab,
                "note": "rank 0's kernel over rank 0's tiles (1/%d of the frame); see the 1-GPU line for the counters "
                        "and DESIGN.md section 8 for what bounds the split of a 1 ms frame" % world,
            }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
