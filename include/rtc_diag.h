/*
 * rtc_diag.h - diagnostics and tuning of librtc_hip.so: NOT part of the render path a host binds (that is rtc.h - create /
 * clone / destroy, the render forms, canvas registration, the tile entries, stats, errors; the precedent is the twelve
 * exports of the reference's /root/reference/src/lib.zig:233-309).  The test suite, bench.py and tools/ use these; no
 * rendered value depends on any of them.
 */
#ifndef RTC_DIAG_H
#define RTC_DIAG_H

#include "rtc.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Tuning and test options, process-wide; none changes a rendered value.  Read when a scene is created (bvh_*,
 * blocks_per_cu, build_threads) or a launch is enqueued (the rest).  Every option is one atomic number: setting one from
 * a thread while another thread creates scenes or enqueues launches is safe; a launch already enqueued is not affected.  Names: "simple3_min_chunks" (chunks from which a world of
 * top-level spheres / planes / cubes runs the three-waves-per-SIMD kernel; 0 always, < 0 the library's choice),
 * "sched_off" (!= 0: no schedule, packet i is chunk i), "cut_above" (shares of a wave above which a chunk is cut into
 * runs of pixels; < 0 never, 0 the library's choice), "pack_rounds", "pull_min_idle", "blocks_per_cu", "sched_tmin",
 * "bvh_leaf", "bvh_one_axis", "bvh_check", "host_bands" (bands of a host-output frame, 1 .. 4; 0 by size),
 * "waves3" (the general kernel at three waves per SIMD: 1 always, 0 never, < 0 measured per handle),
 * "sched_mix" (a | b << 8: behind every wave's first packet the schedule takes a packets from its long end, then b from
 * its short end, ...; 0 longest first throughout), "measure_every" (a view that moves in small steps is measured
 * every n-th frame; 1), "inflight_chunks_per_wave" (a launch of a scene with several handles - frames in flight -
 * runs on at most one wave per this many chunks; 3, 0 = no cap), "build_threads" (threads rtc_scene_create builds the
 * groups' candidate BVHs with, one top-level group each; 0 = the library's choice, at most 16; the tables do not depend
 * on it), "box_cull" (read at create: a world of top-level spheres / planes / cubes runs the kernels whose root loop
 * rejects by world boxes - 1 - or by bounding spheres - 0; < 0: boxes if it has more cubes than spheres).
 * RTC_ERR_INVALID_ARGUMENT for a name the library does not know.
 * (The library reads no environment variables.)
 */
int rtc_set_option(const char *name, double value);

/* Diagnostic: the name of the render kernel the last launch on this handle ran
 * (the name rocprofv3 shows - which variant is picked depends on what the world
 * contains and on the size of the launch); "" before the first launch.  Static storage.
 * (After an rtc_render that went to the host in bands, this, rtc_get_schedule and
 * rtc_get_tile_costs describe the frame's first band; rtc_get_stats the whole frame.) */
const char *rtc_last_kernel_name(const rtc_scene *scene);

/* Diagnostic: the schedule the NEXT launch of the handle's current pixel map would run - the order in which the
 * persistent waves are handed pixels (results never depend on it) - as `*n_packets` rows of 16 items; an item is
 * chunk | first_pixel << 20 | (pixels - 1) << 26 (pixels of an 8x8 chunk in row-major order), 0xFFFFFFFF = none.
 * Synchronises.  `*n_packets` = 0 when the handle has no schedule (no launch yet, or a launch of fewer than 64 chunks).
 * RTC_ERR_INVALID_ARGUMENT if `capacity_items` is too small (`*n_packets` says how many rows there are). */
int rtc_get_schedule(rtc_scene *scene, uint32_t *items, size_t capacity_items, uint32_t *n_packets);

/* Diagnostic (tools/estimate_probe.py): for the pixel map of the handle's last scheduled launch, per 8x8 chunk what the
 * first-frame estimate (rtc_estimate_kernel, for camera `cam`) says it costs and what the last measuring launch measured,
 * both in the packer's ticks of 16 shader cycles.  Either array may be NULL.  Scheduling only: no result depends on it. */
int rtc_get_chunk_times(rtc_scene *scene, const rtc_camera *cam, uint32_t *estimated, uint32_t *measured, size_t capacity,
                        uint32_t *n_chunks);

/* Diagnostic (tests/test_build_threads_cpu.py, tools/): the HOST half of rtc_scene_create by itself - validation and the
 * device tables (depth-first leaf order, bounding spheres, the candidate BVHs and their eight-wide form) - without a
 * device: `*digest` = a 64-bit FNV-1a hash over the tables the kernel walks (eight-wide nodes, leaf list, root records,
 * bounding spheres, reference-tree parents), `*build_ms` = the wall time of the table build.  Either may be NULL.  What
 * "build_threads" must not change. */
int rtc_diag_build_tables(const rtc_scene_desc *desc, uint64_t *digest, double *build_ms);

/* Diagnostic (tests/test_root_boxes_cpu.py): the FP32 world boxes phase 1 of the render kernels' root loop tests
 * (DESIGN.md section 3), as rtc_scene_create builds them, without a device: per table position i (the tables are sorted
 * by kind) `boxes[7 i ...]` = lo x y z, hi x y z, line_only (no finite bound: -3e38 / +3e38), `world_index[i]` = the
 * World.objects entry it belongs to, `scales` = {largest |coordinate| of a finite box, the "parallel rule" factor}:
 * what the test needs to replay the kernel's arithmetic on the host.  Either array may be NULL (then only *n_roots). */
int rtc_diag_root_boxes(const rtc_scene_desc *desc, float *boxes, uint32_t *world_index, uint32_t capacity, uint32_t *n_roots,
                        float *scales);

#ifdef __cplusplus
}
#endif
#endif /* RTC_DIAG_H */
