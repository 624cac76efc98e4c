/*
 * rtc_multi.h — C ABI of the single-process multi-GPU render (librtc_multi.so = librtc_hip.so + RCCL).
 *
 * The reference is ONE process that renders ONE image (src/main.zig:92: camera.render(allocator, world)).  A Zig or C
 * host that wants the 8 GPUs of a node behind that one call binds this: the flat scene is replicated on every GPU,
 * the image is cut into 64x64 tiles, every GPU renders its share into a compact buffer, ONE ncclGather per frame
 * brings the shares to GPU 0 over xGMI, one kernel un-permutes them into the row-major Canvas (canvas.zig:132-137),
 * which is copied to the caller (or clamped to RGBA8 first, or left on GPU 0: the three render entry points).  The split starts round-robin; after the first frame (and, while the camera moves,
 * every 16 frames) the tiles are re-dealt by their MEASURED cost (rtc_get_tile_costs + rtc_assign_tiles of rtc.h); the
 * frame after a re-deal measures the new shares afresh, and a split that then proves uneven (busiest rank more than a
 * quarter above the mean) is dealt again, three times at most.
 *
 * (The multi-PROCESS form of the same path - one rank per GPU under torch.distributed, which is what bench.py's
 * N > 1 mode runs - uses the per-rank entry points of rtc.h directly and the process group's gather.)
 *
 * Pixels are independent (camera.zig:116-121): the image is the one rtc_render gives, whatever the split.
 */
#ifndef RTC_MULTI_H
#define RTC_MULTI_H

#include "rtc.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtc_multi rtc_multi; /* opaque: per frame slot n scene handles and streams; n RCCL communicators */

/* All ranks on device 0, plain device copies instead of RCCL: lets a one-GPU box run the whole split / gather /
 * un-permute / re-balance logic (tests).  Without it n_gpus must not exceed the visible devices. */
#define RTC_MULTI_VIRTUAL 1u

/* Frames in flight, 1 .. 8 (flags |= RTC_MULTI_FRAMES(k); none = 1): k frame slots, each with a scene handle per GPU
 * (the scene itself and k - 1 rtc_scene_clone's of it: one device copy per GPU), streams and tile buffers of its own;
 * rtc_multi_render_device hands the frames to the slots in turn and waits only for the frame that last used the slot.
 * An N-th of a millisecond frame is a handful of dependent iterations per wave - too short to fill a GPU by itself;
 * with three slots the work-groups of a frame start on the CUs the frame before has left (one-GPU rehearsal of the
 * slowest 8-way share, tools/scale_sim.py --inflight 3: dragons 4K 0.66 -> 0.31 ms per frame, teapot 0.18 -> 0.06,
 * cover 0.18 -> 0.11).  The synchronous entry points run one frame at a time whatever k is. */
#define RTC_MULTI_FRAMES(k) (((uint32_t)(k) & 15u) << 8)

/* Replicates the scene on devices 0 .. n_gpus-1 of this process and creates the communicators (ncclCommInitAll). */
int rtc_multi_create(const rtc_scene_desc *desc, uint32_t n_gpus, uint32_t flags, rtc_multi **out);
void rtc_multi_destroy(rtc_multi *m);

/*
 * Camera.render (camera.zig:80-125) of the whole image on all GPUs: rgb_out[y * hsize + x][0..2], host memory,
 * [vsize][hsize][3] doubles.  Synchronous.  Status codes and names as in rtc.h; RTC_ERR_NO_DEVICE carries RCCL
 * failures too.
 * Where the frame goes depends on the canvas: a canvas the caller has handed to rtc_canvas_register (pinned, mapped
 * into every GPU) is written IN PLACE BY EVERY RANK - each GPU sends its own tiles over its own host link
 * (rtc_scatter_tile_list_device), no gather, nothing through GPU 0; a pageable canvas gets the frame the *_device forms
 * produce (one RCCL gather to GPU 0) and one copy over GPU 0's link.  Same bytes either way.  (4K f64: 199 MB over one
 * link is ~3.8 ms at the 53 GB/s measured; an eighth of it over each of eight links ~0.5 ms.)  rtc_multi_render_rgba8
 * does the same with 4 bytes per pixel.
 * A frame whose csg intersection lists ran out (rtc.h, limits) has them enlarged on the handles concerned
 * (rtc_grow_csg_lists) and is rendered again here; the *_device forms report that frame's RTC_ERR_OVERFLOW once (at
 * the next call that finishes its slot) and render the frames after it with the longer lists.
 */
int rtc_multi_render(rtc_multi *m, const rtc_camera *cam, uint32_t max_depth, double *rgb_out);

/*
 * The same frame as the RGBA8 framebuffer of the reference's interactive seam (Renderer, src/lib.zig:135-164; clamp of
 * color.zig:61-71, alpha 255): rgba_out[(y * hsize + x) * 4 + 0..3], host memory.  Every GPU clamps the tiles it rendered:
 * 4 bytes per pixel go through the gather and over the host link instead of 24 (at 8 GPUs the gather of an f64 frame into
 * GPU 0, 44 MB at 1080p, takes longer than a GPU's share of the render).  Synchronous.
 */
int rtc_multi_render_rgba8(rtc_multi *m, const rtc_camera *cam, uint32_t max_depth, uint8_t *rgba_out);

/*
 * The frame left on GPU 0, nothing copied and nothing waited for (but the frame that last ran on the same slot):
 * *d_canvas is device memory on device 0, [vsize][hsize][3] doubles, complete once the work enqueued on
 * rtc_multi_stream() - asked for right after this call: every slot assembles on a stream of its own - has run
 * (rtc_multi_synchronize, or the caller's own work enqueued on that stream).  max(2, k) canvases take turns: a frame's
 * canvas stays valid while the next max(2, k) - 1 frames are enqueued.  The bookkeeping a frame owes (overflow check,
 * re-deal of the tiles: a re-deal finishes every frame in flight first) is done by the next call that uses its slot, or
 * by rtc_multi_synchronize; its status is that call's status.
 */
int rtc_multi_render_device(rtc_multi *m, const rtc_camera *cam, uint32_t max_depth, const double **d_canvas);
/* ... and its RGBA8 form ([vsize][hsize] words R | G << 8 | B << 16 | 255 << 24), same rules, a ring of its own. */
int rtc_multi_render_rgba8_device(rtc_multi *m, const rtc_camera *cam, uint32_t max_depth, const uint32_t **d_rgba);
int rtc_multi_synchronize(rtc_multi *m); /* every frame in flight */
void *rtc_multi_stream(rtc_multi *m); /* the hipStream_t (device 0) behind which the LAST enqueued frame's canvas is complete */

/* (For host output at link speed register the canvas once: rtc_canvas_register of rtc.h.) */

/* Ray counters of the last frame, summed over the GPUs. */
int rtc_multi_get_stats(rtc_multi *m, rtc_stats *out);

/* The split in use: tiles per rank and the measured load share of the busiest rank over the mean (1.0 = even; 0 before
 * the first re-balance). */
int rtc_multi_balance(rtc_multi *m, uint32_t *tiles_per_rank /* [n_gpus] */, double *max_over_mean);

const char *rtc_multi_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RTC_MULTI_H */
