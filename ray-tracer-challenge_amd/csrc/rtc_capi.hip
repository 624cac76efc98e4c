// rtc_capi.hip — the extern "C" boundary of librtc_hip.so (include/rtc.h): validation of
// the flat scene, upload to HBM, kernel launches.  No torch types, no CPU render path: every
// entry point either runs the HIP kernel or fails with an rtc_status.
#include <algorithm>
#include <chrono>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstddef>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include "../../include/rtc.h"
#include "../../include/rtc_diag.h"
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include "rtc_device.h"

extern "C" __global__ void rtc_scatter_tiles_kernel(const double* tiles, const uint32_t* tile_list, const uint32_t n_tiles, const uint32_t tile_w,
                                                    const uint32_t tile_h, const uint32_t hsize, const uint32_t vsize, double* canvas);
extern "C" __global__ void rtc_scatter_tiles_rgba8_kernel(const double* tiles, const uint32_t* tile_list, const uint32_t n_tiles,
                                                          const uint32_t tile_w, const uint32_t tile_h, const uint32_t hsize,
                                                          const uint32_t vsize, uint32_t* rgba);
extern "C" __global__ void rtc_render_kernel(const DevScene S, const DevCamera cam, const DevPixelMap map,
                                             const uint32_t max_depth, double* __restrict__ out,
                                             DevStats* __restrict__ stats, DevStats* __restrict__ next_stats);
extern "C" __global__ void rtc_render_kernel_bigworld(const DevScene S, const DevCamera cam, const DevPixelMap map,
                                                      const uint32_t max_depth, double* __restrict__ out,
                                                      DevStats* __restrict__ stats, DevStats* __restrict__ next_stats);
extern "C" __global__ void rtc_render_kernel_simple(const DevScene S, const DevCamera cam, const DevPixelMap map,
                                                    const uint32_t max_depth, double* __restrict__ out,
                                                    DevStats* __restrict__ stats, DevStats* __restrict__ next_stats);
extern "C" __global__ void rtc_render_kernel_simple_b(const DevScene S, const DevCamera cam, const DevPixelMap map,
                                                      const uint32_t max_depth, double* out, DevStats* stats, DevStats* next_stats);
extern "C" __global__ void rtc_render_kernel_simple3_b(const DevScene S, const DevCamera cam, const DevPixelMap map,
                                                       const uint32_t max_depth, double* out, DevStats* stats, DevStats* next_stats);
extern "C" __global__ void rtc_render_kernel_simple3(const DevScene S, const DevCamera cam, const DevPixelMap map,
                                                     uint32_t max_depth, double* out, DevStats* stats, DevStats* next_stats);
#if RTC_BVH8
extern "C" __global__ void rtc_render_kernel3(const DevScene S, const DevCamera cam, const DevPixelMap map, const uint32_t max_depth,
                                              double* __restrict__ out, DevStats* __restrict__ stats, DevStats* __restrict__ next_stats);
#endif
extern "C" __global__ void rtc_render_kernel_simple_ext(const DevScene S, const DevCamera cam, const DevPixelMap map,
                                                        const uint32_t max_depth, double* __restrict__ out,
                                                        DevStats* __restrict__ stats, DevStats* __restrict__ next_stats);
extern "C" __global__ void rtc_render_kernel_flat(const DevScene S, const DevCamera cam, const DevPixelMap map,
                                                  const uint32_t max_depth, double* __restrict__ out,
                                                  DevStats* __restrict__ stats, DevStats* __restrict__ next_stats);
extern "C" __global__ void rtc_render_kernel_flat_ext(const DevScene S, const DevCamera cam, const DevPixelMap map,
                                                      const uint32_t max_depth, double* __restrict__ out,
                                                      DevStats* __restrict__ stats, DevStats* __restrict__ next_stats);
extern "C" __global__ void rtc_render_kernel_ext(const DevScene S, const DevCamera cam, const DevPixelMap map,
                                                 const uint32_t max_depth, double* __restrict__ out,
                                                 DevStats* __restrict__ stats, DevStats* __restrict__ next_stats);
extern "C" __global__ void rtc_render_kernel_bigworld_ext(const DevScene S, const DevCamera cam, const DevPixelMap map,
                                                          const uint32_t max_depth, double* __restrict__ out,
                                                          DevStats* __restrict__ stats, DevStats* __restrict__ next_stats);
extern "C" __global__ void rtc_estimate_kernel(const DevScene S, const DevCamera cam, const DevPixelMap map,
                                               uint32_t* __restrict__ chunk_cost, uint32_t* __restrict__ chunk_time,
                                               DevChunkShape* __restrict__ chunk_shape, DevPackState* __restrict__ state);
extern "C" __global__ void rtc_chunk_cost_kernel(const uint32_t* __restrict__ cost, const DevPixelMap map, const uint32_t max_depth,
                                                 uint32_t* __restrict__ chunk_cost, uint32_t* __restrict__ chunk_time,
                                                 DevChunkShape* __restrict__ chunk_shape, DevPackState* __restrict__ state);
extern "C" __global__ void rtc_chunk_time_kernel(const uint32_t* __restrict__ prev_order, const uint32_t* __restrict__ prev_n_units_dev,
                                                 const uint32_t prev_n_units_host, const uint32_t* __restrict__ packet_time,
                                                 const uint32_t* __restrict__ chunk_cost, const uint32_t n_chunks,
                                                 uint32_t* __restrict__ chunk_time, DevChunkShape* __restrict__ chunk_shape);
extern "C" __global__ void rtc_pack_class_kernel(const uint32_t* __restrict__ chunk_cost, const uint32_t n_chunks,
                                                 const float cost_to_time, const DevChunkShape* __restrict__ chunk_shape,
                                                 uint32_t* __restrict__ chunk_time, DevPackState* __restrict__ state);
extern "C" __global__ void rtc_pack_extra_kernel(const uint32_t* __restrict__ chunk_time, const uint32_t n_chunks, const float n_waves,
                                                 const float cut_above, const DevChunkShape* __restrict__ chunk_shape, const int round,
                                                 DevPackState* __restrict__ state);
extern "C" __global__ void rtc_pack_sort_kernel(const uint32_t* __restrict__ chunk_time, const uint32_t n_chunks, const float n_waves,
                                                const float t_min, const float cut_above, const int rounds,
                                                const DevChunkShape* __restrict__ chunk_shape, DevPackState* __restrict__ state,
                                                uint32_t* __restrict__ sorted, uint32_t* __restrict__ order_out);
extern "C" __global__ void rtc_pack_emit_kernel(const uint32_t* __restrict__ sorted, const uint32_t n_chunks, const float n_waves,
                                                const float t_min, const float cut_above, const int rounds,
                                                const DevPackState* __restrict__ state, uint32_t* __restrict__ order_out,
                                                DevSchedInfo* __restrict__ info, const int mix);
extern "C" __global__ void rtc_rgba8_kernel(const double* __restrict__ canvas, const size_t n_pixels, uint32_t* __restrict__ rgba);
extern "C" __global__ void rtc_assemble_list_kernel(const double* __restrict__ gathered, const uint32_t* __restrict__ slot_of_tile,
                                                    const uint32_t tile_w, const uint32_t tile_h, const uint32_t hsize,
                                                    const uint32_t vsize, double* __restrict__ canvas);
extern "C" __global__ void rtc_assemble_list_rgba8_kernel(const uint32_t* __restrict__ gathered, const uint32_t* __restrict__ slot_of_tile,
                                                          const uint32_t tile_w, const uint32_t tile_h, const uint32_t hsize,
                                                          const uint32_t vsize, uint32_t* __restrict__ rgba);
extern "C" __global__ void rtc_assemble_kernel(const double* __restrict__ gathered, const uint32_t world,
                                               const uint32_t padded, const uint32_t tile_w, const uint32_t tile_h,
                                               const uint32_t hsize, const uint32_t vsize, double* __restrict__ canvas);

#include "rtc_host_internal.h"
#include "rtc_bounds.h"

namespace {

bool selectChainOnly(const rtc_scene_desc& d, uint32_t idx, int depth = 0) {
  if (depth > 64) return false;
  switch (d.pat_kind[idx]) {
    case RTC_PAT_SOLID:
    case RTC_PAT_TEST: return true;
    case RTC_PAT_STRIPES:
    case RTC_PAT_CHECKERS:
    case RTC_PAT_RINGS: return selectChainOnly(d, d.pat_a[idx], depth + 1) && selectChainOnly(d, d.pat_b[idx], depth + 1);
    case RTC_PAT_PERTURB: return selectChainOnly(d, d.pat_a[idx], depth + 1);
    case RTC_PAT_TEXTURE_MAP: {  // every pattern any face can select
      if (d.pat_a[idx] >= d.n_texmaps) return false;
      for (int f = 0; f < 6; ++f) {
        const uint32_t u = d.tex_uv[6ull * d.pat_a[idx] + f];
        if (u >= d.n_uvs) return false;
        const int n_sub = d.uv_kind[u] == RTC_UV_ALIGN_CHECK ? 5 : (d.uv_kind[u] == RTC_UV_CHECKERS ? 2 : 0);
        for (int k = 0; k < n_sub; ++k)
          if (d.uv_sub[5ull * u + k] >= d.n_patterns || !selectChainOnly(d, d.uv_sub[5ull * u + k], depth + 1)) return false;
      }
      return true;
    }
    default: return false;
  }
}

// Validates the tree shape and measures the deepest traversal stack the kernel's push/pop
// order can reach, so an overflow is reported here instead of on the device.
int walkTree(const rtc_scene_desc& d, std::vector<uint8_t>& leaf_seen, std::vector<uint8_t>& node_seen,
             uint32_t root_node, uint32_t& max_stack) {
  std::vector<uint32_t> stack{root_node};
  while (!stack.empty()) {
    max_stack = std::max<uint32_t>(max_stack, static_cast<uint32_t>(stack.size()));
    const uint32_t n = stack.back();
    stack.pop_back();
    if (n >= d.n_nodes) return fail(RTC_ERR_BAD_INDEX, "node index %u >= n_nodes %u", n, d.n_nodes);
    if (node_seen[n]++) return fail(RTC_ERR_BAD_INDEX, "group node %u is referenced twice (not a tree)", n);
    const uint32_t first = d.node_first[n], count = d.node_count[n];
    if (static_cast<uint64_t>(first) + count > d.n_children)
      return fail(RTC_ERR_BAD_INDEX, "node %u children [%u,+%u) exceed n_children %u", n, first, count, d.n_children);
    for (uint32_t i = 0; i < count; ++i) {
      const uint32_t c = d.children[first + i];
      if (c & RTC_CHILD_NODE_BIT) {
        stack.push_back(c & ~RTC_CHILD_NODE_BIT);
      } else {
        if (c >= d.n_leaves) return fail(RTC_ERR_BAD_INDEX, "leaf index %u >= n_leaves %u", c, d.n_leaves);
        if (leaf_seen[c]++) return fail(RTC_ERR_BAD_INDEX, "leaf %u is referenced twice", c);
      }
    }
    max_stack = std::max<uint32_t>(max_stack, static_cast<uint32_t>(stack.size()));
  }
  return RTC_OK;
}

int buildPixelMapRect(const rtc_camera& cam, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, DevPixelMap& m) {
  if (w == 0 || h == 0) return fail(RTC_ERR_INVALID_ARGUMENT, "empty tile %ux%u", w, h);
  if (static_cast<uint64_t>(x0) + w > cam.hsize || static_cast<uint64_t>(y0) + h > cam.vsize)
    return fail(RTC_ERR_INVALID_ARGUMENT, "tile (%u,%u)+%ux%u outside the %ux%u image", x0, y0, w, h, cam.hsize,
                cam.vsize);
  std::memset(&m, 0, sizeof m);
  m.mode = 0;
  m.x0 = x0;
  m.y0 = y0;
  m.w = w;
  m.h = h;
  m.chunks_x = (w + 7) / 8;
  m.chunks_per_region = m.chunks_x * ((h + 7) / 8);
  m.n_chunks = m.chunks_per_region;
  return RTC_OK;
}

int checkCamera(const rtc_camera* cam) {
  if (!cam) return fail(RTC_ERR_INVALID_ARGUMENT, "camera is null");
  if (cam->hsize == 0 || cam->vsize == 0) return fail(RTC_ERR_INVALID_ARGUMENT, "camera %ux%u", cam->hsize, cam->vsize);
  if (!affineRow(cam->inv_view)) return fail(RTC_ERR_NOT_AFFINE, "camera inverse view last row is not (0,0,0,1)");
  return RTC_OK;
}

DevCamera devCamera(const rtc_camera& c) {
  DevCamera d;
  d.half_width = c.half_width;
  d.half_height = c.half_height;
  d.pixel_size = c.pixel_size;
  std::memcpy(d.inv, c.inv_view, sizeof d.inv);
  d.hsize = c.hsize;
  d.vsize = c.vsize;
  return d;
}

// Everything the handle has enqueued is done: its last launch, whatever stream it ran on, and the packer behind it.
hipError_t handleIdle(const rtc_scene* s) { return hipEventSynchronize(s->launch_done); }

// Both schedule buffers hold at least `words` words (16 per packet; maxPackets() of them).  Growing drops what the
// buffers held.
int ensureScheduleBuffers(rtc_scene* s, size_t words) {
  if (words <= s->sched_capacity) return RTC_OK;
  HIP_TRY(handleIdle(s));  // (the last launch may still be reading a buffer)
  for (int b = 0; b < 2; ++b) {
    if (s->d_sched[b]) (void)hipFree(s->d_sched[b]);
    s->d_sched[b] = nullptr;
  }
  s->sched_capacity = 0;
  s->sched_valid = false;
  for (int b = 0; b < 2; ++b) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_sched[b]), words * sizeof(uint32_t)));
  if (!s->d_sched_info) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_sched_info), 2 * sizeof(DevSchedInfo)));
  if (!s->d_pack_state) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_pack_state), sizeof(DevPackState)));
  s->sched_capacity = words;
  return RTC_OK;
}

// Cheap chunks are handed out several to a packet, up to this much measured time (s_memtime ticks / 16: 8000 is about
// 50 us): the chunks of a packet are image neighbours, and a wave that walks the same BVH nodes or reads the same texels
// for all of them finds them in its CU's L1; a pull of the work counter costs the wave a drain.  Measured at
// 6000 / 8000 / 12000 / 16000 (1080p, after the counters got cache lines of their own - before that a pull was dearer
// and meshes ran best at 16000): teapot 0.312 / 0.308 / 0.324 / 0.339 ms, nefertiti 0.589 / 0.599 / 0.613 / 0.632,
// cylinders 0.291 / 0.283 / 0.296 / 0.305, dragons 4K 2.43 throughout; scenes with texture maps or csg want more:
// earth 0.308 / 0.260 / 0.205 / 0.206, texture_demo 0.338 / 0.316 / 0.302 / 0.317, csg 0.688 / 0.684 / 0.665 / 0.676.
double groupFloor(const rtc_scene* s) {
  const double forced = rtcOptions().sched_tmin;
  return forced > 0.0 ? forced : (s->ext_kernel ? 12000.0 : 8000.0);
}

// The render kernel of a scene whose tables fit in LDS.
// The three-waves-per-SIMD form of the simple kernel pays when every wave has several packets to run (see the kernel):
// from about four chunks per resident wave on (1280x720; tools/simple3_sweep.py).
// Between one and four chunks per wave of the three-wave kernel neither kernel wins everywhere (1080p quarters and
// ninths, two against three waves: cover 960x540 0.304 / 0.260 ms and 640x360 0.176 / 0.182, reflection_and_refraction
// depth 8 0.741 / 0.768 and 0.556 / 0.500): the handle measures it, with the trial the worlds with groups use (KernelTune
// in launch()).
// Three waves per SIMD for this launch: the kernel under trial while a trial frame is being enqueued, else the handle's
// decision (KernelTune in launch()).
bool threeWaves(const rtc_scene* s) { return s->trial_live ? s->trial_three : s->use_three_waves; }

bool simple3Trial(const rtc_scene* s, const DevPixelMap& map) {
  if (!s->simple3_ok || rtcOptions().simple3_min_chunks >= 0.0) return false;
  if (s->tab && s->tab->handles.load(std::memory_order_relaxed) > 1) return false;  // (frames in flight: see usesSimple3)
  const uint64_t per_wave = 4ull * s->n_cus * s->blocks_per_cu_simple3;
  return map.n_chunks >= per_wave && map.n_chunks < 4ull * per_wave;
}
bool usesSimple3(const rtc_scene* s, const DevPixelMap& map) {
  const double forced = rtcOptions().simple3_min_chunks;  // (tests reach the kernel at small sizes with 0: always)
  // (a scene with several handles - rtc_scene_clone - is rendered with frames in flight: the GPU is full of other frames'
  // waves whatever this launch's size, and the kernel that does more per SIMD wins from one chunk per resident wave on.
  // The slowest 8-way share of cover with three frames in flight: 0.125 -> 0.118 ms per frame, 4-way 0.177 -> 0.164)
  const bool in_flight = s->tab && s->tab->handles.load(std::memory_order_relaxed) > 1;
  const uint64_t min_chunks = forced >= 0.0 ? static_cast<uint64_t>(forced) : (in_flight ? 1ull : 4ull) * 4u * s->n_cus * s->blocks_per_cu_simple3;
  if (!s->simple3_ok) return false;
  if (map.n_chunks >= min_chunks) return true;
  return simple3Trial(s, map) && threeWaves(s);  // (between one and four chunks per wave: what the handle's trial says)
}

bool tablesInLds(const rtc_scene* s) {
  return s->dev.n_roots <= RTC_LDS_ROOTS && s->dev.n_materials <= RTC_LDS_MATERIALS &&
         s->dev.n_patterns <= RTC_LDS_PATTERNS && s->dev.n_lights <= RTC_LDS_LIGHTS;
}

struct KernelChoice {
  decltype(&rtc_render_kernel) fn;
  const char* name;
};
#define RTC_KERNEL(k) KernelChoice{k, #k}
// The general kernel at three waves per SIMD (rtc_render_kernel3): forced by option "waves3", else what the handle's
// trial measured (KernelTune in launch()).
bool usesGeneral3(const rtc_scene* s) {
#if RTC_BVH8
  if (!s->general3_ok) return false;
  const double forced = rtcOptions().waves3;
  if (forced >= 0.0) return forced != 0.0;
  return threeWaves(s);
#else
  (void)s;
  return false;
#endif
}
KernelChoice ldsKernel(const rtc_scene* s, const DevPixelMap& map) {
  // (a simple world that is mostly cubes: the kernels whose root loop rejects by world boxes - trace() in rtc_kernels.hip)
  if (usesSimple3(s, map)) return s->box_cull ? RTC_KERNEL(rtc_render_kernel_simple3_b) : RTC_KERNEL(rtc_render_kernel_simple3);
#if RTC_BVH8
  if (usesGeneral3(s)) return RTC_KERNEL(rtc_render_kernel3);
#endif
  if (s->simple_kernel)
    return s->ext_kernel ? RTC_KERNEL(rtc_render_kernel_simple_ext) : (s->box_cull ? RTC_KERNEL(rtc_render_kernel_simple_b) : RTC_KERNEL(rtc_render_kernel_simple));
  if (s->flat_kernel) return s->ext_kernel ? RTC_KERNEL(rtc_render_kernel_flat_ext) : RTC_KERNEL(rtc_render_kernel_flat);
  return s->ext_kernel ? RTC_KERNEL(rtc_render_kernel_ext) : RTC_KERNEL(rtc_render_kernel);
}
KernelChoice renderKernel(const rtc_scene* s, const DevPixelMap& map) {
  if (tablesInLds(s)) return ldsKernel(s, map);
  return s->ext_kernel ? RTC_KERNEL(rtc_render_kernel_bigworld_ext) : RTC_KERNEL(rtc_render_kernel_bigworld);
}
#undef RTC_KERNEL

// Work-groups of the launch's kernel that are resident at once, and the waves in them.
uint32_t residentBlocksAlone(const rtc_scene* s, const DevPixelMap& map) {
  if (usesSimple3(s, map)) return s->n_cus * s->blocks_per_cu_simple3;
  if (tablesInLds(s) && usesGeneral3(s)) return s->n_cus * s->blocks_per_cu_general3;
  return s->n_cus * (tablesInLds(s) ? s->blocks_per_cu_lds : s->blocks_per_cu_big);
}
// With frames in flight (a scene with several handles) a launch gets no more waves than one per three chunks: a rank's
// share of a split frame - 4 050 chunks of cover at 8 ranks - spread over all 3 072 wave slots leaves every lane with a
// pixel or two and nothing to refill it with (SQ_THREAD_CYCLES_VALU: 0.59 lanes active against 0.84 for the whole frame,
// 1.48 x the wave-instructions per pixel), and the other frames in flight are there to fill the slots it leaves.  The
// slowest 8-way share of cover, four frames in flight: 0.096 -> 0.070 ms per frame, teapot 0.036 -> 0.033; dragons 4K
// (16 000 chunks per share) and whole frames are not affected (profiles/r04/frames_in_flight_sweeps.txt).
uint32_t residentBlocks(const rtc_scene* s, const DevPixelMap& map) {
  const uint32_t alone = residentBlocksAlone(s, map);
  const double per_wave = rtcOptions().inflight_chunks_per_wave;
  if (per_wave > 0.0 && s->tab && s->tab->handles.load(std::memory_order_relaxed) > 1) {
    const uint32_t want = static_cast<uint32_t>(std::ceil(static_cast<double>(map.n_chunks) / per_wave / 4.0));
    return std::max(16u, std::min(alone, want));
  }
  return alone;
}
double residentWaves(const rtc_scene* s, const DevPixelMap& map) { return 4.0 * residentBlocks(s, map); }

// Chunks that take more than this many fair shares of a wave are cut into runs of pixels (rtc_pack_sort_kernel).  A launch
// with few chunks per wave - a small image, a rank's share of a split frame - ends when its longest packet does and is
// cut from one share on; a large launch (a full 1080p frame's heaviest chunk is about one share) only where a chunk
// clearly sticks out: cutting costs the runs the depth of their trees again, and with a moving camera the measurement is
// a frame old (cover 1080p, threshold 1.0 / 1.5: static 0.551 / 0.543 ms, orbiting 0.579 / 0.560).
double cutAbove(const rtc_scene* s, const DevPixelMap& map) {
  const double forced = rtcOptions().cut_above;
  if (forced < 0.0) return 0.0;
  if (forced > 0.0) return forced;
  // (frames in flight - a scene with several handles -: a launch need not end when its longest packet does, the next
  // frame's waves fill in; cuts only where a chunk is two shares.  The slowest 8-way share of cover, three in flight:
  // 0.116 -> 0.109 ms per frame, 4-way 0.164 -> 0.146)
  if (s->tab && s->tab->handles.load(std::memory_order_relaxed) > 1) return 2.0;
  // One share, whatever the size of the launch.  (Up to round 3 a launch of four chunks per wave and more cut from 1.5
  // shares on: reflection_and_refraction at depth 8 - a closed box of mirrors whose heaviest chunks cost 1.3-1.6 shares -
  // then settled at 1.68 or at 1.90 ms depending on which side of 1.5 the first frame's noise put them, handle by handle;
  // from 0.6 to 1.0 shares every handle settles at 1.685 (0.4: 1.77-1.92; gpurun_out/r4_rr_cut.txt).  The other
  // scenes do not care: cover 0.514 / 0.516, teapot 0.267 / 0.266, dragons 1.911 / 1.912, csg_demo 2.57 / 2.49.)
  return 1.0;
}

// Packets a device-packed schedule of this pixel map can have at most: one per chunk, and up to fifteen more for every
// chunk that is cut - of which there are at most as many as waves (the chunks above one share cannot outnumber the shares).
size_t maxPackets(const rtc_scene* s, const DevPixelMap& map) {
  const size_t waves = static_cast<size_t>(residentWaves(s, map));
  return static_cast<size_t>(map.n_chunks) + 15u * std::min<size_t>(map.n_chunks, waves);
}

// ... and whichever of the world's kernels runs: the schedule buffers are sized for the larger of the two- and the
// three-wave kernel's resident waves, so that the frame on which a trial changes kernels packs into buffers that are
// already there (sized for the kernel in use only, that frame's schedule overran the documented bound and the next launch
// re-allocated - a host-blocking wait inside rtc_render_device, a re-estimated frame and the trial started over).
size_t maxPacketsAnyKernel(const rtc_scene* s, const DevPixelMap& map) {
  uint32_t blocks = residentBlocksAlone(s, map);
  if (s->simple3_ok) blocks = std::max(blocks, s->n_cus * s->blocks_per_cu_simple3);
  if (s->general3_ok && tablesInLds(s)) blocks = std::max(blocks, s->n_cus * s->blocks_per_cu_general3);
  return std::max(maxPackets(s, map), static_cast<size_t>(map.n_chunks) + 15u * std::min<size_t>(map.n_chunks, 4u * static_cast<size_t>(blocks)));
}

// The measured schedule in use, into a launch's pixel map.
void useSchedule(const rtc_scene* s, DevPixelMap& map) {
  map.order = s->d_sched[s->sched_cur];
  map.n_units_dev = &s->d_sched_info[s->sched_cur].n_units;
  map.n_units = static_cast<uint32_t>(maxPackets(s, map));  // (an upper bound, for the grid: the count is in device memory)
}

// Everything a measuring launch and the packer behind it write to.
int ensureMeasureBuffers(rtc_scene* s, const DevPixelMap& map) {
  if (const int st = ensureScheduleBuffers(s, maxPacketsAnyKernel(s, map) * RTC_PACKET_ITEMS); st != RTC_OK) return st;
  // (a schedule with chunks cut into runs can have more packets than there are chunks: at most 16 parts each)
  const size_t need_pt = static_cast<size_t>(map.n_chunks) * 16u;
  if (need_pt > s->packet_time_capacity) {
    HIP_TRY(handleIdle(s));
    if (s->d_packet_time) (void)hipFree(s->d_packet_time);
    s->d_packet_time = nullptr;
    s->packet_time_capacity = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_packet_time), need_pt * sizeof(uint32_t)));
    s->packet_time_capacity = need_pt;
  }
  if (map.n_chunks > s->pack_capacity) {
    HIP_TRY(handleIdle(s));
    for (uint32_t** p : {&s->d_chunk_time, &s->d_sorted, &s->d_chunk_cost}) {
      if (*p) (void)hipFree(*p);
      *p = nullptr;
    }
    if (s->d_chunk_shape) (void)hipFree(s->d_chunk_shape);
    s->d_chunk_shape = nullptr;
    s->pack_capacity = 0;
    for (uint32_t** p : {&s->d_chunk_time, &s->d_sorted, &s->d_chunk_cost})
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(p), static_cast<size_t>(map.n_chunks) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_chunk_shape), static_cast<size_t>(map.n_chunks) * sizeof(DevChunkShape)));
    s->pack_capacity = map.n_chunks;
  }
  return RTC_OK;
}

// The schedule of one launch (results never depend on it).  All of it is made on the device.
//   * first launch of a pixel map: rtc_estimate_kernel guesses what every chunk will cost from the roots its pixels can
//     see (bounding spheres, material weights) and the packer orders the frame by that.  (Round 1 did this with a
//     three-class heuristic on the host - 14 ms of wall time for dragons at 4K -, an earlier round-2 build with a probe
//     launch of one pixel per chunk.  Measured first frames, estimate / host heuristic / probe, GPU time: cover 1080p
//     0.96 / 1.22 / 1.56 ms, reflection_and_refraction depth 8 3.5 / 3.6 / 4.9, teapot 0.65 / 0.68 / 1.35, dragons 4K
//     3.2 / 5.1 / 4.3.)
//   * a launch whose schedule was not measured on its own view MEASURES: per-pixel ray counts and the time every packet
//     took in the wave that pulled it; it is followed on its stream by rtc_chunk_cost_kernel and the packer's launches:
//     the next launch runs a schedule packed on the device from those measurements.  Nothing waits for the host;
//   * so an orbiting camera (lib.zig:166-190) renders every frame with the schedule of the frame before, and a static
//     view keeps the schedule of its first full frame and measures nothing more;
//   * chunks that take more than a wave's fair share (small images, one rank's share of a frame split over GPUs) are cut
//     into runs of pixels by the packer itself (cutAbove, rtc_pack_sort_kernel): a launch ends when its longest packet does.
struct SchedulePlan {
  bool measure = false;   // this launch collects costs and packet times, and the next schedule is packed from them
  bool estimate = false;  // ... and is preceded by rtc_estimate_kernel + the packer: ITS schedule from the roots' bounds
  bool moved = false;     // the view is not the one the schedule in use was measured with
  bool near = false;      // ... but a small step away from it (nearbyView): an orbit, a walk
};

// A camera a few frames of an orbit or a walk away from `b` (lib.zig:166-190: 0.01-0.1 rad, a step of the scene's scale):
// the same image geometry, every entry of the rotation within 0.05, the position within 5 % of its distance from the
// origin (+ 0.05).  What a schedule measured at `b` is still good for.
bool nearbyView(const rtc_camera& a, const rtc_camera& b) {
  if (a.hsize != b.hsize || a.vsize != b.vsize || a.half_width != b.half_width || a.half_height != b.half_height ||
      a.pixel_size != b.pixel_size)
    return false;
  double t2 = 0.0;
  for (int r = 0; r < 3; ++r) t2 += b.inv_view[4 * r + 3] * b.inv_view[4 * r + 3];
  const double t_tol = 0.05 * std::sqrt(t2) + 0.05;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c)
      if (!(std::fabs(a.inv_view[4 * r + c] - b.inv_view[4 * r + c]) <= (c == 3 ? t_tol : 0.05))) return false;
  return true;
}

int updateSchedule(rtc_scene* s, const rtc_camera& cam, DevPixelMap& map, uint32_t max_depth, size_t out_pixels,
                   hipStream_t stream, SchedulePlan& plan) {
  plan = SchedulePlan{};
  const uint32_t* mp = reinterpret_cast<const uint32_t*>(&map);
  std::vector<uint32_t> mkey(mp, mp + offsetof(DevPixelMap, n_units) / sizeof(uint32_t));
  bool new_map = false;
  if (mkey != s->cost_key) {
    s->cost_key = mkey;
    s->sched_valid = false;
    new_map = true;
  }
  // Pixels of edge tiles that lie outside the image are never written by a launch, and rtc_chunk_cost_kernel sums whole
  // tiles: they must read as zero.  A NEW pixel map (another tile list after a re-deal, another rectangle) re-uses the
  // buffer with other tiles in the slots, so what the old map left there is cleared - on the stream, before the launch.
  if (new_map && s->d_cost != nullptr && out_pixels <= s->cost_capacity)
    HIP_TRY(hipMemsetAsync(s->d_cost, 0, out_pixels * sizeof(uint32_t), stream));
  if (out_pixels > s->cost_capacity) {
    HIP_TRY(handleIdle(s));
    if (s->d_cost) (void)hipFree(s->d_cost);
    s->d_cost = nullptr;
    s->cost_capacity = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_cost), out_pixels * sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(s->d_cost, 0, out_pixels * sizeof(uint32_t), stream));
    s->cost_capacity = out_pixels;
  }
  const bool schedulable = map.n_chunks >= 64 && map.n_chunks < RTC_ITEM_MAX_CHUNKS;
  const bool sched_off = rtcOptions().sched_off != 0.0;  // (diagnostic: no schedule at all, packet i is chunk i)
  if (!schedulable || sched_off) {  // a handful of chunks, or more than an item can name: no schedule
    map.order = nullptr;
    map.n_units_dev = nullptr;
    map.row_packets = (map.n_chunks < 64 && !sched_off) ? 1u : 0u;  // a handful: packet i is row i % 8 of chunk i / 8; else chunk i, whole
    map.n_units = map.row_packets ? map.n_chunks * 8u : map.n_chunks;
    return RTC_OK;
  }
  if (const int st = ensureMeasureBuffers(s, map); st != RTC_OK) return st;
  if (!s->sched_valid) {  // (launch() runs the estimate and the packer and then comes back through useSchedule)
    plan.estimate = plan.measure = true;
    return RTC_OK;
  }
  useSchedule(s, map);
  // A view that moves is measured every frame: the schedule in use is one frame old.  Option "measure_every" = n measures
  // a view that moves in small steps every n-th frame only (n - 1 of n frames then run without the measuring stores and
  // without the 55 us of packer kernels behind them) - which pays where the cost of a chunk does not depend much on the
  // view and loses badly where it does (orbiting at 0.01 rad per frame, every frame / every fourth: teapot 1080p
  // 0.361 / 0.316 ms per frame, cover 0.577 / 0.795, dragons 4K 2.12 / 2.30: cover's glass and dragons' silhouettes move
  // by pixels per frame, and a schedule three frames old is no better than the estimate's).  Hence n = 1.
  // A jump to another view, another depth or image is measured at once whatever n.
  plan.moved = std::memcmp(&cam, &s->sched_cam, sizeof cam) != 0 || max_depth != s->sched_depth;
  if (!plan.moved) {
    s->frames_unmeasured = 0;
  } else {
    const uint32_t every = static_cast<uint32_t>(std::max(1.0, static_cast<double>(rtcOptions().measure_every)));
    const bool near = max_depth == s->sched_depth && nearbyView(cam, s->sched_cam);
    plan.near = near;
    plan.measure = !near || ++s->frames_unmeasured >= every;
    if (plan.measure) s->frames_unmeasured = 0;
  }
  return RTC_OK;
}

// Right after a measuring launch, on its stream: per-chunk sums of the per-pixel costs, then the next frame's schedule
// packed into the buffer that is not in use; the buffers swap.  The first full measurement of a pixel map also goes to
// pinned host memory behind an event (see updateSchedule).  Nothing here waits.
enum class PackFrom { Measurement, Estimate };

int packNextSchedule(rtc_scene* s, const rtc_camera& cam, const DevPixelMap& map, uint32_t max_depth, hipStream_t stream,
                     PackFrom from) {
  const bool unmeasured = from != PackFrom::Measurement;  // (no frame has been measured: nothing to read back)
  const uint32_t n = map.n_chunks, target = s->sched_cur ^ 1u;
  const float n_waves = static_cast<float>(residentWaves(s, map)), t_min = static_cast<float>(groupFloor(s));
  // (what an untimed chunk's cost is worth in ticks: the estimate is in ticks already; a measured frame times every
  // packet, so this only covers chunks whose share of a packet's time rounded to zero)
  const float cost_to_time = from == PackFrom::Estimate ? 1.0f : (s->flat_kernel ? 10.0f : 20.0f);
  const uint32_t prev_packets = map.order == nullptr ? n : map.n_units;  // (device-packed: an upper bound)
  if (from == PackFrom::Estimate) {
    // no frame has run yet: what every chunk is likely to cost, from the roots its pixels look at (in ticks already)
    hipLaunchKernelGGL(rtc_estimate_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, s->dev, devCamera(cam), map, s->d_chunk_cost,
                       s->d_chunk_time, s->d_chunk_shape, s->d_pack_state);
  } else {
    hipLaunchKernelGGL(rtc_chunk_cost_kernel, dim3((n + 3u) / 4u), dim3(256), 0, stream, s->d_cost, map, max_depth, s->d_chunk_cost,
                       s->d_chunk_time, s->d_chunk_shape, s->d_pack_state);
    hipLaunchKernelGGL(rtc_chunk_time_kernel, dim3((prev_packets + 255u) / 256u), dim3(256), 0, stream, map.order, map.n_units_dev,
                       map.n_units, s->d_packet_time, s->d_chunk_cost, n, s->d_chunk_time, s->d_chunk_shape);
  }
  hipLaunchKernelGGL(rtc_pack_class_kernel, dim3((n + 1023u) / 1024u), dim3(1024), 0, stream, s->d_chunk_cost, n, cost_to_time,
                     s->d_chunk_shape, s->d_chunk_time, s->d_pack_state);
  const float cut_above = static_cast<float>(cutAbove(s, map));
  // Rounds of the share a wave gets with the cuts (rtc_pack_extra_kernel; an estimate has no per-pixel spread to go by).
  // An odd number: the share found in round i over-corrects that of round i - 1 (fresnel 300x300 with 0 / 1 / 2 / 3
  // rounds: 0.149 / 0.124 / 0.130 / 0.112 ms; reflection_and_refraction depth 8 at 1080p with 1 / 3: 1.78 / 1.70).
  const int env_rounds = static_cast<int>(rtcOptions().pack_rounds);
  const int rounds = cut_above > 0.0f && from == PackFrom::Measurement ? std::max(0, std::min(4, env_rounds)) : 0;
  for (int round = 0; round < rounds; ++round)
    hipLaunchKernelGGL(rtc_pack_extra_kernel, dim3((n + 1023u) / 1024u), dim3(1024), 0, stream, s->d_chunk_time, n, n_waves, cut_above,
                       s->d_chunk_shape, round, s->d_pack_state);
  hipLaunchKernelGGL(rtc_pack_sort_kernel, dim3((n + 1023u) / 1024u), dim3(1024), 0, stream, s->d_chunk_time, n, n_waves, t_min,
                     cut_above, rounds, s->d_chunk_shape, s->d_pack_state, s->d_sorted, s->d_sched[target]);
  hipLaunchKernelGGL(rtc_pack_emit_kernel, dim3((n + 1023u) / 1024u), dim3(1024), 0, stream, s->d_sorted, n, n_waves, t_min,
                     cut_above, rounds, s->d_pack_state, s->d_sched[target], s->d_sched_info + target,
                     static_cast<int>(rtcOptions().sched_mix));
  HIP_TRY(hipGetLastError());
  if (!unmeasured) {
    s->measured_regions = map.mode == 0u ? 1u : map.n_my_tiles;
    s->measured_chunks_per_region = map.chunks_per_region;
  }
  s->sched_cur = target;
  s->n_packs++;
  s->sched_valid = true;
  s->sched_cam = cam;
  s->sched_depth = max_depth;
  return RTC_OK;
}

// Per-launch scratch in HBM, grown on demand: the lanes' pending-ray stacks ([wave][level][lane], 64 B each) and,
// for scenes with csg, the lanes' intersection lists.
int ensureScratch(rtc_scene* s, DevPixelMap& map, uint32_t blocks, uint32_t max_depth) {
  const size_t need = static_cast<size_t>(blocks) * 4u * (max_depth + 2u) * 64u * 64u;
  if (need > s->ray_stack_capacity) {
    HIP_TRY(handleIdle(s));  // (the last launch may still be using the old buffer)
    if (s->d_ray_stack) (void)hipFree(s->d_ray_stack);
    s->d_ray_stack = nullptr;
    s->ray_stack_capacity = 0;
    HIP_TRY(hipMalloc(&s->d_ray_stack, need));
    s->ray_stack_capacity = need;
  }
  map.ray_stack = static_cast<PendingRec*>(s->d_ray_stack);
  map.ray_stack_levels = max_depth + 2u;
  // Measured with 1 / 16 / 32 / 48 / 64: cover 0.96 / 0.93 / 0.92 / 0.90 / 0.88 ms, reflection_and_refraction depth 8
  // 3.29 / 3.24 / 3.16 / 3.05 / 2.84 ms, dragons 4K 8.33 / 7.99 / 7.59 / 7.18 / 6.60 ms: a wave that finishes its
  // packet before it starts the next keeps neighbouring pixels (the same objects, materials, BVH paths) together;
  // the lanes that run out early are fed by the work sharing of step 2a, not by pixels of another chunk.
  // (Round 5, with the wave priorities: frames that do NOT measure pulling at 60 / 56 / 48 / 32 idle lanes - so that the
  // schedule is still packed from clean packet times -: cover 0.471 / 0.471 / 0.473 / 0.472 against 0.471, dragons 4K
  // 1.707 / 1.707 / 1.709 / 1.708 against 1.706: nothing.  profiles/r05/pull_early_times.txt)
  const uint32_t pull_min = static_cast<uint32_t>(std::max(1.0, static_cast<double>(rtcOptions().pull_min_idle)));
  map.pull_min_idle = std::max(1u, std::min(64u, pull_min));
  s->dev.csg_buf = nullptr;
  if (s->has_csg) {
    if (s->tab)  // (what another handle of this scene - a clone, a frame in flight - has already grown its lists to)
      s->dev.csg_entries = std::max(s->dev.csg_entries, std::min<uint32_t>(RTC_CSG_ENTRIES_MAX, s->tab->csg_entries_wanted.load(std::memory_order_relaxed)));
    const size_t need_csg = static_cast<size_t>(blocks) * 4u * s->dev.csg_entries * 64u * sizeof(CsgRec);
    if (need_csg > s->csg_buf_capacity) {
      HIP_TRY(handleIdle(s));  // (the last launch may still be using the old buffer)
      if (s->d_csg_buf) (void)hipFree(s->d_csg_buf);
      s->d_csg_buf = nullptr;
      s->csg_buf_capacity = 0;
      if (hipMalloc(&s->d_csg_buf, need_csg) != hipSuccess) {
        // The longer lists do not fit: the handle goes back to the length that last rendered (its buffer is allocated
        // again by the next launch), so later renders work as before instead of failing for good.
        (void)hipGetLastError();
        s->d_csg_buf = nullptr;
        const uint32_t wanted = s->dev.csg_entries;
        s->dev.csg_entries = s->csg_entries_ok;
        if (s->tab) s->tab->csg_entries_wanted.store(s->csg_entries_ok, std::memory_order_relaxed);  // (nobody else tries that length either)
        return fail(RTC_ERR_OUT_OF_MEMORY, "csg intersection lists of %u entries per lane need %zu bytes; the handle keeps %u entries", wanted,
                    need_csg, s->csg_entries_ok);
      }
      s->csg_buf_capacity = need_csg;
    }
    s->csg_entries_ok = s->dev.csg_entries;
    s->dev.csg_buf = static_cast<CsgRec*>(s->d_csg_buf);
  }
  return RTC_OK;
}

// One launch of the render kernel for `map` (schedule and measurement buffers already chosen).
int enqueueRender(rtc_scene* s, const rtc_camera& cam, DevPixelMap& map, uint32_t max_depth, double* d_out, hipStream_t stream) {
  // Persistent launch: as many work-groups as the chip can hold at once (never more than there are
  // packets to hand out, 4 waves each); the waves pull packets until the counter runs out.
  const uint32_t resident = residentBlocks(s, map);
  const uint32_t blocks = std::max(1u, std::min(resident, (map.n_units + 3u) / 4u));
  if (const int st = ensureScratch(s, map, blocks, max_depth); st != RTC_OK) return st;
  if (map.packet_time != nullptr)
    HIP_TRY(hipMemsetAsync(map.packet_time, 0, static_cast<size_t>(map.n_units) * sizeof(uint32_t), stream));
  s->stats_parity ^= 1u;
  DevStats* const st_now = s->d_stats + s->stats_parity;
  DevStats* const st_next = s->d_stats + (s->stats_parity ^ 1u);
#ifdef RTC_PROFILE
  HIP_TRY(hipMemsetAsync(st_now, 0, sizeof(DevStats), stream));
  HIP_TRY(hipMemsetAsync(&st_now->prof_t0, 0xFF, sizeof(unsigned long long), stream));
#endif
  // No clear of the canvas: every pixel of the rectangle is either stored once or zeroed by the lane that first
  // hands part of its ray tree to a neighbour (render_body step 2a).
  const KernelChoice kernel = renderKernel(s, map);
  s->last_kernel = kernel.name;
  hipLaunchKernelGGL(kernel.fn, dim3(blocks), dim3(256), 0, stream, s->dev, devCamera(cam), map, max_depth, d_out, st_now,
                     st_next);
  HIP_TRY(hipGetLastError());
  return RTC_OK;
}

int launch(rtc_scene* s, const rtc_camera& cam, const DevPixelMap& map_in, uint32_t max_depth, double* d_out,
           size_t out_pixels, hipStream_t stream) {
  DevPixelMap map = map_in;
  if (max_depth > RTC_MAX_DEPTH)
    return fail(RTC_ERR_INVALID_ARGUMENT, "max_depth %u exceeds the per-lane ray stack (%d)", max_depth, RTC_MAX_DEPTH);
  if (map.n_chunks == 0) return fail(RTC_ERR_INVALID_ARGUMENT, "nothing to render");
  HIP_TRY(hipSetDevice(s->device));
  // Launches on one handle share its counters, work counter, pending-ray stacks, csg lists and schedule buffers, and
  // launch N + 1 clears the counters of launch N + 2: they must run one after the other.  Stream order gives that on
  // one stream; when the caller changes streams (rtc_render_device on its own stream, then rtc_render on the handle's),
  // the new stream first waits for everything the handle enqueued before (the event recorded at the end of launch()).
  if (s->last_stream != nullptr && stream != s->last_stream) HIP_TRY(hipStreamWaitEvent(stream, s->launch_done, 0));
  if (!s->stats_zeroed) {  // (the handle's first launch: both counter blocks, on this stream, in front of everything)
    HIP_TRY(hipMemsetAsync(s->d_stats, 0, 2 * sizeof(DevStats), stream));
    s->stats_zeroed = true;
  }
  const bool send_ahead = !s->kernel_warm && s->d_ray_stack != nullptr;
  s->kernel_warm = true;  // (the FIRST launch only, whether or not the empty dispatch can be sent: on a later launch it would be pointless and would count into the previous launch's block)
  if (send_ahead) {
    // A handle's first launch: the render kernel goes to the stream ONCE WITH NO WORK before anything else.  The first
    // dispatch of a kernel that needs scratch memory (every render kernel spills) stalls its queue while the runtime sets
    // that memory up - 130 us between submission and start, measured in front of a 0.5 ms frame
    // (profiles/r04/first_frame_api_trace.txt).  Submitted here, that set-up runs under the ~300 us of host-side
    // allocations a first launch makes next (cost, schedule and measurement buffers), and the real kernels start when
    // they are submitted.  n_units = 0: every wave's first pull finds the counter past the end and leaves; the launch
    // counters it touches are the ones the next launch clears or finds zero.
    DevPixelMap idle = map;
    idle.order = nullptr;
    idle.n_units_dev = nullptr;
    idle.n_units = 0;
    idle.cost = nullptr;
    idle.packet_time = nullptr;
    idle.ray_stack = static_cast<PendingRec*>(s->d_ray_stack);
    idle.ray_stack_levels = 2;
    idle.pull_min_idle = 64;
    const KernelChoice kernel = renderKernel(s, map);
    DevScene dev = s->dev;
    dev.csg_buf = nullptr;
    hipLaunchKernelGGL(kernel.fn, dim3(residentBlocks(s, map)), dim3(256), 0, stream, dev, devCamera(cam), idle, max_depth, d_out,
                       s->d_stats + s->stats_parity, s->d_stats + (s->stats_parity ^ 1u));
    HIP_TRY(hipGetLastError());
  }
  s->trial_live = false;  // (set again below if this launch is a frame of a running trial)
  SchedulePlan plan;
  if (const int st = updateSchedule(s, cam, map, max_depth, out_pixels, stream, plan); st != RTC_OK) return st;
  // ---- two or three waves per SIMD for a world with groups: measured on the handle's own frames (KernelTune).  Once a
  // pixel map has a measured schedule (a static view, or one that moves in small steps), three of its frames are timed on
  // the two-wave kernel and three on the three-wave kernel (rtc_render_kernel / rtc_render_kernel3; for a simple world's
  // mid-sized launches rtc_render_kernel_simple / _simple3), each on a schedule packed for its own waves, with HIP events
  // around the render kernel (polled, never waited for); the handle keeps the three-wave kernel if its fastest frame
  // was at least 3 % faster.
  // Results do not depend on the kernel (same code, other launch bounds and table sizes).  No trial with frames in
  // flight (several handles share the GPU: a frame's time says little) or when option "waves3" forces a kernel.
  int trial_slot = -1;
  {
    rtc_scene::KernelTune& T = s->tune;
    const bool eligible = (s->general3_ok && rtcOptions().waves3 < 0.0 && tablesInLds(s) &&
                           !(s->tab && s->tab->handles.load(std::memory_order_relaxed) > 1) &&
                           static_cast<double>(map.n_chunks) >= 4.0 * 4.0 * s->n_cus * s->blocks_per_cu_lds) ||
                          simple3Trial(s, map);  // (a simple world's launch of a size where neither of its kernels always wins)
    if (eligible && T.key != s->cost_key) {  // another pixel map: a trial of its own (the last choice stands until it ends)
      T.key = s->cost_key;
      T.state = 0;
      T.frames = 0;
      T.n[0] = T.n[1] = 0;
      T.switched = false;
      for (int& w : T.which) w = -1;
    }
    // (a view that moves in small steps - consecutive frames of an orbit cost the same - is as good as a still one: an
    // interactive host whose camera never rests gets its trial too; a jump to another view makes the trial wait)
    const bool steady = (!plan.moved || plan.near) && !plan.estimate && map.order != nullptr;
    if (eligible && T.state == 1 && !steady) {  // (a jump to another view: the trial starts over when the view rests again)
      T.state = 0;
      T.frames = 0;
      T.n[0] = T.n[1] = 0;
      T.switched = false;
      for (int& w : T.which) w = -1;
    }
    if (eligible && T.state != 2 && steady) {
      for (int k = 0; k < rtc_scene::KernelTune::kRing; ++k) {
        if (T.which[k] < 0) continue;
        const hipError_t q = hipEventQuery(T.ev[k][1]);
        if (q == hipSuccess) {
          float ms = 0.0f;
          if (hipEventElapsedTime(&ms, T.ev[k][0], T.ev[k][1]) == hipSuccess) {
            const int w = T.which[k];
            T.best[w] = T.n[w] == 0 ? ms : std::min(T.best[w], ms);
            T.n[w]++;
          }
          T.which[k] = -1;
        } else {
          (void)hipGetLastError();  // (hipErrorNotReady: the frame is still running)
        }
      }
      // The trial runs in phases, so that each kernel is timed on a schedule packed for ITS number of resident waves (a
      // small launch's chunks are cut for the waves there are: the three-wave kernel on a two-wave schedule lost cover
      // 960x540 by a margin it wins by on its own): three timed frames of the two-wave kernel on the schedule in use; one
      // frame of the three-wave kernel that measures (its schedule is packed behind it); three timed frames of that; the
      // decision - and, if the two-wave kernel stays, one more measuring frame for its schedule.
      constexpr uint32_t kSamples = rtc_scene::KernelTune::kSamples;
      if (T.n[0] >= kSamples && T.n[1] >= kSamples) {
        s->use_three_waves = T.best[1] < 0.97f * T.best[0];
        T.state = 2;
        if (!s->use_three_waves) {
          // Back to the two-wave kernel and to ITS schedule: the one its samples ran is still in the other buffer if
          // nothing but the switch frame has packed since (a still view); a view that moves packs every frame anyway.
          if (s->n_packs == T.packs_at_switch + 1u && !plan.measure) {
            s->sched_cur = T.sched_before;
            useSchedule(s, map);
          } else {
            plan.measure = true;
          }
        }
      } else {
        uint32_t pending[2] = {0, 0};
        int free_slot = -1;
        for (int k = 0; k < rtc_scene::KernelTune::kRing; ++k) {
          if (T.which[k] >= 0) pending[T.which[k]]++;
          else if (free_slot < 0) free_slot = k;
        }
        T.state = 1;
        int timed = -1;  // which kernel this frame times, if any
        s->trial_live = true;
        if (T.n[0] + pending[0] < kSamples) {
          s->trial_three = false;
          timed = 0;
        } else if (!T.switched) {
          s->trial_three = true;  // (the switch: this frame runs the two-wave schedule and measures)
          plan.measure = true;
          T.switched = true;
          T.sched_before = s->sched_cur;
          T.packs_at_switch = s->n_packs;
        } else {
          s->trial_three = true;
          if (T.n[1] + pending[1] < kSamples) timed = 1;  // (else: every sample is in flight)
        }
        if (timed >= 0 && free_slot >= 0) {
          for (hipEvent_t& e : T.ev[free_slot])
            if (!e) HIP_TRY(hipEventCreate(&e));
          T.which[free_slot] = timed;
          T.frames++;
          trial_slot = free_slot;
        }
      }
    }
  }
  // Every allocation of this launch BEFORE anything is enqueued (the pending-ray levels are 59 MB at depth 5): a
  // hipMalloc between the estimate's packer and the render kernel left the GPU idle for 150 us of a first frame
  // (profiles/r04/first_frame_trace.txt).  enqueueRender finds the buffers in place.
  {
    DevPixelMap sized = map;
    const uint32_t resident = residentBlocks(s, map);
    // (the packet count of an estimate-scheduled launch is not known yet: at least a packet per resident wave whenever
    // there are that many chunks - the same bound enqueueRender applies to n_units)
    const uint32_t blocks = std::max(1u, std::min(resident, (std::max(map.n_units, map.n_chunks) + 3u) / 4u));
    if (const int st = ensureScratch(s, sized, blocks, max_depth); st != RTC_OK) return st;
  }
  if (plan.estimate) {
    DevPixelMap none = map_in;  // (no schedule ran before: the packer sees no packets and no times)
    none.order = nullptr;
    none.n_units_dev = nullptr;
    none.n_units = 0;
    if (const int st = packNextSchedule(s, cam, none, max_depth, stream, PackFrom::Estimate); st != RTC_OK) return st;
    useSchedule(s, map);
  }
  map.cost = plan.measure ? s->d_cost : nullptr;
  // (a packet's time is what its wave took for it only if the wave finishes one packet before it pulls the next - the
  // default; with the tuning option "pull_min_idle" below 64 packets overlap in a wave, their times mean nothing - a
  // schedule packed from them came out as thousands of one-pixel packets - and the packer goes by the ray counts alone)
  const bool timed = plan.measure && rtcOptions().pull_min_idle >= 64.0;
  map.packet_time = timed ? s->d_packet_time : nullptr;
  if (plan.measure && !timed) HIP_TRY(hipMemsetAsync(s->d_packet_time, 0, static_cast<size_t>(map.n_units) * sizeof(uint32_t), stream));
  if (trial_slot >= 0) HIP_TRY(hipEventRecord(s->tune.ev[trial_slot][0], stream));
  if (const int st = enqueueRender(s, cam, map, max_depth, d_out, stream); st != RTC_OK) return st;
  if (trial_slot >= 0) HIP_TRY(hipEventRecord(s->tune.ev[trial_slot][1], stream));
  // (The packer runs on the launch's own stream.  On a stream of its own, so that a frame ends with its render kernel -
  // tried in round 4 -, every hand-over between the two queues cost more than the 55 us of packer kernels it took off
  // the frame: cover orbiting 0.563 -> 0.81 ms per frame, first frame 0.79 -> 0.87.)
  if (plan.measure)
    if (const int st = packNextSchedule(s, cam, map, max_depth, stream, PackFrom::Measurement); st != RTC_OK) return st;
  HIP_TRY(hipEventRecord(s->launch_done, stream));
  s->last_stream = stream;
  s->last_bands = 1;
  return RTC_OK;
}

}  // namespace

extern "C" {

const char* rtc_last_error(void) { return g_error.c_str(); }

const char* rtc_status_name(int status) {
  switch (status) {
    case RTC_OK: return "Ok";
    case RTC_ERR_INVALID_ARGUMENT: return "InvalidArgument";
    case RTC_ERR_OUT_OF_MEMORY: return "OutOfMemory";
    case RTC_ERR_NOT_INVERTIBLE: return "NotInvertible";
    case RTC_ERR_UNSUPPORTED: return "Unsupported";
    case RTC_ERR_BAD_INDEX: return "BadIndex";
    case RTC_ERR_NO_DEVICE: return "NoDevice";
    case RTC_ERR_NOT_AFFINE: return "NotAffine";
    case RTC_ERR_OVERFLOW: return "StackOverflow";
    default: return "Unknown";
  }
}

}  // extern "C"

namespace {

struct RootCull {  // host form of one bounding sphere; uploaded two to a RootCullPair
  float cx, cy, cz, r2;
};
struct RootBox {   // host form of one world box; uploaded two to a RootBoxPair.  lo > hi: never kept (table padding)
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  float line_only = 0.0f;
};

// What validateScene learns about a scene on the way.
struct SceneTraits {
  bool has_csg = false;     // some node is a csg operation
  bool ext_kernel = false;  // csg, texture maps or nested mixing patterns: the `_ext` kernels
  bool nested_patterns = false;  // a gradient / blend below a gradient / blend
  uint32_t max_stack = 0;   // traversal stack the deepest group tree needs
};

// The device tables of a scene, on the host (buildTables fills them, uploadTables copies them).
struct HostTables {
  std::vector<uint4> leaf_meta;
  std::vector<uint32_t> roots;
  std::vector<uint32_t> kids;
  std::vector<uint32_t> leaf_parent;
  std::vector<uint32_t> node_parent;
  std::vector<BvhNode> bvh_nodes;      // the binary SAH trees (host only)
  std::vector<Bvh4Node> bvh4_nodes;    // ... collapsed to four children per node: what the kernel walks (RTC_BVH8 == 0)
  std::vector<Bvh8Node> bvh8_nodes;    // ... or to eight compressed children per node (RTC_BVH8 == 1)
  std::vector<uint32_t> bvh8_leaves;   // the leaf list in the eight-wide tree's order, then the groups' unbounded leaves
  std::vector<uint2> root_always;      // per root: {first, count} of its unbounded leaves in bvh8_leaves
  bool chain_nested = false;           // every reference node box lies inside its parent's box
  std::vector<uint32_t> bvh_leaves;
  std::vector<uint32_t> node_info;
  std::vector<uint2> node_range;
  std::vector<RootRec> root_recs;
  std::vector<RootCull> root_cull;
  std::vector<RootBox> root_box;       // the same roots' world boxes (RootBoxPair: what the render kernels' root loop tests)
  std::vector<uint32_t> root_order;    // table position -> World.objects index (the tables are sorted by kind)
  float cull_bmax = 0.0f;              // max |coordinate| of any finite root box
  float cull_par = 0.0f;               // 1.2e-5 x the largest scale of any cube (DevScene::cull_par)
  std::vector<float> root_weight;      // per root (table order): what a chunk that looks at it costs (rtc_estimate_kernel)
  std::vector<double> xf;
  std::vector<DevPattern> pat;
  std::vector<DevCyl> cyl;
  std::vector<double> tri;
  std::vector<double> trin;
  std::vector<DevMaterial> mat;
  std::vector<uint2> node_kids;
  std::vector<double> node_box;
  std::vector<double> light;
  uint32_t n_live = 0;
  uint32_t n_root_kind[3] = {0, 0, 0};  // top-level spheres, planes, cubes (the table: [spheres][cubes][the rest][planes])
  float bvh_mag = 0.0f;
  float cull_cmax = 0.0f;
};

// Everything rtc_scene_create refuses, checked on the host before anything touches the GPU.
int validateScene(const rtc_scene_desc& d, SceneTraits& traits) {
  for (uint32_t i = 0; i < d.n_xforms; ++i)
    if (!affineRow(d.xf_inv + 16ull * i)) return fail(RTC_ERR_NOT_AFFINE, "xform %u: last row is not (0,0,0,1)", i);
  for (uint32_t i = 0; i < d.n_patterns; ++i) {
    if (!affineRow(d.pat_inv + 16ull * i)) return fail(RTC_ERR_NOT_AFFINE, "pattern %u: last row is not (0,0,0,1)", i);
    const uint8_t k = d.pat_kind[i];
    if (k > RTC_PAT_TEST) return fail(RTC_ERR_UNSUPPORTED, "pattern %u: kind %u is not implemented by this kernel", i, k);
    if (k == RTC_PAT_TEXTURE_MAP && d.pat_a[i] >= d.n_texmaps)
      return fail(RTC_ERR_BAD_INDEX, "pattern %u: texture map index out of range", i);
    if (d.pat_a[i] >= d.n_patterns || d.pat_b[i] >= d.n_patterns)
      return fail(RTC_ERR_BAD_INDEX, "pattern %u: sub-pattern index out of range", i);
    if (k == RTC_PAT_PERTURB) {  // pat_rgb = PerturbInfo {scale_value, octaves, persistence}
      const double oct = d.pat_rgb[3ull * i + 1];
      if (!(oct >= 0.0 && oct <= 64.0) || oct != std::floor(oct))
        return fail(RTC_ERR_INVALID_ARGUMENT, "pattern %u: perturb octaves %g", i, oct);
    }
  }
  for (uint32_t i = 0; i < d.n_texmaps; ++i) {
    if (d.tex_mapping[i] > RTC_TEX_CUBIC) return fail(RTC_ERR_UNSUPPORTED, "texture map %u: mapping %u", i, d.tex_mapping[i]);
    for (int f = 0; f < 6; ++f)
      if (d.tex_uv[6ull * i + f] >= d.n_uvs) return fail(RTC_ERR_BAD_INDEX, "texture map %u: uv pattern index out of range", i);
  }
  for (uint32_t i = 0; i < d.n_uvs; ++i) {
    if (d.uv_kind[i] > RTC_UV_TEST) return fail(RTC_ERR_UNSUPPORTED, "uv pattern %u: kind %u", i, d.uv_kind[i]);
    for (int k = 0; k < 5; ++k)
      if (d.uv_sub[5ull * i + k] >= d.n_patterns) return fail(RTC_ERR_BAD_INDEX, "uv pattern %u: sub-pattern index out of range", i);
    if (d.uv_kind[i] == RTC_UV_IMAGE) {
      if (d.uv_image[i] >= d.n_images) return fail(RTC_ERR_BAD_INDEX, "uv pattern %u: image index out of range", i);
      const uint32_t im = d.uv_image[i];
      if (d.img_width[im] == 0 || d.img_height[im] == 0) return fail(RTC_ERR_INVALID_ARGUMENT, "image %u is empty", im);
    }
  }
  // Mixing patterns (gradient, radial gradient, blend) nested in one another run the *_ext kernels (pattern_tree);
  // the walk's stack holds RTC_PATTERN_STACK of them on one path.
  traits.nested_patterns = false;
  for (uint32_t i = 0; i < d.n_patterns; ++i) {
    const uint8_t k = d.pat_kind[i];
    if (k == RTC_PAT_GRADIENT || k == RTC_PAT_RADIAL_GRADIENT || k == RTC_PAT_BLEND) {
      if (!selectChainOnly(d, d.pat_a[i]) || !selectChainOnly(d, d.pat_b[i])) traits.nested_patterns = true;
    }
  }
  {
    // Depth of the deepest chain of patterns below each pattern (itself included): the device follows a chain for at
    // most 64 steps (pattern_chain), so a deeper one - or a cycle - is refused here rather than cut short there.
    std::vector<uint32_t> depth(d.n_patterns, 0u);  // 0: not visited, 0xFFFFFFFF: on the current path
    std::vector<uint32_t> mixing(d.n_patterns, 0u);  // gradients / blends on the deepest path below (itself included)
    std::vector<std::pair<uint32_t, uint32_t>> stack;  // (pattern, next child)
    auto kidsOf = [&](uint32_t i, uint32_t out[32]) -> uint32_t {
      switch (d.pat_kind[i]) {
        case RTC_PAT_STRIPES: case RTC_PAT_RINGS: case RTC_PAT_CHECKERS: case RTC_PAT_GRADIENT: case RTC_PAT_RADIAL_GRADIENT:
        case RTC_PAT_BLEND: out[0] = d.pat_a[i]; out[1] = d.pat_b[i]; return 2u;
        case RTC_PAT_PERTURB: out[0] = d.pat_a[i]; return 1u;
        case RTC_PAT_TEXTURE_MAP: {
          uint32_t n = 0;
          for (int f = 0; f < 6; ++f) {
            const uint32_t u = d.tex_uv[6ull * d.pat_a[i] + f];
            const uint32_t n_sub = d.uv_kind[u] == RTC_UV_ALIGN_CHECK ? 5u : (d.uv_kind[u] == RTC_UV_CHECKERS ? 2u : 0u);
            for (uint32_t k = 0; k < n_sub; ++k) out[n++] = d.uv_sub[5ull * u + k];
          }
          return n;
        }
        default: return 0u;
      }
    };
    for (uint32_t root = 0; root < d.n_patterns; ++root) {
      if (depth[root] != 0u) continue;
      stack.push_back({root, 0u});
      depth[root] = 0xFFFFFFFFu;
      while (!stack.empty()) {
        const uint32_t i = stack.back().first;
        uint32_t kids[32];
        const uint32_t nk = kidsOf(i, kids);
        if (stack.back().second < nk) {
          const uint32_t c = kids[stack.back().second++];
          if (depth[c] == 0xFFFFFFFFu) return fail(RTC_ERR_UNSUPPORTED, "pattern %u: the pattern graph has a cycle", c);
          if (depth[c] == 0u) {
            depth[c] = 0xFFFFFFFFu;
            stack.push_back({c, 0u});
          }
          continue;
        }
        uint32_t deepest = 0u, mixes = 0u;
        for (uint32_t k = 0; k < nk; ++k) {
          deepest = std::max(deepest, depth[kids[k]]);
          mixes = std::max(mixes, mixing[kids[k]]);
        }
        depth[i] = deepest + 1u;
        const uint8_t kind = d.pat_kind[i];
        mixing[i] = mixes + ((kind == RTC_PAT_GRADIENT || kind == RTC_PAT_RADIAL_GRADIENT || kind == RTC_PAT_BLEND) ? 1u : 0u);
        if (depth[i] > 64u) return fail(RTC_ERR_UNSUPPORTED, "pattern %u: a chain of more than 64 nested patterns", i);
        if (mixing[i] > RTC_PATTERN_STACK)
          return fail(RTC_ERR_UNSUPPORTED, "pattern %u: more than %d gradient / blend patterns nested in one another", i, RTC_PATTERN_STACK);
        stack.pop_back();
      }
    }
  }
  for (uint32_t i = 0; i < d.n_materials; ++i)
    if (d.mat_pattern[i] >= d.n_patterns) return fail(RTC_ERR_BAD_INDEX, "material %u: pattern index out of range", i);
  std::unordered_set<uint32_t> ids;
  for (uint32_t i = 0; i < d.n_leaves; ++i) {
    const uint8_t k = d.leaf_kind[i];
    if (k > RTC_CONE) return fail(RTC_ERR_UNSUPPORTED, "leaf %u: kind %u is not implemented by this kernel", i, k);
    if (d.leaf_xform[i] >= d.n_xforms) return fail(RTC_ERR_BAD_INDEX, "leaf %u: xform index out of range", i);
    if (d.leaf_material[i] >= d.n_materials) return fail(RTC_ERR_BAD_INDEX, "leaf %u: material index out of range", i);
    if ((k == RTC_CYLINDER || k == RTC_CONE) && d.leaf_geom[i] >= d.n_cyls)
      return fail(RTC_ERR_BAD_INDEX, "leaf %u: cylinder index out of range", i);
    if ((k == RTC_TRIANGLE || k == RTC_SMOOTH_TRIANGLE) && d.leaf_geom[i] >= d.n_tris)
      return fail(RTC_ERR_BAD_INDEX, "leaf %u: triangle index out of range", i);
    if (!ids.insert(d.leaf_id[i]).second)
      return fail(RTC_ERR_UNSUPPORTED, "leaf %u: Shape.id %u appears on more than one leaf", i, d.leaf_id[i]);
  }
  bool& has_csg = traits.has_csg;
  has_csg = false;
  for (uint32_t n = 0; n < d.n_nodes && d.node_op; ++n) {
    if (d.node_op[n] == RTC_CSG_NONE) continue;
    if (d.node_op[n] > RTC_CSG_DIFFERENCE) return fail(RTC_ERR_INVALID_ARGUMENT, "node %u: csg operation %u", n, d.node_op[n]);
    if (d.node_count[n] != 2) return fail(RTC_ERR_INVALID_ARGUMENT, "csg node %u has %u children, not left and right", n, d.node_count[n]);
    has_csg = true;
  }
  bool& ext_kernel = traits.ext_kernel;
  ext_kernel = has_csg;
  for (uint32_t i = 0; i < d.n_patterns; ++i) ext_kernel |= d.pat_kind[i] == RTC_PAT_TEXTURE_MAP;
  ext_kernel |= traits.nested_patterns;
  std::vector<uint8_t> leaf_seen(d.n_leaves, 0), node_seen(d.n_nodes, 0);
  uint32_t& max_stack = traits.max_stack;
  max_stack = 0;
  for (uint32_t i = 0; i < d.n_roots; ++i) {
    const uint32_t r = d.roots[i];
    if (r & RTC_CHILD_NODE_BIT) {
      const int st = walkTree(d, leaf_seen, node_seen, r & ~RTC_CHILD_NODE_BIT, max_stack);
      if (st != RTC_OK) return st;
    } else {
      if (r >= d.n_leaves) return fail(RTC_ERR_BAD_INDEX, "root %u: leaf index out of range", i);
      if (leaf_seen[r]++) return fail(RTC_ERR_BAD_INDEX, "leaf %u is referenced twice", r);
    }
  }
  if (max_stack > RTC_TRAV_STACK)
    return fail(RTC_ERR_OVERFLOW, "group tree needs a traversal stack of %u entries, kernel has %d", max_stack, RTC_TRAV_STACK);

  return RTC_OK;
}

// One record, one bounding sphere and one cost weight (for the first frame's schedule) per World.objects entry.
void buildRootTables(const rtc_scene_desc& d, const std::vector<uint32_t>& dfs_of, const std::vector<uint32_t>& bvh_root_of,
                     HostTables& T) {
  auto opOf = [&](uint32_t n) -> uint32_t { return d.node_op ? d.node_op[n] : RTC_CSG_NONE; };
  auto& root_recs = T.root_recs;
  auto& root_cull = T.root_cull;
  auto& cull_cmax = T.cull_cmax;
  root_recs.assign(d.n_roots, RootRec{});
  // padded to a multiple of 4 with entries no ray keeps (r2 = -inf), see trace() phase 1
  root_cull.assign((d.n_roots + 3u) & ~3u, RootCull{0.0f, 0.0f, 0.0f, -INFINITY});
  T.root_box.assign((d.n_roots + 7u) & ~7u, RootBox{});  // (phase 1 of the root loop takes the boxes in blocks of eight roots)
  T.cull_bmax = 0.0f;
  T.root_weight.assign((d.n_roots + 3u) & ~3u, 0.0f);
  cull_cmax = 0.0f;
  for (uint32_t i = 0; i < d.n_roots; ++i) {
    RootRec& R = root_recs[i];
    std::memset(&R, 0, sizeof R);
    const uint32_t ref = d.roots[i];
    Sphere sp;
    Aabb rb;                 // the same bound as a world box (not finite: no bound)
    bool line_only = false;  // entries may lie outside the bound: only "the line misses it" may cull
    if (ref & RTC_CHILD_NODE_BIT) {
      const uint32_t n = ref & ~RTC_CHILD_NODE_BIT;
      R.kind_flags = RTC_ROOT_IS_GROUP | (opOf(n) != RTC_CSG_NONE ? RTC_ROOT_IS_CSG : 0u);
      R.index = n;
      R.geom = bvh_root_of[i];
      R.always_first = T.root_always[i].x;
      R.always_count = T.root_always[i].y;
      static_assert(sizeof(Bvh8Node) <= sizeof R.inv && alignof(RootRec) >= 8, "the root node's copy lives in RootRec::inv");
      if (!(R.kind_flags & RTC_ROOT_IS_CSG) && R.geom < T.bvh8_nodes.size()) std::memcpy(R.inv, &T.bvh8_nodes[R.geom], sizeof(Bvh8Node));
      // every entry of the group lies on a line that passes the group's own box test
      const double lo[3] = {d.node_min[3ull * n], d.node_min[3ull * n + 1], d.node_min[3ull * n + 2]};
      const double hi[3] = {d.node_max[3ull * n], d.node_max[3ull * n + 1], d.node_max[3ull * n + 2]};
      const double I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
      bool csg_below = false, cone_below = false;
      {
        std::vector<uint32_t> todo{n};
        while (!todo.empty() && !csg_below) {
          const uint32_t m = todo.back();
          todo.pop_back();
          csg_below = opOf(m) != RTC_CSG_NONE;
          for (uint32_t k = 0; k < d.node_count[m]; ++k) {
            const uint32_t c = d.children[d.node_first[m] + k];
            if (c & RTC_CHILD_NODE_BIT) {
              todo.push_back(c & ~RTC_CHILD_NODE_BIT);
            } else if (d.leaf_kind[c] == RTC_CONE) {
              // A cone reports the root of a ray parallel to one of its halves without the min < y < max filter
              // (cone.zig): an entry OUTSIDE its own box, hence outside this group's.  Same consequence as a csg
              // below, except that the box itself stays a valid bound for "the line misses it": the sphere of the box
              // is kept and flagged, and the cull then skips its "entirely behind / in front of the origin" rules
              // (found by widening the random-scene fuzz to 400 more seeds).
              cone_below = true;
            }
          }
        }
      }
      line_only = cone_below;
      if (!csg_below) {
        if (lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]) {
          sp = sphereOfBox(I, lo, hi);
          rb.add(lo);
          rb.add(hi);
        }
      } else if (lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]) {
        // A csg keeps the box it was built with when a transform is pushed through it (shape.zig:298-302),
        // so its leaves - and the entries they report - may lie outside it and outside the groups above.
        // "The line misses the box => no entry" still holds; "the box is behind the origin => so is every
        // entry" does not.  The sphere therefore covers the box AND the leaves' world boxes.
        Aabb all;
        all.add(lo);
        all.add(hi);
        bool bounded = true;
        std::vector<uint32_t> todo{n};
        while (!todo.empty() && bounded) {
          const uint32_t m = todo.back();
          todo.pop_back();
          for (uint32_t k = 0; k < d.node_count[m] && bounded; ++k) {
            const uint32_t c = d.children[d.node_first[m] + k];
            if (c & RTC_CHILD_NODE_BIT) {
              todo.push_back(c & ~RTC_CHILD_NODE_BIT);
            } else {
              const Aabb b = leafWorldBox(d, c);
              if (b.finite()) {
                all.merge(b);
              } else {
                bounded = false;
              }
            }
          }
        }
        if (bounded && all.finite()) {
          sp = sphereOfBox(I, all.lo, all.hi);
          rb = all;
        }
      }
    } else {
      const uint8_t k = d.leaf_kind[ref];
      const uint32_t g = d.leaf_geom[ref];
      std::memcpy(R.inv, d.xf_inv + 16ull * d.leaf_xform[ref], sizeof R.inv);
      R.kind_flags = static_cast<uint32_t>(k) | (d.leaf_shadow[ref] ? 0x100u : 0u);
      if (k == RTC_CYLINDER || k == RTC_CONE) {
        R.ymin = d.cyl_min[g];
        R.ymax = d.cyl_max[g];
        if (d.cyl_closed[g]) R.kind_flags |= 0x200u;
      }
      if (k == RTC_CUBE && d.n_lights > 0) {  // a room: every light strictly inside the cube (object space)
        bool room = true;
        for (uint32_t l = 0; l < d.n_lights && room; ++l)
          for (int r = 0; r < 3 && room; ++r) {
            const double* m = R.inv + 4 * r;
            const double c = ((m[0] * d.light_pos[3ull * l] + m[1] * d.light_pos[3ull * l + 1]) + m[2] * d.light_pos[3ull * l + 2]) + m[3];
            room = std::fabs(c) <= 1.0 - 1e-6;
          }
        if (room) R.kind_flags |= RTC_ROOT_ROOM;
      }
      R.index = dfs_of[ref];
      R.material = d.leaf_material[ref];
      R.geom = (k == RTC_TRIANGLE || k == RTC_SMOOTH_TRIANGLE) ? g : 0u;
      sp = leafSphere(d, ref);
      rb = leafWorldBox(d, ref);
    }
    sp = inflate(sp);
    {
      // does anything under this root branch the ray tree (reflective AND transparent, world.zig:101-102)?
      bool branches = false, reflects = false, refracts = false;
      uint32_t n_below = 0;
      std::vector<uint32_t> todo{ref};
      while (!todo.empty()) {
        const uint32_t r = todo.back();
        todo.pop_back();
        if (r & RTC_CHILD_NODE_BIT) {
          const uint32_t n = r & ~RTC_CHILD_NODE_BIT;
          for (uint32_t k = 0; k < d.node_count[n]; ++k) todo.push_back(d.children[d.node_first[n] + k]);
        } else {
          const double* mp = d.mat_params + static_cast<size_t>(RTC_MAT_STRIDE) * d.leaf_material[r];
          branches = branches || (mp[4] != 0.0 && mp[5] != 0.0);
          reflects = reflects || mp[4] != 0.0;
          refracts = refracts || mp[5] != 0.0;
          ++n_below;
        }
      }
      // What a chunk of pixels that looks at this root costs, for the first frame's schedule (rtc_estimate_kernel): in
      // the packer's time unit (16 shader cycles; 150 of them a microsecond), from round-1 measurements on cover and
      // teapot - an opaque box 10 us per chunk, a mirror three times that, glass that reflects twenty-five times (the
      // heaviest cover chunk: 0.37 ms); a mesh by the depth of its BVH.  Only the order of magnitude matters: the first
      // frame measures, and the second runs on measurements.
      {
        // (round 4, tools/estimate_probe.py: with 1500 the estimates of the five configs summed to 1.9 - 5.2 x what
        // their first frames then measured - cheap chunks shared packets a third as often as they should; 600 centres them)
        float w = 600.0f * (1.0f + 0.75f * std::log2(1.0f + static_cast<float>(n_below)));
        if (branches) w *= 25.0f;
        else if (refracts) w *= 4.0f;
        else if (reflects) w *= 3.0f;
        T.root_weight[i] = w;
      }
    }
    {
      // the box: inflated like the sphere, planes rounded outward to FP32; no finite bound (or a sphere that has none: the
      // two tables say the same): -3e38 / +3e38, which every ray is inside of
      RootBox& B = T.root_box[i];
      B.line_only = line_only ? 1.0f : 0.0f;
      bool ok = rb.finite() && sp.finite();
      for (int k = 0; k < 3 && ok; ++k) {
        const double pad = 1e-6 * (std::fabs(rb.lo[k]) + std::fabs(rb.hi[k])) + 1e-9;
        B.lo[k] = BvhBuilder::down(rb.lo[k] - pad);
        B.hi[k] = BvhBuilder::up(rb.hi[k] + pad);
        ok = std::isfinite(B.lo[k]) && std::isfinite(B.hi[k]) && std::fabs(B.lo[k]) < 1.0e37f && std::fabs(B.hi[k]) < 1.0e37f;
      }
      if (ok) {
        for (int k = 0; k < 3; ++k) T.cull_bmax = std::fmax(T.cull_bmax, std::fmax(std::fabs(B.lo[k]), std::fabs(B.hi[k])));
      } else {
        for (int k = 0; k < 3; ++k) {
          B.lo[k] = -3.0e38f;
          B.hi[k] = 3.0e38f;
        }
      }
    }
    RootCull& C = root_cull[i];
    if (sp.finite()) {
      // FP32 copy: centre to nearest (its rounding is covered by the kernel's margin, which scales with
      // max|c|), r^2 rounded UP after a further 1e-5 relative inflation
      C.cx = static_cast<float>(sp.cx);
      C.cy = static_cast<float>(sp.cy);
      C.cz = static_cast<float>(sp.cz);
      const double r2 = sp.r * sp.r * (1.0 + 1e-5);
      float r2f = static_cast<float>(r2);
      if (static_cast<double>(r2f) < r2) r2f = std::nextafterf(r2f, INFINITY);
      // the lowest mantissa bit carries `line_only` (set by rounding up once more where it has the wrong value)
      uint32_t bits;
      std::memcpy(&bits, &r2f, sizeof bits);
      if ((bits & 1u) != (line_only ? 1u : 0u)) r2f = std::nextafterf(r2f, INFINITY);
      C.r2 = r2f;
      const float cm = static_cast<float>(std::sqrt(sp.cx * sp.cx + sp.cy * sp.cy + sp.cz * sp.cz) * (1.0 + 1e-6));
      cull_cmax = std::fmax(cull_cmax, std::isfinite(cm) ? cm : 0.0f);
      if (!std::isfinite(C.cx) || !std::isfinite(C.cy) || !std::isfinite(C.cz) || !std::isfinite(C.r2) || !std::isfinite(cm))
        C = RootCull{0.0f, 0.0f, 0.0f, INFINITY};  // out of FP32 range: no bound
    } else {
      C = RootCull{0.0f, 0.0f, 0.0f, INFINITY};
    }
  }
  // Sorted by kind for the kernel's root loop (trace() phase 2 runs one kind at a time): spheres, cubes, the rest, and the
  // PLANES LAST; World.objects order inside a kind.  The record carries everything that depends on the object's identity
  // (depth-first leaf index, material), so the table order is free.  A plane has no bound: phase 1 of the root loop has
  // nothing to reject it by and, with the planes at the table's tail, does not look at them at all
  // (reflection_and_refraction: seven spheres in a room of six planes - two steps of four roots per trace instead of four).
  auto klass = [&](const RootRec& R) -> int {
    if (R.kind_flags & RTC_ROOT_IS_GROUP) return 2;
    switch (R.kind_flags & 0xFFu) {
      case RTC_SPHERE: return 0;
      case RTC_CUBE: return 1;
      case RTC_PLANE: return 3;
      default: return 2;
    }
  };
  std::vector<uint32_t> perm(d.n_roots);
  for (uint32_t i = 0; i < d.n_roots; ++i) perm[i] = i;
  std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return klass(root_recs[a]) < klass(root_recs[b]); });
  std::vector<RootRec> recs2(d.n_roots);
  std::vector<RootCull> cull2(root_cull.size(), RootCull{0.0f, 0.0f, 0.0f, -INFINITY});
  std::vector<RootBox> box2(T.root_box.size(), RootBox{});
  std::vector<float> weight2(T.root_weight.size(), 0.0f);
  T.n_root_kind[0] = T.n_root_kind[1] = T.n_root_kind[2] = 0;
  for (uint32_t i = 0; i < d.n_roots; ++i) {
    recs2[i] = root_recs[perm[i]];
    cull2[i] = root_cull[perm[i]];
    box2[i] = T.root_box[perm[i]];
    weight2[i] = T.root_weight[perm[i]];
    if (!(recs2[i].kind_flags & RTC_ROOT_IS_GROUP)) {
      const uint32_t kind = recs2[i].kind_flags & 0xFFu;
      if (kind == RTC_SPHERE) T.n_root_kind[0]++;
      if (kind == RTC_PLANE) T.n_root_kind[1]++;
      if (kind == RTC_CUBE) T.n_root_kind[2]++;
    }
  }
  T.root_order = perm;
  root_recs.swap(recs2);
  root_cull.swap(cull2);
  T.root_box.swap(box2);
  T.root_weight.swap(weight2);
  // the reference's "parallel" rule for cubes (cube.zig:28-35): the largest scale of any cube of the scene, top-level or
  // inside a group (Frobenius norm of its forward transform: at least its largest singular value)
  double cube_scale = 0.0;
  for (uint32_t l = 0; l < d.n_leaves; ++l) {
    if (d.leaf_kind[l] != RTC_CUBE) continue;
    double M[12];
    if (!forwardOf(d.xf_inv + 16ull * d.leaf_xform[l], M)) {
      cube_scale = INFINITY;
      break;
    }
    double fro = 0.0;
    for (int k = 0; k < 3; ++k) fro += M[4 * k] * M[4 * k] + M[4 * k + 1] * M[4 * k + 1] + M[4 * k + 2] * M[4 * k + 2];
    cube_scale = std::fmax(cube_scale, std::sqrt(fro));
  }
  T.cull_par = static_cast<float>(std::fmin(1.2e-5 * cube_scale, 1.0e30));
}

// The tables that are the caller's arrays in the kernel's element layout.
void copyPlainTables(const rtc_scene_desc& d, HostTables& T) {
  auto& xf = T.xf;
  auto& pat = T.pat;
  auto& cyl = T.cyl;
  auto& tri = T.tri;
  auto& trin = T.trin;
  auto& mat = T.mat;
  auto& node_kids = T.node_kids;
  auto& node_box = T.node_box;
  auto& light = T.light;
  auto rows12 = [](const double* src, uint32_t n) {
    std::vector<double> v(12ull * n);
    for (uint32_t i = 0; i < n; ++i) std::memcpy(&v[12ull * i], src + 16ull * i, 12 * sizeof(double));
    return v;
  };
  xf = rows12(d.xf_inv, d.n_xforms);
  pat.assign(d.n_patterns, DevPattern{});
  for (uint32_t i = 0; i < d.n_patterns; ++i) {
    std::memset(&pat[i], 0, sizeof(DevPattern));
    std::memcpy(pat[i].inv, d.pat_inv + 16ull * i, sizeof pat[i].inv);
    for (int k = 0; k < 3; ++k) pat[i].rgb[k] = d.pat_rgb[3ull * i + k];
    pat[i].kind = d.pat_kind[i];
    pat[i].a = d.pat_a[i];
    pat[i].b = d.pat_b[i];
  }
  cyl.assign(d.n_cyls, DevCyl{});
  for (uint32_t i = 0; i < d.n_cyls; ++i) cyl[i] = {d.cyl_min[i], d.cyl_max[i], d.cyl_closed[i] ? 1u : 0u, 0u};
  tri.assign(9ull * d.n_tris, 0.0);
  trin.assign(9ull * d.n_tris, 0.0);
  for (uint32_t i = 0; i < d.n_tris; ++i) {
    for (int k = 0; k < 3; ++k) {
      tri[9ull * i + k] = d.tri_p1[3ull * i + k];
      tri[9ull * i + 3 + k] = d.tri_e1[3ull * i + k];
      tri[9ull * i + 6 + k] = d.tri_e2[3ull * i + k];
      trin[9ull * i + k] = d.tri_n1[3ull * i + k];
      trin[9ull * i + 3 + k] = d.tri_n2[3ull * i + k];
      trin[9ull * i + 6 + k] = d.tri_n3[3ull * i + k];
    }
  }
  mat.assign(d.n_materials, DevMaterial{});
  for (uint32_t i = 0; i < d.n_materials; ++i) {
    const double* p = d.mat_params + static_cast<size_t>(RTC_MAT_STRIDE) * i;
    const double shininess = p[3];
    const bool small_int = shininess >= 2.0 && shininess <= 1048576.0 && shininess == std::floor(shininess);
    mat[i] = {p[0], p[1], p[2], p[3], p[4], p[5], p[6], d.mat_pattern[i], small_int ? static_cast<uint32_t>(shininess) : 0u};
  }
  node_kids.assign(d.n_nodes, uint2{0, 0});
  node_box.assign(6ull * d.n_nodes, 0.0);
  for (uint32_t i = 0; i < d.n_nodes; ++i) {
    for (int k = 0; k < 3; ++k) {
      node_box[6ull * i + k] = d.node_min[3ull * i + k];
      node_box[6ull * i + 3 + k] = d.node_max[3ull * i + k];
    }
    node_kids[i] = {d.node_first[i], d.node_count[i]};
  }
  light.assign(6ull * d.n_lights, 0.0);
  for (uint32_t i = 0; i < d.n_lights; ++i) {
    for (int k = 0; k < 3; ++k) {
      light[6ull * i + k] = d.light_pos[3ull * i + k];
      light[6ull * i + 3 + k] = d.light_rgb[3ull * i + k];
    }
  }

}

// The flat device tables of a validated scene.
// Threads of the table build when the host does not say (option "build_threads"): what the process may use - its CPU
// affinity and, in a container, its cgroup's CPU quota (a GPU box here: 256 hardware threads, a quota of 16) - up to 16.
size_t defaultBuildThreads() {
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n, std::max(1, CPU_COUNT(&set)));
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" or "max <period>"
    long long quota = 0, period = 0;
    if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
      n = std::min<unsigned>(n, static_cast<unsigned>(std::max<long long>(1, quota / period)));
    std::fclose(f);
  }
  return std::min(16u, n);
}

int buildTables(const rtc_scene_desc& d, const SceneTraits& traits, HostTables& T) {
  (void)traits;
  auto& leaf_meta = T.leaf_meta;
  auto& roots = T.roots;
  auto& kids = T.kids;
  auto& leaf_parent = T.leaf_parent;
  auto& node_parent = T.node_parent;
  auto& bvh_nodes = T.bvh_nodes;
  auto& bvh4_nodes = T.bvh4_nodes;
  (void)bvh4_nodes;
  auto& bvh_leaves = T.bvh_leaves;
  auto& node_info = T.node_info;
  auto& node_range = T.node_range;
  auto& n_live = T.n_live;
  auto& bvh_mag = T.bvh_mag;
  auto opOf = [&](uint32_t n) -> uint32_t { return d.node_op ? d.node_op[n] : RTC_CSG_NONE; };
  // ---- device tables.  Leaves are re-indexed to depth-first order: the leaf index IS the
  // equal-t tie-break (see ClosestVisitor), whatever order the caller's arrays are in.
  std::vector<uint32_t> dfs_of(d.n_leaves, RTC_NO_LEAF);
  {
    uint32_t next = 0;
    std::vector<uint32_t> st;
    for (uint32_t i = 0; i < d.n_roots; ++i) {
      // explicit pre-order walk, children in list order
      std::vector<std::pair<uint32_t, uint32_t>> frames;  // (node, next child)
      const uint32_t r = d.roots[i];
      if (!(r & RTC_CHILD_NODE_BIT)) {
        dfs_of[r] = next++;
        continue;
      }
      frames.push_back({r & ~RTC_CHILD_NODE_BIT, 0});
      while (!frames.empty()) {
        auto& f = frames.back();
        if (f.second >= d.node_count[f.first]) {
          frames.pop_back();
          continue;
        }
        const uint32_t c = d.children[d.node_first[f.first] + f.second++];
        if (c & RTC_CHILD_NODE_BIT) {
          frames.push_back({c & ~RTC_CHILD_NODE_BIT, 0});
        } else {
          dfs_of[c] = next++;
        }
      }
    }
    // leaves not reachable from any root keep RTC_NO_LEAF and are dropped
  }
  n_live = 0;
  for (uint32_t v : dfs_of) n_live += (v != RTC_NO_LEAF);

  leaf_meta.assign(n_live, uint4{0, 0, 0, 0});
  for (uint32_t i = 0; i < d.n_leaves; ++i) {
    if (dfs_of[i] == RTC_NO_LEAF) continue;
    uint4 m;
    m.x = static_cast<uint32_t>(d.leaf_kind[i]) | (d.leaf_shadow[i] ? 0x100u : 0u);
    m.y = d.leaf_xform[i];
    m.z = d.leaf_material[i];
    m.w = d.leaf_geom[i];
    leaf_meta[dfs_of[i]] = m;
  }
  auto remap = [&](uint32_t ref) { return (ref & RTC_CHILD_NODE_BIT) ? ref : dfs_of[ref]; };
  roots.assign(d.n_roots, 0u);
  kids.assign(d.n_children, 0u);
  for (uint32_t i = 0; i < d.n_roots; ++i) roots[i] = remap(d.roots[i]);
  for (uint32_t i = 0; i < d.n_children; ++i) {
    const uint32_t c = d.children[i];
    kids[i] = (c & RTC_CHILD_NODE_BIT) ? c : (c < d.n_leaves && dfs_of[c] != RTC_NO_LEAF ? dfs_of[c] : 0u);
  }
  // reference-tree parents (for the box-chain re-check) and the candidate BVH of every group root
  leaf_parent.assign(n_live, RTC_NO_LEAF);
  node_parent.assign(d.n_nodes, RTC_NO_LEAF);
  std::vector<uint32_t> bvh_root_of(d.n_roots, 0);
  std::vector<uint32_t> bvh2_root_of(d.n_roots, 0);
  T.root_always.assign(d.n_roots, uint2{0u, 0u});
  T.bvh8_nodes.clear();
  T.bvh8_leaves.clear();
  bvh_mag = 0.0f;
  // ---- reference-tree bookkeeping for every node: parent, which child of its parent it is (a csg's left
  // is child 0), the contiguous range of depth-first leaves below it, and the csg UNITS: a csg whose parent
  // is not a csg.  A unit is evaluated as a whole (its leaves' entries are merged, sorted and filtered by
  // every csg node on the way up, csg.zig:51-95), so the candidate BVH treats it as one primitive.
  node_info.assign(d.n_nodes, 0u);       // op | slot << 8 | side << 16 | is_unit << 17
  node_range.assign(d.n_nodes, uint2{0, 0});
  std::vector<uint8_t> leaf_side(n_live, 0);
  {
    struct Frame { uint32_t node, next, unit, first_leaf; };
    std::vector<uint32_t> unit_slots(d.n_nodes, 0);  // csg nodes counted per unit (indexed by the unit's node)
    uint32_t next_leaf = 0;
    for (uint32_t i = 0; i < d.n_roots; ++i) {
      const uint32_t r = d.roots[i];
      if (!(r & RTC_CHILD_NODE_BIT)) {
        next_leaf++;
        continue;
      }
      std::vector<Frame> frames;
      auto enter = [&](uint32_t n, uint32_t parent, uint32_t side, uint32_t unit) -> int {
        uint32_t info = opOf(n);
        if (info != RTC_CSG_NONE) {
          if (unit == RTC_NO_LEAF) {
            unit = n;
            info |= 1u << 17;
          }
          const uint32_t slot = unit_slots[unit]++;
          if (slot >= 64) return fail(RTC_ERR_UNSUPPORTED, "csg node %u: more than 64 csg nodes under one csg", n);
          info |= slot << 8;
        }
        info |= side << 16;
        node_info[n] = info;
        node_parent[n] = parent;
        frames.push_back({n, 0, unit, next_leaf});
        return RTC_OK;
      };
      int st = enter(r & ~RTC_CHILD_NODE_BIT, RTC_NO_LEAF, 0, RTC_NO_LEAF);
      if (st != RTC_OK) return st;
      while (!frames.empty()) {
        Frame& f = frames.back();
        if (f.next >= d.node_count[f.node]) {
          node_range[f.node] = uint2{f.first_leaf, next_leaf - f.first_leaf};
          frames.pop_back();
          continue;
        }
        const uint32_t k = f.next++;
        const uint32_t c = d.children[d.node_first[f.node] + k];
        const uint32_t node = f.node, unit = f.unit;
        if (c & RTC_CHILD_NODE_BIT) {
          st = enter(c & ~RTC_CHILD_NODE_BIT, node, k == 0 ? 0u : 1u, unit);  // `f` is dangling from here on
          if (st != RTC_OK) return st;
        } else {
          leaf_parent[dfs_of[c]] = node;
          leaf_side[dfs_of[c]] = k == 0 ? 0 : 1;
          next_leaf++;
        }
      }
    }
  }
  for (uint32_t l = 0; l < n_live; ++l)
    if (leaf_side[l]) leaf_meta[l].x |= 0x400u;
  // ---- the candidate BVH of every top-level GROUP (a top-level csg is one unit and needs none)
  auto unitPrim = [&](uint32_t n) {
    BvhPrim p;
    // world box of everything below the unit
    std::vector<uint32_t> todo{n};
    Aabb box;
    bool bounded = true;
    bool any = false;
    while (!todo.empty()) {
      const uint32_t m = todo.back();
      todo.pop_back();
      for (uint32_t k = 0; k < d.node_count[m]; ++k) {
        const uint32_t c = d.children[d.node_first[m] + k];
        if (c & RTC_CHILD_NODE_BIT) {
          todo.push_back(c & ~RTC_CHILD_NODE_BIT);
        } else {
          const Aabb b = leafWorldBox(d, c);
          if (!b.finite()) {
            bounded = false;
          } else {
            for (int a = 0; a < 3; ++a) {
              box.lo[a] = any ? std::fmin(box.lo[a], b.lo[a]) : b.lo[a];
              box.hi[a] = any ? std::fmax(box.hi[a], b.hi[a]) : b.hi[a];
            }
            any = true;
          }
        }
      }
    }
    p.box = (bounded && any) ? box : Aabb{};
    for (int a = 0; a < 3; ++a) p.c[a] = p.box.finite() ? 0.5 * (p.box.lo[a] + p.box.hi[a]) : 0.0;
    p.leaf = RTC_NODE_BIT | n;
    return p;
  };
  // One candidate BVH per top-level group.  The groups' trees are independent, and the build - a 16-bin SAH over the
  // world boxes of every leaf below the group, then the eight-wide collapse - is most of rtc_scene_create for a mesh
  // scene (dragons.json: 141 k leaves in six groups): every group is built by itself, into tables of its own with
  // indices that start at zero, on up to eight threads (option "build_threads"), and the tables are then joined in
  // World.objects order with the indices moved by where each group's part begins - exactly the tables one thread
  // building group after group into shared vectors makes (tests/test_build_threads_cpu.py holds the two bit-equal).
  struct GroupBuild {
    std::vector<BvhNode> nodes;       // binary tree, local indices
    std::vector<uint32_t> leaves;     // its leaf list
    std::vector<Bvh8Node> nodes8;     // eight-wide tree, local indices
    std::vector<uint32_t> leaves8;    // its leaf records' leaves, then the group's unbounded leaves
    uint32_t n_tree_leaves8 = 0, n_unbounded = 0;
    uint32_t root2 = 0, root8 = 0;
    float mag = 0.0f;
    int status = RTC_OK;              // RTC_ERR_UNSUPPORTED: the eight-wide encoding; RTC_ERR_OVERFLOW: too deep
    uint32_t depth = 0;
  };
  std::vector<uint32_t> group_roots;
  for (uint32_t i = 0; i < d.n_roots; ++i) {
    const uint32_t r = d.roots[i];
    if (!(r & RTC_CHILD_NODE_BIT)) continue;
    if (opOf(r & ~RTC_CHILD_NODE_BIT) != RTC_CSG_NONE) continue;
    group_roots.push_back(i);
  }
  // (the plain tables - transforms, triangles, normals, materials, patterns: copies of the caller's arrays that nothing
  // below reads - are made beside the groups' builds, on a thread of their own)
  std::thread plain_tables([&]() { copyPlainTables(d, T); });
  struct JoinOnExit {
    std::thread& t;
    ~JoinOnExit() {
      if (t.joinable()) t.join();
    }
  } plain_tables_joined{plain_tables};
  std::vector<GroupBuild> built(group_roots.size());
  // (threads: one group each while there are groups, and what is left over inside a group's own build - a single mesh,
  // nefertiti.json, has all of them: BvhBuilder::buildFixed)
  const double threads_asked = rtcOptions().build_threads;
  const size_t threads_total = threads_asked >= 1.0 ? static_cast<size_t>(threads_asked) : defaultBuildThreads();
  const size_t group_threads = std::max<size_t>(1, std::min(built.size(), threads_total));
  int spawn_levels = 0;
  while ((group_threads << (spawn_levels + 1)) <= threads_total) ++spawn_levels;
  auto buildGroup = [&](size_t g) {
    GroupBuild& G = built[g];
    const uint32_t r = d.roots[group_roots[g]];
    std::vector<BvhPrim> items;
    std::vector<uint32_t> todo{r & ~RTC_CHILD_NODE_BIT};
    while (!todo.empty()) {
      const uint32_t n = todo.back();
      todo.pop_back();
      for (uint32_t k = 0; k < d.node_count[n]; ++k) {
        const uint32_t c = d.children[d.node_first[n] + k];
        if (c & RTC_CHILD_NODE_BIT) {
          if (opOf(c & ~RTC_CHILD_NODE_BIT) != RTC_CSG_NONE) {
            items.push_back(unitPrim(c & ~RTC_CHILD_NODE_BIT));
          } else {
            todo.push_back(c & ~RTC_CHILD_NODE_BIT);
          }
        } else {
          BvhPrim p;
          p.box = leafWorldBox(d, c);
          for (int a = 0; a < 3; ++a) p.c[a] = p.box.finite() ? 0.5 * (p.box.lo[a] + p.box.hi[a]) : 0.0;
          p.leaf = dfs_of[c];
          items.push_back(p);
        }
      }
    }
#if RTC_BVH8
    // Leaves (and csg units) without a finite bound - planes, cones - stay out of the tree: the walk visits them first,
    // unconditionally, and every box of the tree is finite (the eight-wide nodes quantise their children's boxes).
    std::vector<BvhPrim> unbounded;
    {
      std::vector<BvhPrim> bounded;
      for (BvhPrim& p : items) (p.box.finite() ? bounded : unbounded).push_back(p);
      items.swap(bounded);
    }
#endif
    BvhBuilder builder{G.nodes, G.leaves};
    G.root2 = builder.buildRoot(std::move(items), spawn_levels);
    G.mag = builder.mag;
#if RTC_BVH8
    Bvh8Collapse wide{G.nodes, G.leaves, G.nodes8, G.leaves8};
    wide.leaf_meta = &leaf_meta;
    G.root8 = wide.convertRoot(G.root2);
    G.depth = wide.max_depth;
    if (!wide.ok) {
      G.status = RTC_ERR_UNSUPPORTED;
      return;
    }
    if (wide.max_depth + 1 > RTC_TRAV_STACK) {
      G.status = RTC_ERR_OVERFLOW;
      return;
    }
    G.n_tree_leaves8 = static_cast<uint32_t>(G.leaves8.size());
    G.n_unbounded = static_cast<uint32_t>(unbounded.size());
    for (const BvhPrim& p : unbounded) G.leaves8.push_back(p.leaf);
#endif
  };
  {
    const size_t n_threads = group_threads;
    if (n_threads <= 1) {
      for (size_t g = 0; g < built.size(); ++g) buildGroup(g);
    } else {
      std::atomic<size_t> next{0};
      std::vector<std::thread> workers;
      for (size_t t = 0; t < n_threads; ++t)
        workers.emplace_back([&]() {
          for (size_t g = next.fetch_add(1); g < built.size(); g = next.fetch_add(1)) buildGroup(g);
        });
      for (std::thread& w : workers) w.join();
    }
  }
  {
    size_t n2 = 0, l2 = 0, n8 = 0, l8 = 0;
    for (const GroupBuild& G : built) {
      n2 += G.nodes.size();
      l2 += G.leaves.size();
      n8 += G.nodes8.size();
      l8 += G.leaves8.size();
    }
    bvh_nodes.reserve(bvh_nodes.size() + n2);
    bvh_leaves.reserve(bvh_leaves.size() + l2);
    T.bvh8_nodes.reserve(T.bvh8_nodes.size() + n8);
    T.bvh8_leaves.reserve(T.bvh8_leaves.size() + l8);
  }
  for (size_t g = 0; g < built.size(); ++g) {  // joined in World.objects order
    GroupBuild& G = built[g];
    const uint32_t i = group_roots[g];
#if RTC_BVH8
    if (G.status == RTC_ERR_UNSUPPORTED) return fail(RTC_ERR_UNSUPPORTED, "root %u: its candidate BVH does not fit the eight-wide node encoding", i);
    if (G.status == RTC_ERR_OVERFLOW)
      return fail(RTC_ERR_OVERFLOW, "root %u: its candidate BVH is %u levels deep, the kernel's traversal stack holds %d", i, G.depth,
                  RTC_TRAV_STACK);
#endif
    const uint32_t node_off = static_cast<uint32_t>(bvh_nodes.size()), leaf_off = static_cast<uint32_t>(bvh_leaves.size());
    auto moved2 = [&](uint32_t ref) -> uint32_t {  // a child reference of the binary tree, local -> joined
      if (ref == RTC_NO_LEAF) return ref;
      if (ref & RTC_NODE_BIT) return RTC_NODE_BIT | ((((ref & ~RTC_NODE_BIT) >> 3) + leaf_off) << 3) | (ref & 7u);
      return ref + node_off;
    };
    for (BvhNode N : G.nodes) {
      N.c0 = moved2(N.c0);
      N.c1 = moved2(N.c1);
      bvh_nodes.push_back(N);
    }
    bvh_leaves.insert(bvh_leaves.end(), G.leaves.begin(), G.leaves.end());
    bvh_mag = std::fmax(bvh_mag, G.mag);
    bvh2_root_of[i] = bvh_root_of[i] = G.root2 + node_off;
#if RTC_BVH8
    const uint32_t node8_off = static_cast<uint32_t>(T.bvh8_nodes.size()), leaf8_off = static_cast<uint32_t>(T.bvh8_leaves.size());
    if (static_cast<uint64_t>(leaf8_off) + G.leaves8.size() >= (1u << 24))
      return fail(RTC_ERR_UNSUPPORTED, "%zu leaves inside groups exceed the eight-wide BVH's leaf addressing", leaf8_off + G.leaves8.size());
    for (Bvh8Node N : G.nodes8) {
      N.child_base += node8_off;
      N.leaf_base_lmask = (((N.leaf_base_lmask & 0xFFFFFFu) + leaf8_off) & 0xFFFFFFu) | (N.leaf_base_lmask & 0xFF000000u);
      T.bvh8_nodes.push_back(N);
    }
    T.bvh8_leaves.insert(T.bvh8_leaves.end(), G.leaves8.begin(), G.leaves8.end());
    bvh_root_of[i] = G.root8 + node8_off;
    T.root_always[i] = uint2{leaf8_off + G.n_tree_leaves8, G.n_unbounded};
#else
    uint32_t stack_need = 0;
    bvh_root_of[i] = Bvh4Collapse{bvh_nodes, bvh4_nodes}.convert(bvh_root_of[i], stack_need);
    if (stack_need + 1 > RTC_TRAV_STACK)
      return fail(RTC_ERR_OVERFLOW, "root %u: walking its candidate BVH can take %u stack entries, the kernel's traversal stack holds %d", i,
                  stack_need + 1, RTC_TRAV_STACK);
#endif
    G = GroupBuild{};  // (the group's own tables are not needed any more)
  }
  if (rtcOptions().bvh_check != 0.0) {
    // diagnostic: every leaf once, every stored child box contains the world boxes below it
    std::vector<Aabb> world(n_live);
    for (uint32_t c = 0; c < d.n_leaves; ++c)
      if (dfs_of[c] != RTC_NO_LEAF) world[dfs_of[c]] = leafWorldBox(d, c);
    std::vector<uint32_t> seen(n_live, 0);
    size_t bad = 0;
    for (uint32_t i = 0; i < d.n_roots; ++i) {
      if (!(d.roots[i] & RTC_CHILD_NODE_BIT)) continue;
      struct Item { uint32_t ref; const float* lo; const float* hi; };
      std::vector<Item> todo;
      const BvhNode& R0 = bvh_nodes[bvh2_root_of[i]];  // (the binary tree: the four-wide one copies its boxes)
      todo.push_back({R0.c0, R0.lo0, R0.hi0});
      todo.push_back({R0.c1, R0.lo1, R0.hi1});
      while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        if (it.ref == RTC_NO_LEAF) continue;
        // collect leaves below it.ref
        std::vector<uint32_t> st{it.ref};
        while (!st.empty()) {
          const uint32_t r = st.back();
          st.pop_back();
          if (r == RTC_NO_LEAF) continue;
          if (r & RTC_NODE_BIT) {
            const uint32_t first = (r & ~RTC_NODE_BIT) >> 3, count = (r & 7u) + 1u;
            for (uint32_t k = 0; k < count; ++k) {
              const uint32_t leaf = bvh_leaves[first + k];
              if (leaf & RTC_NODE_BIT) continue;  // a csg unit
              const Aabb& w = world[leaf];
              if (w.finite())
                for (int a = 0; a < 3; ++a)
                  if (w.lo[a] < it.lo[a] || w.hi[a] > it.hi[a]) {
                    if (bad++ < 5)
                      std::fprintf(stderr, "rtc bvh check: leaf %u axis %d [%g,%g] outside stored box [%g,%g] (ref %08x)\n", leaf, a,
                                   w.lo[a], w.hi[a], it.lo[a], it.hi[a], it.ref);
                  }
              if (!w.finite() && (it.lo[0] > -1e38f || it.hi[0] < 1e38f) && bad++ < 5)
                std::fprintf(stderr, "rtc bvh check: unbounded leaf %u under a bounded box\n", leaf);
            }
          } else {
            st.push_back(bvh_nodes[r].c0);
            st.push_back(bvh_nodes[r].c1);
          }
        }
        if (!(it.ref & RTC_NODE_BIT)) {
          const BvhNode& N = bvh_nodes[it.ref];
          todo.push_back({N.c0, N.lo0, N.hi0});
          todo.push_back({N.c1, N.lo1, N.hi1});
        } else {
          const uint32_t first = (it.ref & ~RTC_NODE_BIT) >> 3, count = (it.ref & 7u) + 1u;
          for (uint32_t k = 0; k < count; ++k)
            if (!(bvh_leaves[first + k] & RTC_NODE_BIT)) seen[bvh_leaves[first + k]]++;
        }
      }
    }
    size_t in_groups = 0, once = 0;
    for (uint32_t l = 0; l < n_live; ++l) {
      if (leaf_parent[l] != RTC_NO_LEAF) {
        in_groups++;
        once += seen[l] == 1;
      }
    }
    std::fprintf(stderr, "rtc bvh check: %zu nodes, %zu leaves in groups, %zu seen exactly once, %zu containment violations\n",
                 bvh_nodes.size(), in_groups, once, bad);
  }
  if (T.bvh8_leaves.size() >= (1u << 24)) return fail(RTC_ERR_UNSUPPORTED, "%zu leaves inside groups exceed the eight-wide BVH's leaf addressing", T.bvh8_leaves.size());
  if (bvh_leaves.size() >= (1u << 28)) return fail(RTC_ERR_UNSUPPORTED, "%zu leaves inside groups exceed the BVH leaf-range encoding", bvh_leaves.size());
  buildRootTables(d, dfs_of, bvh_root_of, T);
  plain_tables.join();
  // Group boxes are unions of their children's boxes (group.zig:23-37), so they nest - unless a csg kept a stale box
  // (shape.zig:298-302) or the caller's tables are not the reference's.  Checked, not assumed: the kernel's chain
  // replay stops at the innermost box only if they do.
  T.chain_nested = true;
  for (uint32_t n = 0; n < d.n_nodes && T.chain_nested; ++n) {
    const uint32_t p = node_parent[n];
    if (p == RTC_NO_LEAF) continue;
    for (int k = 0; k < 3; ++k)
      if (!(d.node_min[3ull * n + k] >= d.node_min[3ull * p + k] && d.node_max[3ull * n + k] <= d.node_max[3ull * p + k]))
        T.chain_nested = false;
  }

  return RTC_OK;
}

// Copies the tables into HBM and fills in the scene handle.
// What a handle owns by itself (a clone has its own): its stream, the launch counters, the event launches on other
// streams order themselves by.  Schedules, cost buffers, ray stacks follow on first use.
// Nothing is enqueued here and no stream is made: the counters are zeroed by the handle's FIRST launch, on the stream
// that launch comes on, and the handle's own stream exists from the first call that needs it (a NULL stream argument,
// rtc_render's host forms).  The HIP runtime deals its few hardware queues to streams in the order of their first use: a
// handle that used a stream of its own at creation took a queue away from the caller's streams, and two of a host's three
// streams of frames in flight could end up on ONE queue, their frames one after the other (the slowest 8-way share of
// cover, three in flight: 0.08 or 0.16 ms per frame depending on how many handles had been created before the streams
// were first used; profiles/r04/frames_in_flight_sweeps.txt).
int initLaunchState(rtc_scene* s) {
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_stats), 2 * sizeof(DevStats)));
  HIP_TRY(hipEventCreateWithFlags(&s->launch_done, hipEventDisableTiming));
  s->stats_zeroed = false;
  s->last_stream = nullptr;
  return RTC_OK;
}
hipError_t ensureOwnStream(rtc_scene* s) {
  return s->stream != nullptr ? hipSuccess : hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
}

int uploadTables(const rtc_scene_desc& d, const SceneTraits& traits, const HostTables& T, rtc_scene* s) {
  const auto& leaf_meta = T.leaf_meta;
  const auto& roots = T.roots;
  const auto& kids = T.kids;
  const auto& leaf_parent = T.leaf_parent;
  const auto& node_parent = T.node_parent;
  const auto& bvh4_nodes = T.bvh4_nodes;
  const auto& bvh_leaves = T.bvh_leaves;
  (void)bvh_leaves;
  const auto& node_info = T.node_info;
  const auto& node_range = T.node_range;
  const auto& root_recs = T.root_recs;
  const auto& root_cull = T.root_cull;
  const auto& xf = T.xf;
  const auto& pat = T.pat;
  const auto& cyl = T.cyl;
  const auto& tri = T.tri;
  const auto& trin = T.trin;
  const auto& mat = T.mat;
  const auto& node_kids = T.node_kids;
  const auto& node_box = T.node_box;
  const auto& light = T.light;
  const auto& n_live = T.n_live;
  const auto& bvh_mag = T.bvh_mag;
  const auto& cull_cmax = T.cull_cmax;
  HIP_TRY(hipGetDevice(&s->device));
  s->tab = std::make_shared<SceneTables>();
  s->tab->handles.fetch_add(1, std::memory_order_relaxed);
  HIP_TRY(s->tab->roots.upload(roots));
  HIP_TRY(s->tab->root_recs.upload(root_recs));
  std::vector<RootCullPair> root_cull_pairs(root_cull.size() / 2u);
  for (size_t i = 0; i < root_cull_pairs.size(); ++i) {
    const RootCull &a = root_cull[2 * i], &b = root_cull[2 * i + 1];
    root_cull_pairs[i].cx = {a.cx, b.cx};
    root_cull_pairs[i].cy = {a.cy, b.cy};
    root_cull_pairs[i].cz = {a.cz, b.cz};
    root_cull_pairs[i].r2 = {a.r2, b.r2};
  }
  HIP_TRY(s->tab->root_cull.upload(root_cull_pairs));
  std::vector<RootBoxPair> root_box_pairs(T.root_box.size() / 2u);
  for (size_t i = 0; i < root_box_pairs.size(); ++i) {
    const RootBox &a = T.root_box[2 * i], &b = T.root_box[2 * i + 1];
    RootBoxPair::Pair* const axis[3] = {root_box_pairs[i].x, root_box_pairs[i].y, root_box_pairs[i].z};
    for (int k = 0; k < 3; ++k) {
      axis[k][0] = axis[k][2] = {a.lo[k], b.lo[k]};
      axis[k][1] = {a.hi[k], b.hi[k]};
    }
    root_box_pairs[i].line_only = {a.line_only, b.line_only};
  }
  HIP_TRY(s->tab->root_box.upload(root_box_pairs));
  HIP_TRY(s->tab->root_weight.upload(T.root_weight));
  HIP_TRY(s->tab->kids.upload(kids));
  HIP_TRY(s->tab->leaf_meta.upload(leaf_meta));
  HIP_TRY(s->tab->xf.upload(xf));
  HIP_TRY(s->tab->cyl.upload(cyl));
  HIP_TRY(s->tab->tri.upload(tri));
  HIP_TRY(s->tab->trin.upload(trin));
  HIP_TRY(s->tab->mat.upload(mat));
  HIP_TRY(s->tab->pat.upload(pat));
  HIP_TRY(s->tab->node_box.upload(node_box));
  HIP_TRY(s->tab->node_kids.upload(node_kids));
  HIP_TRY(s->tab->bvh.upload(bvh4_nodes));
  HIP_TRY(s->tab->bvh8.upload(T.bvh8_nodes));
#if RTC_BVH8
  const std::vector<uint32_t>& walk_leaves = T.bvh8_leaves;
#else
  const std::vector<uint32_t>& walk_leaves = bvh_leaves;
#endif
  std::vector<BvhLeafRec> leaf_recs(walk_leaves.size());
  for (size_t i = 0; i < leaf_recs.size(); ++i) {
    BvhLeafRec& L = leaf_recs[i];
    std::memset(&L, 0, sizeof L);
    L.leaf = walk_leaves[i];
    if (L.leaf & RTC_NODE_BIT) continue;  // a csg unit
    const uint4 m = leaf_meta[L.leaf];
    L.kind_flags = m.x;
    L.xform = m.y;
    L.material = m.z;
    L.geom = m.w;
    L.parent = leaf_parent[L.leaf];
    const uint32_t kind = m.x & 0xFFu;
    if (kind == RTC_TRIANGLE || kind == RTC_SMOOTH_TRIANGLE) std::memcpy(L.tri, &tri[9ull * m.w], sizeof L.tri);
  }
  HIP_TRY(s->tab->bvh_leaf.upload(leaf_recs));
  HIP_TRY(s->tab->leaf_parent.upload(leaf_parent));
  HIP_TRY(s->tab->node_parent.upload(node_parent));
  {
    std::vector<DevTexMap> tex(d.n_texmaps);
    for (uint32_t i = 0; i < d.n_texmaps; ++i) {
      tex[i].mapping = d.tex_mapping[i];
      for (int f = 0; f < 6; ++f) tex[i].uv[f] = d.tex_uv[6ull * i + f];
      tex[i].pad_ = 0;
    }
    std::vector<DevUv> uv(d.n_uvs);
    for (uint32_t i = 0; i < d.n_uvs; ++i) {
      std::memset(&uv[i], 0, sizeof(DevUv));
      uv[i].width = d.uv_size[2ull * i];
      uv[i].height = d.uv_size[2ull * i + 1];
      uv[i].kind = d.uv_kind[i];
      uv[i].interp = d.uv_interp[i];
      uv[i].image = d.uv_image[i];
      for (int k = 0; k < 5; ++k) uv[i].sub[k] = d.uv_sub[5ull * i + k];
    }
    std::vector<DevImage> img(d.n_images);
    size_t total_px = 0;
    for (uint32_t i = 0; i < d.n_images; ++i) {
      img[i] = DevImage{d.img_offset[i], d.img_width[i], d.img_height[i]};
      total_px = std::max<size_t>(total_px, d.img_offset[i] + static_cast<size_t>(d.img_width[i]) * d.img_height[i]);
    }
    std::vector<float> rgb(d.img_rgb, d.img_rgb + 3 * total_px);
    HIP_TRY(s->tab->tex.upload(tex));
    HIP_TRY(s->tab->uv.upload(uv));
    HIP_TRY(s->tab->img.upload(img));
    HIP_TRY(s->tab->img_rgb.upload(rgb));
  }
  HIP_TRY(s->tab->node_info.upload(node_info));
  HIP_TRY(s->tab->node_range.upload(node_range));
  const bool has_csg = traits.has_csg, ext_kernel = traits.ext_kernel;
  s->has_csg = has_csg;
  s->ext_kernel = ext_kernel;
  // Worlds without groups (or csg: a csg is a node) that fit the LDS tables run kernels without the group traversal:
  // `flat` (every leaf kind), or `simple` when all leaves are spheres, planes or cubes; each with an `_ext` form when
  // texture maps are among the patterns.
  s->flat_kernel = d.n_nodes == 0 && d.n_roots <= RTC_LDS_ROOTS && d.n_materials <= RTC_LDS_MATERIALS &&
                   d.n_patterns <= RTC_LDS_PATTERNS && d.n_lights <= RTC_LDS_LIGHTS;
  s->simple_kernel = s->flat_kernel;
  for (uint32_t i = 0; i < d.n_roots && s->flat_kernel; ++i) {
    if (d.roots[i] & RTC_CHILD_NODE_BIT) s->flat_kernel = s->simple_kernel = false;
    else if (d.leaf_kind[d.roots[i]] > RTC_CUBE) s->simple_kernel = false;
  }
  {  // spheres against cubes among the top-level objects (planes have no bound either way)
    uint32_t n_spheres = 0, n_cubes = 0;
    for (uint32_t i = 0; i < d.n_roots; ++i) {
      if (d.roots[i] & RTC_CHILD_NODE_BIT) continue;
      n_spheres += d.leaf_kind[d.roots[i]] == RTC_SPHERE;
      n_cubes += d.leaf_kind[d.roots[i]] == RTC_CUBE;
    }
    s->box_cull = n_cubes > n_spheres;
    if (rtcOptions().box_cull >= 0.0) s->box_cull = rtcOptions().box_cull != 0.0;  // (tests and fuzzers reach either form on any world)
  }
  s->simple3_ok = s->simple_kernel && !ext_kernel && d.n_roots <= RTC_LDS3_ROOTS && d.n_materials <= RTC_LDS3_MATERIALS &&
                  d.n_patterns <= RTC_LDS3_PATTERNS && d.n_lights <= RTC_LDS3_LIGHTS;
  s->general3_ok = RTC_BVH8 && !s->flat_kernel && !ext_kernel && d.n_roots <= RTC_LDS3_ROOTS && d.n_materials <= RTC_LDS3_MATERIALS &&
                   d.n_patterns <= RTC_LDS3_PATTERNS && d.n_lights <= RTC_LDS3_LIGHTS;
  HIP_TRY(s->tab->light.upload(light));
  if (const int st = initLaunchState(s); st != RTC_OK) return st;
  s->max_trav_stack = traits.max_stack;
  {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, s->device));
    s->n_cus = static_cast<uint32_t>(prop.multiProcessorCount);
    int nb = 0;
    DevPixelMap no_map{};  // (n_chunks 0: the kernel of a small launch)
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &nb, ldsKernel(s, no_map).fn, 256, 0));
    s->blocks_per_cu_lds = static_cast<uint32_t>(std::max(nb, 1));
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rtc_render_kernel_simple3, 256, 0));
    s->blocks_per_cu_simple3 = static_cast<uint32_t>(std::max(nb, 1));
#if RTC_BVH8
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rtc_render_kernel3, 256, 0));
    s->blocks_per_cu_general3 = static_cast<uint32_t>(std::max(nb, 1));
#endif
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &nb, ext_kernel ? rtc_render_kernel_bigworld_ext : rtc_render_kernel_bigworld, 256, 0));
    s->blocks_per_cu_big = static_cast<uint32_t>(std::max(nb, 1));
    if (const int v = static_cast<int>(rtcOptions().blocks_per_cu); v >= 1)  // (tuning option)
      s->blocks_per_cu_lds = std::min<uint32_t>(s->blocks_per_cu_lds, v), s->blocks_per_cu_big = std::min<uint32_t>(s->blocks_per_cu_big, v);
  }
  DevScene& D = s->dev;
  D.root_recs = s->tab->root_recs.p;
  D.root_cull = s->tab->root_cull.p;
  D.root_box = s->tab->root_box.p;
  D.root_weight = s->tab->root_weight.p;
  D.roots = s->tab->roots.p;
  D.leaf_meta = s->tab->leaf_meta.p;
  D.xf = s->tab->xf.p;
  D.cyl = s->tab->cyl.p;
  D.tri = s->tab->tri.p;
  D.trin = s->tab->trin.p;
  D.mat = s->tab->mat.p;
  D.pat = s->tab->pat.p;
  D.bvh = s->tab->bvh.p;
  D.bvh8 = s->tab->bvh8.p;
  D.n_bvh_nodes = static_cast<uint32_t>(RTC_BVH8 ? T.bvh8_nodes.size() : bvh4_nodes.size());
  D.n_bvh_leaves = static_cast<uint32_t>(walk_leaves.size());
  D.bvh_leaf = s->tab->bvh_leaf.p;
  D.leaf_parent = s->tab->leaf_parent.p;
  D.node_parent = s->tab->node_parent.p;
  D.node_info = s->tab->node_info.p;
  D.tex = s->tab->tex.p;
  D.uv = s->tab->uv.p;
  D.img = s->tab->img.p;
  D.img_rgb = s->tab->img_rgb.p;
  D.node_range = s->tab->node_range.p;
  D.bvh_mag = bvh_mag;
  D.chain_nested = T.chain_nested ? 1u : 0u;
  D.all_solid = 1u;
  for (uint32_t i = 0; i < d.n_patterns; ++i)
    if (d.pat_kind[i] != RTC_PAT_SOLID) D.all_solid = 0u;
#ifdef RTC_NO_ALL_SOLID
  D.all_solid = 0u;
#endif
  D.csg_entries = RTC_CSG_ENTRIES;
  D.node_box = s->tab->node_box.p;
  D.node_kids = s->tab->node_kids.p;
  D.kids = s->tab->kids.p;
  D.light = s->tab->light.p;
  D.n_roots = d.n_roots;
  D.n_leaves = n_live;
  D.n_nodes = d.n_nodes;
  D.n_lights = d.n_lights;
  D.n_materials = d.n_materials;
  D.n_patterns = d.n_patterns;
  D.cull_cmax = cull_cmax;
  D.cull_bmax = T.cull_bmax;
  D.cull_par = T.cull_par;
  D.n_root_spheres = T.n_root_kind[0];
  D.n_root_planes = T.n_root_kind[1];
  D.n_root_cubes = T.n_root_kind[2];
  return RTC_OK;
}

}  // namespace

extern "C" {

// The HIP runtime sets its device-to-host path for pageable memory up at the first such copy of 64 KB or more on a
// device: 6.8 ms, whatever the size (profiles/r04/first_copy_probe.txt: a 4 KB copy does not trigger it, a 64 KB copy
// pays all of it).  The first handle created on a device makes that copy - into a buffer of its own, from the ray
// levels just allocated, whose contents do not matter - so that the set-up is part of creating a scene (next to the
// runtime's own start-up, a hundred times longer) and not of the first rtc_render into a host canvas.
static void warmHostCopies(const rtc_scene* s) {
  static std::atomic<uint64_t> warmed{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !s->d_ray_stack || s->ray_stack_capacity < 65536u) {
    (void)hipGetLastError();
    return;
  }
  if (warmed.fetch_or(uint64_t{1} << dev) & (uint64_t{1} << dev)) return;
  std::vector<char> sink(65536, 0);
  if (hipMemcpy(sink.data(), s->d_ray_stack, sink.size(), hipMemcpyDeviceToHost) != hipSuccess) (void)hipGetLastError();
}

int rtc_scene_create(const rtc_scene_desc* desc, rtc_scene** out) {
  g_error.clear();
  if (!desc || !out) return fail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  const rtc_scene_desc& d = *desc;
  if (d.abi_version != RTC_ABI_VERSION)
    return fail(RTC_ERR_INVALID_ARGUMENT, "abi_version %u, library speaks %u", d.abi_version, RTC_ABI_VERSION);

  SceneTraits traits;
  if (const int st = validateScene(d, traits); st != RTC_OK) return st;
  HostTables tables;
  if (const int st = buildTables(d, traits, tables); st != RTC_OK) return st;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
    return fail(RTC_ERR_NO_DEVICE, "no HIP device is visible; this library has no CPU path");
  auto s = new (std::nothrow) rtc_scene();
  if (!s) return fail(RTC_ERR_OUT_OF_MEMORY, "host allocation");
  struct Guard {
    rtc_scene* s;
    ~Guard() {
      if (s) rtc_scene_destroy(s);
    }
  } guard{s};
  if (const int st = uploadTables(d, traits, tables, s); st != RTC_OK) return st;
  {
    // The pending-ray levels of a full-size launch at the reference's depth (camera.zig:118), allocated here with the
    // rest of the scene: the first frame - the only one a host that renders a scene once ever sees - then starts
    // without a 59 MB hipMalloc in front of it.  Another depth or launch size re-allocates at its first launch.
    DevPixelMap full{};
    full.n_chunks = 1u << 20;
    const size_t need = static_cast<size_t>(residentBlocks(s, full)) * 4u * (5u + 2u) * 64u * 64u;
    if (hipMalloc(&s->d_ray_stack, need) == hipSuccess) {
      s->ray_stack_capacity = need;
    } else {
      (void)hipGetLastError();  // (not fatal here: the first launch asks again and reports)
      s->d_ray_stack = nullptr;
    }
  }
  warmHostCopies(s);
  guard.s = nullptr;
  *out = s;
  return RTC_OK;
}

int rtc_scene_clone(const rtc_scene* src, rtc_scene** out) {
  g_error.clear();
  if (!src || !out) return fail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  auto s = new (std::nothrow) rtc_scene();
  if (!s) return fail(RTC_ERR_OUT_OF_MEMORY, "host allocation");
  struct Guard {
    rtc_scene* s;
    ~Guard() {
      if (s) rtc_scene_destroy(s);
    }
  } guard{s};
  // the scene and what was derived from it; nothing of the source's launches (schedule, measurements, buffers)
  s->device = src->device;
  s->tab = src->tab;
  s->tab->handles.fetch_add(1, std::memory_order_relaxed);
  s->dev = src->dev;
  s->dev.csg_buf = nullptr;
  s->has_csg = src->has_csg;
  s->ext_kernel = src->ext_kernel;
  s->simple_kernel = src->simple_kernel;
  s->flat_kernel = src->flat_kernel;
  s->simple3_ok = src->simple3_ok;
  s->box_cull = src->box_cull;
  s->max_trav_stack = src->max_trav_stack;
  s->n_cus = src->n_cus;
  s->blocks_per_cu_lds = src->blocks_per_cu_lds;
  s->blocks_per_cu_big = src->blocks_per_cu_big;
  s->blocks_per_cu_simple3 = src->blocks_per_cu_simple3;
  s->general3_ok = src->general3_ok;
  s->blocks_per_cu_general3 = src->blocks_per_cu_general3;
  s->use_three_waves = src->use_three_waves;  // (a clone starts from what its source has measured; frames in flight run no trial of their own)
  HIP_TRY(hipSetDevice(s->device));
  if (const int st = initLaunchState(s); st != RTC_OK) return st;
  guard.s = nullptr;
  *out = s;
  return RTC_OK;
}

void rtc_scene_destroy(rtc_scene* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->copy_stream) {
    (void)hipStreamSynchronize(s->copy_stream);
    (void)hipStreamDestroy(s->copy_stream);
  }
  for (rtc_scene* b : s->band) rtc_scene_destroy(b);
  s->band.clear();
  for (hipEvent_t& e : s->band_done)
    if (e) (void)hipEventDestroy(e);
  (void)hipSetDevice(s->device);
  if (s->launch_done) (void)hipEventSynchronize(s->launch_done);  // the last launch, whatever stream it ran on
  if (s->stream) {
    (void)hipStreamSynchronize(s->stream);
    (void)hipStreamDestroy(s->stream);
  }
  for (auto& pair : s->tune.ev)
    for (hipEvent_t& e : pair)
      if (e) (void)hipEventDestroy(e);
  if (s->d_stats) (void)hipFree(s->d_stats);
  if (s->d_frame) (void)hipFree(s->d_frame);
  for (int b = 0; b < 2; ++b)
    if (s->d_sched[b]) (void)hipFree(s->d_sched[b]);
  if (s->d_sched_info) (void)hipFree(s->d_sched_info);
  if (s->d_pack_state) (void)hipFree(s->d_pack_state);
  if (s->d_tile_list) (void)hipFree(s->d_tile_list);
  if (s->d_chunk_time) (void)hipFree(s->d_chunk_time);
  if (s->d_chunk_shape) (void)hipFree(s->d_chunk_shape);
  if (s->d_sorted) (void)hipFree(s->d_sorted);
  if (s->d_cost) (void)hipFree(s->d_cost);
  if (s->d_chunk_cost) (void)hipFree(s->d_chunk_cost);
  if (s->d_packet_time) (void)hipFree(s->d_packet_time);
  if (s->launch_done) (void)hipEventDestroy(s->launch_done);
  if (s->d_ray_stack) (void)hipFree(s->d_ray_stack);
  if (s->d_csg_buf) (void)hipFree(s->d_csg_buf);
  if (s->tab && !s->is_band) s->tab->handles.fetch_sub(1, std::memory_order_relaxed);
  delete s;
}

int rtc_render_device(rtc_scene* s, const rtc_camera* cam, uint32_t max_depth, uint32_t x0, uint32_t y0, uint32_t w,
                      uint32_t h, double* d_rgb_out, void* hip_stream) {
  g_error.clear();
  if (!s || !d_rgb_out) return fail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  int st = checkCamera(cam);
  if (st != RTC_OK) return st;
  DevPixelMap map;
  st = buildPixelMapRect(*cam, x0, y0, w, h, map);
  if (st != RTC_OK) return st;
  if (!hip_stream) HIP_TRY(ensureOwnStream(s));
  return launch(s, *cam, map, max_depth, d_rgb_out, static_cast<size_t>(w) * h,
                hip_stream ? static_cast<hipStream_t>(hip_stream) : s->stream);
}

int rtc_render_tiles_device(rtc_scene* s, const rtc_camera* cam, uint32_t max_depth, uint32_t tile_w, uint32_t tile_h,
                            uint32_t first_tile, uint32_t tile_stride, uint32_t n_my_tiles, double* d_rgb_out,
                            void* hip_stream) {
  g_error.clear();
  if (!s || !d_rgb_out) return fail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  int st = checkCamera(cam);
  if (st != RTC_OK) return st;
  if (tile_w == 0 || tile_h == 0 || tile_stride == 0 || n_my_tiles == 0)
    return fail(RTC_ERR_INVALID_ARGUMENT, "tile %ux%u stride %u count %u", tile_w, tile_h, tile_stride, n_my_tiles);
  DevPixelMap map;
  std::memset(&map, 0, sizeof map);
  map.mode = 1;
  map.tile_w = tile_w;
  map.tile_h = tile_h;
  map.first_tile = first_tile;
  map.tile_stride = tile_stride;
  map.n_my_tiles = n_my_tiles;
  map.tiles_x = (cam->hsize + tile_w - 1) / tile_w;
  const uint32_t tiles_y = (cam->vsize + tile_h - 1) / tile_h;
  const uint64_t last = static_cast<uint64_t>(first_tile) + static_cast<uint64_t>(n_my_tiles - 1) * tile_stride;
  if (last >= static_cast<uint64_t>(map.tiles_x) * tiles_y)
    return fail(RTC_ERR_INVALID_ARGUMENT, "tile %llu outside the %ux%u tiling", (unsigned long long)last, map.tiles_x, tiles_y);
  map.chunks_x = (tile_w + 7) / 8;
  map.chunks_per_region = map.chunks_x * ((tile_h + 7) / 8);
  const uint64_t total_chunks = static_cast<uint64_t>(map.chunks_per_region) * n_my_tiles;
  if (total_chunks > 0x7FFFFFFFull) return fail(RTC_ERR_INVALID_ARGUMENT, "%llu chunks", (unsigned long long)total_chunks);
  map.n_chunks = static_cast<uint32_t>(total_chunks);
  if (!hip_stream) HIP_TRY(ensureOwnStream(s));
  return launch(s, *cam, map, max_depth, d_rgb_out, static_cast<size_t>(n_my_tiles) * tile_w * tile_h,
                hip_stream ? static_cast<hipStream_t>(hip_stream) : s->stream);
}

int rtc_render_tile_list_device(rtc_scene* s, const rtc_camera* cam, uint32_t max_depth, uint32_t tile_w, uint32_t tile_h,
                                const uint32_t* tiles, uint32_t n_my_tiles, double* d_rgb_out, void* hip_stream) {
  g_error.clear();
  if (!s || !d_rgb_out || !tiles) return fail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  int st = checkCamera(cam);
  if (st != RTC_OK) return st;
  if (tile_w == 0 || tile_h == 0 || n_my_tiles == 0)
    return fail(RTC_ERR_INVALID_ARGUMENT, "tile %ux%u count %u", tile_w, tile_h, n_my_tiles);
  DevPixelMap map;
  std::memset(&map, 0, sizeof map);
  map.mode = 2;
  map.tile_w = tile_w;
  map.tile_h = tile_h;
  map.n_my_tiles = n_my_tiles;
  map.tiles_x = (cam->hsize + tile_w - 1) / tile_w;
  const uint64_t n_tiles = static_cast<uint64_t>(map.tiles_x) * ((cam->vsize + tile_h - 1) / tile_h);
  {
    std::vector<uint8_t> seen(n_tiles, 0);
    for (uint32_t i = 0; i < n_my_tiles; ++i) {
      if (tiles[i] >= n_tiles) return fail(RTC_ERR_INVALID_ARGUMENT, "tile %u outside the %llu tiles of the image", tiles[i], (unsigned long long)n_tiles);
      if (seen[tiles[i]]++) return fail(RTC_ERR_INVALID_ARGUMENT, "tile %u appears twice in the list", tiles[i]);
    }
  }
  HIP_TRY(hipSetDevice(s->device));
  if (!hip_stream) HIP_TRY(ensureOwnStream(s));
  hipStream_t stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : s->stream;
  if (s->h_tile_list.size() != n_my_tiles || std::memcmp(s->h_tile_list.data(), tiles, n_my_tiles * sizeof(uint32_t)) != 0) {
    if (n_my_tiles > s->tile_list_capacity) {
      HIP_TRY(handleIdle(s));
      if (s->d_tile_list) (void)hipFree(s->d_tile_list);
      s->d_tile_list = nullptr;
      s->tile_list_capacity = 0;
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_tile_list), n_my_tiles * sizeof(uint32_t)));
      s->tile_list_capacity = n_my_tiles;
    }
    // (launches of the old list may still be running on another stream: the copy is ordered like a launch)
    if (s->last_stream != nullptr && stream != s->last_stream) HIP_TRY(hipStreamWaitEvent(stream, s->launch_done, 0));
    s->h_tile_list.assign(tiles, tiles + n_my_tiles);
    HIP_TRY(hipMemcpyAsync(s->d_tile_list, s->h_tile_list.data(), n_my_tiles * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(s->launch_done, stream));
    s->last_stream = stream;
    s->tile_list_gen++;
  }
  map.first_tile = s->tile_list_gen;  // a new list is a new pixel map: its schedule is measured afresh
  map.tile_list = s->d_tile_list;
  map.chunks_x = (tile_w + 7) / 8;
  map.chunks_per_region = map.chunks_x * ((tile_h + 7) / 8);
  const uint64_t total_chunks = static_cast<uint64_t>(map.chunks_per_region) * n_my_tiles;
  if (total_chunks > 0x7FFFFFFFull) return fail(RTC_ERR_INVALID_ARGUMENT, "%llu chunks", (unsigned long long)total_chunks);
  map.n_chunks = static_cast<uint32_t>(total_chunks);
  return launch(s, *cam, map, max_depth, d_rgb_out, static_cast<size_t>(n_my_tiles) * tile_w * tile_h, stream);
}

int rtc_get_tile_costs(rtc_scene* s, double* cost_out, uint32_t n_regions) {
  g_error.clear();
  if (!s || !cost_out) return fail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  if (s->measured_regions == 0 || n_regions != s->measured_regions || !s->d_chunk_time)
    return fail(RTC_ERR_INVALID_ARGUMENT, "no measured launch with %u regions on this handle (the last one had %u)", n_regions,
                s->measured_regions);
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(handleIdle(s));
  const size_t n = static_cast<size_t>(s->measured_regions) * s->measured_chunks_per_region;
  std::vector<uint32_t> t(n);
  HIP_TRY(hipMemcpy(t.data(), s->d_chunk_time, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  for (uint32_t r = 0; r < n_regions; ++r) {
    double sum = 0.0;
    for (uint32_t c = 0; c < s->measured_chunks_per_region; ++c) sum += t[static_cast<size_t>(r) * s->measured_chunks_per_region + c];
    cost_out[r] = sum;
  }
  return RTC_OK;
}

int rtc_assign_tiles(const double* tile_cost, uint32_t n_tiles, uint32_t world, uint32_t* rank_of_tile, uint32_t* slot_of_tile) {
  g_error.clear();
  if (!tile_cost || !rank_of_tile || !slot_of_tile || n_tiles == 0 || world == 0) return fail(RTC_ERR_INVALID_ARGUMENT, "null or empty argument");
  // Longest job first onto the rank with the least work so far - among the ranks that still have a free slot: every
  // rank's buffer holds ceil(n_tiles / world) tiles (one equal-count gather).  Ties go to the lower tile / lower rank,
  // so every rank that computes this from the same costs gets the same table.
  const uint32_t padded = (n_tiles + world - 1u) / world;
  std::vector<uint32_t> order(n_tiles);
  for (uint32_t t = 0; t < n_tiles; ++t) order[t] = t;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return tile_cost[a] > tile_cost[b]; });
  std::vector<double> load(world, 0.0);
  std::vector<uint32_t> count(world, 0u);
  for (uint32_t t : order) {
    uint32_t best = world;
    for (uint32_t r = 0; r < world; ++r)
      if (count[r] < padded && (best == world || load[r] < load[best])) best = r;
    rank_of_tile[t] = best;
    load[best] += tile_cost[t] > 0.0 ? tile_cost[t] : 0.0;
    count[best]++;
  }
  // slots in tile order inside a rank (the rank renders its list in that order: image neighbours stay neighbours)
  std::fill(count.begin(), count.end(), 0u);
  for (uint32_t t = 0; t < n_tiles; ++t) slot_of_tile[t] = rank_of_tile[t] * padded + count[rank_of_tile[t]]++;
  return RTC_OK;
}

int rtc_assemble_tile_list_device(const double* d_gathered, const uint32_t* d_slot_of_tile, uint32_t tile_w, uint32_t tile_h,
                                  uint32_t hsize, uint32_t vsize, double* d_canvas, void* hip_stream) {
  g_error.clear();
  if (!d_gathered || !d_slot_of_tile || !d_canvas || !hip_stream) return fail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  if (tile_w == 0 || tile_h == 0 || hsize == 0 || vsize == 0)
    return fail(RTC_ERR_INVALID_ARGUMENT, "tile %ux%u image %ux%u", tile_w, tile_h, hsize, vsize);
  const size_t n = static_cast<size_t>(hsize) * vsize * 3u;
  const uint32_t blocks = static_cast<uint32_t>(std::min<size_t>((n + 255) / 256, 256u * 64u));
  hipLaunchKernelGGL(rtc_assemble_list_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), d_gathered,
                     d_slot_of_tile, tile_w, tile_h, hsize, vsize, d_canvas);
  HIP_TRY(hipGetLastError());
  return RTC_OK;
}

int rtc_assemble_tile_list_rgba8_device(const uint32_t* d_gathered_rgba, const uint32_t* d_slot_of_tile, uint32_t tile_w,
                                        uint32_t tile_h, uint32_t hsize, uint32_t vsize, uint32_t* d_rgba, void* hip_stream) {
  g_error.clear();
  if (!d_gathered_rgba || !d_slot_of_tile || !d_rgba || !hip_stream) return fail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  if (tile_w == 0 || tile_h == 0 || hsize == 0 || vsize == 0)
    return fail(RTC_ERR_INVALID_ARGUMENT, "tile %ux%u image %ux%u", tile_w, tile_h, hsize, vsize);
  const size_t n = static_cast<size_t>(hsize) * vsize;
  const uint32_t blocks = static_cast<uint32_t>(std::min<size_t>((n + 255) / 256, 256u * 64u));
  hipLaunchKernelGGL(rtc_assemble_list_rgba8_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                     d_gathered_rgba, d_slot_of_tile, tile_w, tile_h, hsize, vsize, d_rgba);
  HIP_TRY(hipGetLastError());
  return RTC_OK;
}

namespace {

// The frame of a host-output render: device staging for w x h pixels.
int ensureFrame(rtc_scene* s, size_t doubles) {
  if (doubles <= s->frame_capacity) return RTC_OK;
  HIP_TRY(handleIdle(s));
  if (s->d_frame) (void)hipFree(s->d_frame);
  s->d_frame = nullptr;
  s->frame_capacity = 0;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_frame), std::max<size_t>(doubles, 1) * sizeof(double)));
  s->frame_capacity = doubles;
  return RTC_OK;
}

// A lane that ran out of traversal stack, pending-ray stack or csg list space has dropped work: the image is not the
// reference's.  Say so instead of returning it (asynchronous callers check rtc_get_stats).
int checkOverflow(rtc_scene* s) {
  // (overflow and csg_needed share the cache line of `overflow`: one copy)
  struct {
    unsigned long long dropped;
    unsigned int csg_needed;
  } h{0, 0};
  const DevStats* st = s->d_stats + s->stats_parity;
  static_assert(offsetof(DevStats, csg_needed) == offsetof(DevStats, overflow) + sizeof(unsigned long long), "one copy reads both");
  HIP_TRY(hipMemcpy(&h, &st->overflow, sizeof(unsigned long long) + sizeof(unsigned int), hipMemcpyDeviceToHost));
  s->csg_needed = h.csg_needed;
  if (h.dropped) {
    if (h.csg_needed)
      return fail(RTC_ERR_OVERFLOW, "%llu lanes ran out of csg intersection list (%u entries; a list of up to %u was needed)", h.dropped,
                  s->dev.csg_entries, h.csg_needed);
    return fail(RTC_ERR_OVERFLOW, "%llu lanes overflowed a per-lane stack", h.dropped);
  }
  return RTC_OK;
}

// Csg.filterIntersections works on a list of any length (csg.zig:51-95); a lane's list here has DevScene::csg_entries
// slots in HBM.  A frame whose lists ran out says how long the longest one had to be (DevStats::csg_needed, read by
// checkOverflow): the lists are sized for that in ONE step (the next power of two, up to RTC_CSG_ENTRIES_MAX) and the
// handle keeps them.  Only a csg list's overflow is answered this way - a traversal or pending-ray stack that ran out is
// not fixed by longer lists (those are sized or refused at create).  Returns true if another attempt is worth it.
bool growCsgLists(rtc_scene* s) {
  if (!s->has_csg || s->csg_needed <= s->dev.csg_entries || s->dev.csg_entries >= RTC_CSG_ENTRIES_MAX) return false;
  uint32_t want = s->dev.csg_entries;
  while (want < s->csg_needed && want < RTC_CSG_ENTRIES_MAX) want *= 2u;
  s->dev.csg_entries = want;
  s->csg_needed = 0;
  if (s->tab) {  // (every handle of the scene follows at its next launch: ensureScratch)
    uint32_t seen = s->tab->csg_entries_wanted.load(std::memory_order_relaxed);
    while (seen < want && !s->tab->csg_entries_wanted.compare_exchange_weak(seen, want, std::memory_order_relaxed)) {}
  }
  g_error.clear();
  return true;
}

}  // namespace

int rtc_grow_csg_lists(rtc_scene* s) {
  g_error.clear();
  if (!s) return fail(RTC_ERR_INVALID_ARGUMENT, "null scene");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(handleIdle(s));
  if (!s->stats_zeroed) return RTC_OK;  // (no launch yet: the counters have not even been cleared - nothing overflowed)
  const int ov = checkOverflow(s);
  if (ov != RTC_ERR_OVERFLOW) return ov;  // (RTC_OK: nothing overflowed, nothing to do)
  return growCsgLists(s) ? RTC_OK : ov;
}

int rtc_canvas_register(void* canvas, size_t bytes) {
  g_error.clear();
  if (!canvas || bytes == 0) return fail(RTC_ERR_INVALID_ARGUMENT, "null canvas or no bytes");
  HIP_TRY(hipHostRegister(canvas, bytes, hipHostRegisterPortable | hipHostRegisterMapped));  // (mapped: every GPU of the node may write it)
  return RTC_OK;
}

int rtc_canvas_unregister(void* canvas) {
  g_error.clear();
  if (!canvas) return fail(RTC_ERR_INVALID_ARGUMENT, "null canvas");
  HIP_TRY(hipHostUnregister(canvas));
  return RTC_OK;
}

int rtc_diag_build_tables(const rtc_scene_desc* desc, uint64_t* digest, double* build_ms) {
  g_error.clear();
  if (!desc) return fail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  SceneTraits traits;
  if (const int st = validateScene(*desc, traits); st != RTC_OK) return st;
  HostTables tables;
  const auto t0 = std::chrono::steady_clock::now();
  if (const int st = buildTables(*desc, traits, tables); st != RTC_OK) return st;
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (build_ms) *build_ms = ms;
  if (digest) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t bytes) {
      const unsigned char* b = static_cast<const unsigned char*>(p);
      for (size_t i = 0; i < bytes; ++i) h = (h ^ b[i]) * 1099511628211ull;
    };
    auto vec = [&](const auto& v) {
      const uint64_t n = v.size();
      mix(&n, sizeof n);
      if (!v.empty()) mix(v.data(), v.size() * sizeof(v[0]));
    };
    vec(tables.bvh8_nodes);
    vec(tables.bvh8_leaves);
    vec(tables.bvh_nodes);
    vec(tables.bvh_leaves);
    vec(tables.root_recs);
    vec(tables.root_cull);
    vec(tables.root_box);
    vec(tables.root_always);
    vec(tables.leaf_meta);
    vec(tables.leaf_parent);
    vec(tables.node_parent);
    mix(&tables.bvh_mag, sizeof tables.bvh_mag);
    *digest = h;
  }
  return RTC_OK;
}

int rtc_diag_root_boxes(const rtc_scene_desc* desc, float* boxes, uint32_t* world_index, uint32_t capacity, uint32_t* n_roots,
                        float* scales) {
  g_error.clear();
  if (!desc || !n_roots) return fail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  SceneTraits traits;
  if (const int st = validateScene(*desc, traits); st != RTC_OK) return st;
  HostTables tables;
  if (const int st = buildTables(*desc, traits, tables); st != RTC_OK) return st;
  *n_roots = desc->n_roots;
  if (scales) {
    scales[0] = tables.cull_bmax;
    scales[1] = tables.cull_par;
  }
  if (capacity < desc->n_roots) return (boxes || world_index) ? fail(RTC_ERR_INVALID_ARGUMENT, "%u roots, room for %u", desc->n_roots, capacity) : RTC_OK;
  for (uint32_t i = 0; i < desc->n_roots; ++i) {
    if (boxes) {
      const RootBox& B = tables.root_box[i];
      for (int k = 0; k < 3; ++k) {
        boxes[7ull * i + k] = B.lo[k];
        boxes[7ull * i + 3 + k] = B.hi[k];
      }
      boxes[7ull * i + 6] = B.line_only;
    }
    if (world_index) world_index[i] = tables.root_order[i];
  }
  return RTC_OK;
}

int rtc_set_option(const char* name, double value) {
  g_error.clear();
  if (!name) return fail(RTC_ERR_INVALID_ARGUMENT, "null option name");
  RtcOptions& o = rtcOptions();
  const struct {
    const char* name;
    RtcOption* slot;
  } table[] = {{"simple3_min_chunks", &o.simple3_min_chunks}, {"sched_off", &o.sched_off}, {"cut_above", &o.cut_above},
               {"pack_rounds", &o.pack_rounds}, {"pull_min_idle", &o.pull_min_idle}, {"blocks_per_cu", &o.blocks_per_cu},
               {"sched_tmin", &o.sched_tmin}, {"bvh_leaf", &o.bvh_leaf}, {"bvh_one_axis", &o.bvh_one_axis},
               {"bvh_check", &o.bvh_check}, {"host_bands", &o.host_bands}, {"waves3", &o.waves3},
               {"measure_every", &o.measure_every}, {"sched_mix", &o.sched_mix},
               {"inflight_chunks_per_wave", &o.inflight_chunks_per_wave}, {"build_threads", &o.build_threads}, {"box_cull", &o.box_cull}};
  for (const auto& e : table)
    if (std::strcmp(e.name, name) == 0) {
      e.slot->set(value);
      return RTC_OK;
    }
  return fail(RTC_ERR_INVALID_ARGUMENT, "unknown option '%s'", name);
}

namespace {

// The copy of a frame to the host takes as long as the render or longer (24 B per pixel over the link: 0.9 ms for
// 1080p, which renders in 0.5): a large frame is rendered in horizontal bands one after the other and every band is
// copied on a second stream while the next one renders.  cover 1080p 1.44 -> 1.09 ms per frame into a registered canvas
// with two bands (four: 1.11, eight: 1.45), dragons 4K 5.9 -> 4.2 with four (tools/banded_output_time.py).
uint32_t hostBands(uint32_t w, uint32_t h) {
  const int forced = static_cast<int>(rtcOptions().host_bands);
  const uint64_t pixels = static_cast<uint64_t>(w) * h;
  uint32_t bands = forced >= 1 ? static_cast<uint32_t>(forced) : (pixels >= 6000000ull ? 4u : (pixels >= 400000ull ? 2u : 1u));
  bands = std::min(bands, RTC_MAX_HOST_BANDS);
  while (bands > 1u && h / bands < 64u) --bands;
  return bands;
}

int ensureBands(rtc_scene* s, uint32_t bands) {
  while (s->band.size() + 1u < bands) {
    rtc_scene* clone = nullptr;
    if (const int st = rtc_scene_clone(s, &clone); st != RTC_OK) return st;
    clone->is_band = true;  // (bands run one after the other: not frames in flight)
    s->tab->handles.fetch_sub(1, std::memory_order_relaxed);
    s->band.push_back(clone);
  }
  if (!s->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
  for (uint32_t b = 0; b < bands; ++b)
    if (!s->band_done[b]) HIP_TRY(hipEventCreateWithFlags(&s->band_done[b], hipEventDisableTiming));
  return RTC_OK;
}

// A host canvas the process has never touched (a one-shot render: main.zig:92 renders every scene once into a fresh
// Canvas) costs more in page faults than the frame costs to render and copy: the copy of a 1080p frame into untouched
// pageable memory took 11 ms against 1.5 ms into the same pages afterwards.  The faults cannot be avoided - the pages are
// the caller's, and nothing is remembered about them - but they need not wait for the GPU, nor be taken one trap at a
// time by the runtime's copy: while the render kernels run, the caller's thread has the kernel populate the canvas's
// pages in one call (MADV_POPULATE_WRITE; where the running kernel lacks it, one read-modify-write per page - the
// contents stay).  Three sampled pages that are resident mean the canvas has been used before: nothing is done then.
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
void prefaultCanvas(void* canvas, size_t bytes) {
  const uintptr_t page = static_cast<uintptr_t>(sysconf(_SC_PAGESIZE));
  const uintptr_t lo = (reinterpret_cast<uintptr_t>(canvas) + page - 1u) & ~(page - 1u);
  const uintptr_t hi = (reinterpret_cast<uintptr_t>(canvas) + bytes) & ~(page - 1u);
  if (hi < lo + 64u * page) return;  // (a small canvas: not worth a system call)
  const uintptr_t probes[3] = {lo, lo + (((hi - lo) / 2u) & ~(page - 1u)), hi - page};
  bool resident = true;
  for (const uintptr_t q : probes) {
    unsigned char vec = 0;
    if (mincore(reinterpret_cast<void*>(q), page, &vec) != 0 || (vec & 1u) == 0u) resident = false;
  }
  if (resident) return;
  if (madvise(reinterpret_cast<void*>(lo), hi - lo, MADV_POPULATE_WRITE) == 0) return;
  for (uintptr_t q = lo; q < hi; q += page) {
    volatile unsigned char* b = reinterpret_cast<volatile unsigned char*>(q);
    *b = *b;
  }
}

int renderBanded(rtc_scene* s, const rtc_camera* cam, uint32_t max_depth, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                 double* rgb_out, uint32_t bands) {
  if (const int st = ensureBands(s, bands); st != RTC_OK) return st;
  HIP_TRY(ensureOwnStream(s));
  for (;;) {
    uint32_t row[RTC_MAX_HOST_BANDS + 1];
    for (uint32_t b = 0; b < bands; ++b) row[b] = static_cast<uint32_t>(static_cast<uint64_t>(h) * b / bands) & ~7u;  // (whole rows of chunks)
    row[bands] = h;
    for (uint32_t b = 0; b < bands; ++b) {  // the renders, one after the other on the handle's stream ...
      rtc_scene* const who = b == 0 ? s : s->band[b - 1];
      const int st = rtc_render_device(who, cam, max_depth, x0, y0 + row[b], w, row[b + 1] - row[b],
                                       s->d_frame + 3ull * w * row[b], s->stream);
      if (st != RTC_OK) return st;
      HIP_TRY(hipEventRecord(s->band_done[b], s->stream));
    }
    prefaultCanvas(rgb_out, 3ull * w * h * sizeof(double));  // (a first-use canvas: its page faults, under the renders)
    for (uint32_t b = 0; b < bands; ++b) {  // ... and behind each its copy, on the other
      HIP_TRY(hipStreamWaitEvent(s->copy_stream, s->band_done[b], 0));
      HIP_TRY(hipMemcpyAsync(rgb_out + 3ull * w * row[b], s->d_frame + 3ull * w * row[b],
                             3ull * w * (row[b + 1] - row[b]) * sizeof(double), hipMemcpyDeviceToHost, s->copy_stream));
    }
    HIP_TRY(hipStreamSynchronize(s->copy_stream));
    s->last_bands = bands;
    int ov = RTC_OK;
    for (uint32_t b = 0; b < bands && ov == RTC_OK; ++b) ov = checkOverflow(b == 0 ? s : s->band[b - 1]);
    if (ov != RTC_ERR_OVERFLOW) return ov;
    bool grown = growCsgLists(s);
    for (rtc_scene* o : s->band) grown = growCsgLists(o) || grown;
    if (!grown) return ov;
  }
}

}  // namespace

int rtc_render(rtc_scene* s, const rtc_camera* cam, uint32_t max_depth, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
               double* rgb_out) {
  g_error.clear();
  if (!s || !rgb_out) return fail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  const size_t need = 3ull * w * h;
  HIP_TRY(hipSetDevice(s->device));
  if (const int st = ensureFrame(s, need); st != RTC_OK) return st;
  HIP_TRY(ensureOwnStream(s));
  if (const uint32_t bands = hostBands(w, h); bands > 1u) return renderBanded(s, cam, max_depth, x0, y0, w, h, rgb_out, bands);
  for (;;) {
    const int st = rtc_render_device(s, cam, max_depth, x0, y0, w, h, s->d_frame, s->stream);
    if (st != RTC_OK) return st;
    // (into pageable memory the copy runs at a fifth of the link's rate - 5.6 ms for a 1080p frame; a host that renders
    // frame after frame registers its canvas once: rtc_canvas_register)
    const size_t bytes = need * sizeof(double);
    prefaultCanvas(rgb_out, bytes);  // (a first-use canvas: its page faults, while the kernel runs)
    HIP_TRY(hipMemcpyAsync(rgb_out, s->d_frame, bytes, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    const int ov = checkOverflow(s);
    if (ov != RTC_ERR_OVERFLOW || !growCsgLists(s)) return ov;
  }
}

int rtc_rgba8_device(const double* d_canvas, size_t n_pixels, uint32_t* d_rgba, void* hip_stream) {
  g_error.clear();
  if (!d_canvas || !d_rgba || !hip_stream) return fail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  if (n_pixels == 0 || n_pixels > (static_cast<size_t>(1) << 40)) return fail(RTC_ERR_INVALID_ARGUMENT, "%zu pixels", n_pixels);
  hipLaunchKernelGGL(rtc_rgba8_kernel, dim3(static_cast<uint32_t>((n_pixels + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(hip_stream), d_canvas, n_pixels, d_rgba);
  HIP_TRY(hipGetLastError());
  return RTC_OK;
}

int rtc_render_rgba8(rtc_scene* s, const rtc_camera* cam, uint32_t max_depth, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                     uint8_t* rgba_out) {
  g_error.clear();
  if (!s || !rgba_out) return fail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  const size_t n = static_cast<size_t>(w) * h;
  HIP_TRY(hipSetDevice(s->device));
  if (const int st = ensureFrame(s, 3 * n + (n + 1) / 2); st != RTC_OK) return st;  // the f64 frame, then n u32 behind it
  HIP_TRY(ensureOwnStream(s));
  for (;;) {
    const int st = rtc_render_device(s, cam, max_depth, x0, y0, w, h, s->d_frame, s->stream);
    if (st != RTC_OK) return st;
    uint32_t* d_rgba = reinterpret_cast<uint32_t*>(s->d_frame + 3 * n);
    if (const int st2 = rtc_rgba8_device(s->d_frame, n, d_rgba, s->stream); st2 != RTC_OK) return st2;
    prefaultCanvas(rgba_out, n * sizeof(uint32_t));
    HIP_TRY(hipMemcpyAsync(rgba_out, d_rgba, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    const int ov = checkOverflow(s);
    if (ov != RTC_ERR_OVERFLOW || !growCsgLists(s)) return ov;
  }
}

int rtc_assemble_tiles_device(const double* d_gathered, uint32_t world, uint32_t padded_tiles, uint32_t tile_w,
                              uint32_t tile_h, uint32_t hsize, uint32_t vsize, double* d_canvas, void* hip_stream) {
  g_error.clear();
  if (!d_gathered || !d_canvas || !hip_stream) return fail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  if (world == 0 || tile_w == 0 || tile_h == 0 || hsize == 0 || vsize == 0)
    return fail(RTC_ERR_INVALID_ARGUMENT, "world %u tile %ux%u image %ux%u", world, tile_w, tile_h, hsize, vsize);
  const uint64_t n_tiles = static_cast<uint64_t>((hsize + tile_w - 1) / tile_w) * ((vsize + tile_h - 1) / tile_h);
  if (static_cast<uint64_t>(padded_tiles) * world < n_tiles)
    return fail(RTC_ERR_INVALID_ARGUMENT, "%u ranks x %u tiles cannot hold the %llu tiles of the image", world, padded_tiles,
                (unsigned long long)n_tiles);
  const size_t n = static_cast<size_t>(hsize) * vsize * 3u;
  const uint32_t blocks = static_cast<uint32_t>(std::min<size_t>((n + 255) / 256, 256u * 64u));
  hipLaunchKernelGGL(rtc_assemble_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), d_gathered,
                     world, padded_tiles, tile_w, tile_h, hsize, vsize, d_canvas);
  HIP_TRY(hipGetLastError());
  return RTC_OK;
}

int rtc_scatter_tile_list_device(const double* d_tiles, const uint32_t* d_tile_list, uint32_t n_tiles, uint32_t tile_w,
                                 uint32_t tile_h, uint32_t hsize, uint32_t vsize, double* canvas, void* hip_stream) {
  g_error.clear();
  if (!d_tiles || !d_tile_list || !canvas || !hip_stream) return fail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  if (n_tiles == 0 || tile_w == 0 || tile_h == 0 || hsize == 0 || vsize == 0)
    return fail(RTC_ERR_INVALID_ARGUMENT, "%u tiles of %ux%u, image %ux%u", n_tiles, tile_w, tile_h, hsize, vsize);
  const size_t n = static_cast<size_t>(n_tiles) * tile_w * tile_h * 3u;
  const uint32_t blocks = static_cast<uint32_t>(std::min<size_t>((n + 255) / 256, 256u * 64u));
  hipLaunchKernelGGL(rtc_scatter_tiles_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), d_tiles, d_tile_list,
                     n_tiles, tile_w, tile_h, hsize, vsize, canvas);
  HIP_TRY(hipGetLastError());
  return RTC_OK;
}

int rtc_scatter_tile_list_rgba8_device(const double* d_tiles, const uint32_t* d_tile_list, uint32_t n_tiles, uint32_t tile_w,
                                       uint32_t tile_h, uint32_t hsize, uint32_t vsize, uint32_t* rgba, void* hip_stream) {
  g_error.clear();
  if (!d_tiles || !d_tile_list || !rgba || !hip_stream) return fail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  if (n_tiles == 0 || tile_w == 0 || tile_h == 0 || hsize == 0 || vsize == 0)
    return fail(RTC_ERR_INVALID_ARGUMENT, "%u tiles of %ux%u, image %ux%u", n_tiles, tile_w, tile_h, hsize, vsize);
  const size_t n = static_cast<size_t>(n_tiles) * tile_w * tile_h;
  const uint32_t blocks = static_cast<uint32_t>(std::min<size_t>((n + 255) / 256, 256u * 64u));
  hipLaunchKernelGGL(rtc_scatter_tiles_rgba8_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), d_tiles,
                     d_tile_list, n_tiles, tile_w, tile_h, hsize, vsize, rgba);
  HIP_TRY(hipGetLastError());
  return RTC_OK;
}

int rtc_scene_synchronize(rtc_scene* s) {
  g_error.clear();
  if (!s) return fail(RTC_ERR_INVALID_ARGUMENT, "null scene");
  HIP_TRY(hipSetDevice(s->device));
  if (s->stream != nullptr) HIP_TRY(hipStreamSynchronize(s->stream));  // (no stream of its own yet: nothing was enqueued on it)
  return RTC_OK;
}

const char* rtc_last_kernel_name(const rtc_scene* s) { return s != nullptr ? s->last_kernel : ""; }

int rtc_get_schedule(rtc_scene* s, uint32_t* items, size_t capacity_items, uint32_t* n_packets) {
  g_error.clear();
  if (!s || !n_packets) return fail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  *n_packets = 0;
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(handleIdle(s));
  if (!s->sched_valid || !s->d_sched_info) return RTC_OK;
  DevSchedInfo info{};
  HIP_TRY(hipMemcpy(&info, s->d_sched_info + s->sched_cur, sizeof info, hipMemcpyDeviceToHost));
  *n_packets = info.n_units;
  const size_t need = static_cast<size_t>(info.n_units) * RTC_PACKET_ITEMS;
  if (need > s->sched_capacity) return fail(RTC_ERR_OVERFLOW, "the schedule has %u packets, its buffer holds %zu", info.n_units, s->sched_capacity / RTC_PACKET_ITEMS);
  if (need > capacity_items || (need != 0 && !items))
    return fail(RTC_ERR_INVALID_ARGUMENT, "the schedule has %u packets of %u items", info.n_units, RTC_PACKET_ITEMS);
  HIP_TRY(hipMemcpy(items, s->d_sched[s->sched_cur], need * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return RTC_OK;
}

int rtc_get_chunk_times(rtc_scene* s, const rtc_camera* cam, uint32_t* estimated, uint32_t* measured, size_t capacity, uint32_t* n_chunks) {
  g_error.clear();
  if (!s || !cam || !n_chunks) return fail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  *n_chunks = 0;
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(handleIdle(s));
  if (s->cost_key.empty() || !s->d_chunk_time || s->pack_capacity == 0) return RTC_OK;  // nothing scheduled has run
  DevPixelMap map{};
  std::memcpy(&map, s->cost_key.data(), std::min(s->cost_key.size() * sizeof(uint32_t), sizeof map));
  const uint32_t n = map.n_chunks;
  *n_chunks = n;
  if (n > s->pack_capacity) return fail(RTC_ERR_INVALID_ARGUMENT, "the pixel map has %u chunks, the buffers %u", n, s->pack_capacity);
  if (!estimated && !measured) return RTC_OK;  // (a size query)
  if (n > capacity) return fail(RTC_ERR_INVALID_ARGUMENT, "%u chunks, room for %zu", n, capacity);
  if (measured) HIP_TRY(hipMemcpy(measured, s->d_chunk_time, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (estimated) {
    uint32_t *d_est = nullptr, *d_time = nullptr;
    DevChunkShape* d_shape = nullptr;
    DevPackState* d_state = nullptr;
    auto release = [&]() {
      for (void* p : {static_cast<void*>(d_est), static_cast<void*>(d_time), static_cast<void*>(d_shape), static_cast<void*>(d_state)})
        if (p) (void)hipFree(p);
    };
    if (hipMalloc(reinterpret_cast<void**>(&d_est), n * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&d_time), n * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&d_shape), n * sizeof(DevChunkShape)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&d_state), sizeof(DevPackState)) != hipSuccess) {
      (void)hipGetLastError();
      release();
      return fail(RTC_ERR_OUT_OF_MEMORY, "scratch for %u chunks", n);
    }
    if (ensureOwnStream(s) != hipSuccess) {
      (void)hipGetLastError();
      release();
      return fail(RTC_ERR_OUT_OF_MEMORY, "no stream");
    }
    hipLaunchKernelGGL(rtc_estimate_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s->stream, s->dev, devCamera(*cam), map, d_est, d_time,
                       d_shape, d_state);
    const hipError_t e1 = hipStreamSynchronize(s->stream);
    const hipError_t e2 = e1 == hipSuccess ? hipMemcpy(estimated, d_est, n * sizeof(uint32_t), hipMemcpyDeviceToHost) : e1;
    release();
    HIP_TRY(e2);
  }
  return RTC_OK;
}

int rtc_get_stats(rtc_scene* s, rtc_stats* out) {
  g_error.clear();
  if (!s || !out) return fail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(handleIdle(s));  // the last launch on this handle, whatever stream it ran on
  if (!s->stats_zeroed) {  // (no launch yet: the counters have not even been cleared)
    std::memset(out, 0, sizeof *out);
    return RTC_OK;
  }
  const std::unique_ptr<DevStats> hp(new DevStats);  // large in diagnostic layouts: off the stack, and not shared between handles
  DevStats& h = *hp;
  HIP_TRY(hipMemcpy(&h, s->d_stats + s->stats_parity, sizeof h, hipMemcpyDeviceToHost));
  out->primary = h.primary;
  out->secondary = h.secondary;
  out->shadow_calls = h.shadow_calls;
  out->shadow_traced = h.shadow_traced;
  out->overflow = h.overflow;
  for (uint32_t b = 1; b < s->last_bands; ++b) {  // (a frame rtc_render cut into bands: the other bands' counters)
    rtc_scene* const o = s->band[b - 1];
    HIP_TRY(handleIdle(o));
    HIP_TRY(hipMemcpy(&h, o->d_stats + o->stats_parity, sizeof h, hipMemcpyDeviceToHost));
    out->primary += h.primary;
    out->secondary += h.secondary;
    out->shadow_calls += h.shadow_calls;
    out->shadow_traced += h.shadow_traced;
    out->overflow += h.overflow;
  }
#ifdef RTC_PROFILE
  if (getenv("RTC_PROFILE_DUMP")) {
    std::fprintf(stderr, "rtc prof:");
    for (int i = 0; i < 16; ++i) std::fprintf(stderr, " %llu", h.prof[i]);
    std::fprintf(stderr, " | wave lifetime min %llu max %llu (last unit %llu) sum %llu | stolen %u\n", h.prof_t0,
                 h.prof_t1 >> 24, h.prof_t1 & 0xFFFFFFull, h.prof_busy, h.stolen);
    std::fprintf(stderr, "rtc traces (invocations, lanes): closest %llu %llu | shadow %llu %llu | behind %llu %llu\n", h.prof2[0],
                 h.prof2[1], h.prof2[2], h.prof2[3], h.prof2[4], h.prof2[5]);
    // (prof4[7]: the four-wide walk counts the node steps of walks that reach no leaf; the eight-wide walk the references
    // it refused to follow because they point outside the node / leaf tables - must be 0)
    std::fprintf(stderr, "rtc walks: %llu lanes %llu node-steps %llu leaf-steps %llu lanes-at-nodes %llu lanes-at-leaves %llu | without a leaf: %llu walks | %s %llu\n",
                 h.prof4[0], h.prof4[1], h.prof4[2], h.prof4[3], h.prof4[4], h.prof4[5], h.prof4[6], RTC_BVH8 ? "bad refs" : "their node-steps", h.prof4[7]);
    std::fprintf(stderr, "rtc walk cycles: at nodes %llu at leaves %llu\n", h.prof5[5], h.prof5[6]);
    for (int k = 0; k < 3; ++k)
      std::fprintf(stderr, "rtc walks of %s traces: %llu lanes %llu node-steps %llu leaf-steps %llu lanes-at-nodes %llu lanes-at-leaves %llu cycles-at-nodes %llu cycles-at-leaves %llu\n",
                   k == 0 ? "closest" : k == 1 ? "shadow" : "containers", h.prof6[8 * k], h.prof6[8 * k + 1], h.prof6[8 * k + 2], h.prof6[8 * k + 3],
                   h.prof6[8 * k + 4], h.prof6[8 * k + 5], h.prof6[8 * k + 6], h.prof6[8 * k + 7]);
    std::fprintf(stderr, "rtc trace cycles by lanes with a ray (1-2, 3-4, 5-8, 9-16, 17-32, 33-48, 49-64):");
    for (int k = 0; k < 3; ++k) {
      std::fprintf(stderr, " %s", k == 0 ? "closest" : k == 1 ? "| shadow" : "| behind");
      for (int i = 0; i < 7; ++i) std::fprintf(stderr, " %llu", h.prof3[k * 7 + i]);
    }
    std::fprintf(stderr, "\n");
    if (const char* path = getenv("RTC_PROFILE_LOG")) {
      if (FILE* f = std::fopen(path, "w")) {
        for (int i = 0; i < 4096; ++i)
          if (h.prof_log[i][0]) {
            std::fprintf(f, "%d %llu %llu %llu %llu %llu", i, h.prof_log[i][0], h.prof_log[i][1], h.prof_log[i][2],
                         h.prof_log[i][3] >> 32, h.prof_log[i][3] & 0xFFFFFFFFull);
            for (int k = 0; k < 16; ++k) std::fprintf(f, " %llu", h.prof_last[i][k]);
            std::fprintf(f, "\n");
          }
        std::fclose(f);
      }
    }
  }
#endif
  return RTC_OK;
}

}  // extern "C"
