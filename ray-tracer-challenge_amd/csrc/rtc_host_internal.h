// rtc_host_internal.h — what the host-side translation units of librtc_hip.so share: error reporting, the
// device-buffer holder and the scene handle behind `rtc_scene*` (include/rtc.h).  Included by rtc_capi.hip only.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstddef>
#include <cstring>
#include <string>
#include <memory>
#include <vector>

#include "../../include/rtc.h"
#include "rtc_device.h"

namespace {

thread_local std::string g_error;

int fail(int status, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_error = std::string(rtc_status_name(status)) + ": " + buf;
  return status;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      (void)hipGetLastError(); /* reported here: not the next call's hipGetLastError() */     \
      return fail(e_ == hipErrorOutOfMemory ? RTC_ERR_OUT_OF_MEMORY : RTC_ERR_NO_DEVICE,      \
                  "%s -> %s", #expr, hipGetErrorString(e_));                                  \
    }                                                                                         \
  } while (0)

// Tuning and test options (rtc_set_option, include/rtc_diag.h): process-wide, read when a scene is created or a launch is
// enqueued; none changes a result.  0 / negative = the library's own choice.  Every option is an atomic double (relaxed:
// an option is one independent number): a host thread may set one while another thread's launch reads it.
struct RtcOption {
  std::atomic<double> v;
  RtcOption(double d) : v(d) {}
  operator double() const { return v.load(std::memory_order_relaxed); }
  void set(double d) { v.store(d, std::memory_order_relaxed); }
};
struct RtcOptions {
  RtcOption simple3_min_chunks{-1.0};  // chunks from which a simple world runs the three-wave kernel (0: always)
  RtcOption sched_off{0.0};            // != 0: no schedule at all (packet i is chunk i)
  RtcOption cut_above{0.0};            // shares of a wave above which a chunk is cut into runs (< 0: never)
  RtcOption pack_rounds{3.0};          // rounds of rtc_pack_extra_kernel
  RtcOption pull_min_idle{64.0};       // idle lanes before a wave pulls its next packet
  RtcOption blocks_per_cu{0.0};        // cap on resident work-groups per CU
  RtcOption sched_tmin{0.0};           // time up to which cheap chunks share a packet
  RtcOption bvh_leaf{RTC_BVH8 ? 1.0 : 2.0};  // leaves per candidate-BVH leaf (eight-wide tree, 1 / 2 / 3 / 4: dragons 4K 2.03 / 2.10 / 2.18 / 2.30 ms, nefertiti 0.522 / 0.534 / 0.548 / 0.562, teapot 0.264 / 0.265 / 0.263 / 0.272)
  RtcOption bvh_one_axis{0.0};         // != 0: SAH on the longest axis only
  RtcOption bvh_check{0.0};            // != 0: host self-check of the candidate BVH at create (stderr)
  RtcOption waves3{-1.0};              // the general kernel at three waves per SIMD: 1 always (where the tables fit), 0 never, < 0: measured per handle
  RtcOption sched_mix{773.0};          // a | b << 8: behind every wave's first packet the schedule takes a packets from its long end, b from its short end, ... (5 : 3); 0: longest first throughout
  RtcOption inflight_chunks_per_wave{3.0};  // a launch of a scene with frames in flight runs on at most one wave per this many chunks (0: no cap)
  RtcOption measure_every{1.0};        // a moving view is measured (and its schedule re-packed) every this many frames (see updateSchedule)
  RtcOption host_bands{0.0};           // bands rtc_render cuts a frame into (copy of band i under the render of band i + 1); 0: by size
  RtcOption box_cull{-1.0};            // a simple world's kernels reject roots by world boxes (1) or bounding spheres (0); < 0: boxes if it has more cubes than spheres
  RtcOption build_threads{0.0};        // threads of rtc_scene_create's candidate-BVH build (one top-level group each); 0: as many as the host allows, up to 8
};
inline RtcOptions& rtcOptions() {
  static RtcOptions options;
  return options;
}

template <typename T>
struct DevBuf {
  T* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t upload(const std::vector<T>& v) {
    const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);  // never a null table
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), bytes);
    if (e != hipSuccess) return e;
    if (!v.empty()) e = hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return e;
  }
};

}  // namespace

// conservative world-space bounding sphere (see leafSphere below)
struct Sphere {
  double cx = 0, cy = 0, cz = 0, r = INFINITY;
  bool finite() const { return std::isfinite(r) && std::isfinite(cx) && std::isfinite(cy) && std::isfinite(cz); }
};

// The scene's tables in device memory: written by rtc_scene_create, read-only from then on, so a handle and its clones
// (rtc_scene_clone: one handle per frame in flight) share one copy; freed with the last of them.
struct SceneTables {
  std::atomic<int> handles{0};  // handles that share this copy: more than one = frames in flight (kernel choice, rtc_capi.hip)
  // the csg list length any handle of this scene has grown to (rtc_grow_csg_lists): the others - clones, one per frame in
  // flight - take it over at their next launch instead of overflowing, growing and rendering again one by one
  std::atomic<uint32_t> csg_entries_wanted{0};
  DevBuf<uint32_t> roots, kids;
  DevBuf<RootRec> root_recs;
  DevBuf<RootCullPair> root_cull;
  DevBuf<RootBoxPair> root_box;
  DevBuf<float> root_weight;
  DevBuf<uint4> leaf_meta;
  DevBuf<double> xf, tri, trin, node_box, light;
  DevBuf<DevPattern> pat;
  DevBuf<DevCyl> cyl;
  DevBuf<DevMaterial> mat;
  DevBuf<uint2> node_kids;
  DevBuf<Bvh4Node> bvh;
  DevBuf<Bvh8Node> bvh8;
  DevBuf<BvhLeafRec> bvh_leaf;
  DevBuf<uint32_t> leaf_parent, node_parent, node_info;
  DevBuf<uint2> node_range;
  DevBuf<DevTexMap> tex;
  DevBuf<DevUv> uv;
  DevBuf<DevImage> img;
  DevBuf<float> img_rgb;
};

constexpr uint32_t RTC_MAX_HOST_BANDS = 4;

struct rtc_scene {
  int device = 0;
  hipStream_t stream = nullptr;
  DevScene dev{};
  DevStats* d_stats = nullptr;  // two, used alternately (see DevStats)
  uint32_t stats_parity = 0;    // which of the two the last launch counted in
  double* d_frame = nullptr;  // staging for rtc_render (host output)
  size_t frame_capacity = 0;  // in doubles
  std::shared_ptr<SceneTables> tab;  // the scene in device memory (`dev` points into it): shared with the handle's clones
  bool has_csg = false;
  bool ext_kernel = false;         // csg nodes or texture maps: the *_ext kernels
  bool simple_kernel = false;      // only top-level spheres / planes / cubes: the `simple` kernel
  bool flat_kernel = false;        // no groups at all (any leaf kind): the `flat` kernel
  const char* last_kernel = "";   // name of the render kernel of the last launch (rtc_last_kernel_name)
  bool simple3_ok = false;         // a simple world whose tables fit the three-waves-per-SIMD kernel's LDS (RTC_LDS3_*)
  void* d_csg_buf = nullptr;       // DevPixelMap::csg_buf, only for scenes with csg nodes
  size_t csg_buf_capacity = 0;     // bytes
  uint32_t csg_entries_ok = RTC_CSG_ENTRIES;  // the list length the buffer was last allocated for (what a failed enlargement falls back to)
  uint32_t csg_needed = 0;         // what the last checked frame said its longest csg list needed (0: no csg list ran out)
  uint32_t max_trav_stack = 0;
  // ---- the general kernel at two or at three waves per SIMD: measured, not guessed (launch(), KernelTune)
  bool box_cull = false;           // a simple world with more cubes than spheres at top level: its kernels reject roots by world boxes (rtc_render_kernel_simple_b / _simple3_b)
  bool general3_ok = false;        // a world with groups, no csg / texture maps, whose tables fit the three-wave kernel's LDS
  uint32_t blocks_per_cu_general3 = 1;
  bool use_three_waves = false;    // the three-wave form of the world's kernel (rtc_render_kernel3 / _simple3 at mid sizes): what the handle's last finished trial DECIDED (clones inherit this, nothing else)
  bool trial_live = false;         // the launch being enqueued is a frame of a running trial ...
  bool trial_three = false;        // ... of this kernel; anything that ends or suspends the trial (a clone made, another pixel map, an option) falls back to the decision
  struct KernelTune {
    enum { kSamples = 3, kRing = 8 };
    int state = 0;                 // 0: no trial yet for this pixel map, 1: trial running, 2: decided
    uint32_t frames = 0;           // trial frames enqueued
    float best[2] = {0.0f, 0.0f};  // fastest frame seen per kernel (two waves, three waves), ms
    uint32_t n[2] = {0, 0};        // samples resolved per kernel
    bool switched = false;         // the frame that changes to the three-wave kernel (and measures for its schedule) has been enqueued
    uint32_t sched_before = 0;     // ... the schedule buffer the two-wave samples ran, and how many schedules had been packed then
    uint32_t packs_at_switch = 0;
    hipEvent_t ev[kRing][2] = {};  // timing events around the render kernel of the trial frames (created on first use)
    int which[kRing] = {};         // which kernel a ring slot timed; -1: slot free / resolved
    std::vector<uint32_t> key;     // the pixel map the trial belongs to
  } tune;
  bool kernel_warm = false;        // the handle's first launch has sent the render kernel ahead once with no work (launch())
  uint32_t n_cus = 0, blocks_per_cu_lds = 1, blocks_per_cu_big = 1, blocks_per_cu_simple3 = 1;
  // ---- the schedule (DevPixelMap::order): two device buffers, used alternately.  d_sched[sched_cur] is what the next
  // launch runs; a measuring launch is followed by the packer's launches, which pack the other buffer from what the
  // launch measured (the first launch of a pixel map: from rtc_estimate_kernel's guesses), and the buffers swap - no host
  // in the loop, nothing is read back.
  uint32_t* d_sched[2] = {nullptr, nullptr};
  size_t sched_capacity = 0;              // words per buffer
  uint32_t sched_cur = 0;
  DevSchedInfo* d_sched_info = nullptr;   // [2], beside the buffers: the packet count of each (DevPixelMap::n_units_dev)
  DevPackState* d_pack_state = nullptr;   // the packer's histogram and totals
  bool sched_valid = false;               // d_sched[sched_cur] holds a schedule for the pixel map `cost_key`
  rtc_camera sched_cam{};                 // the view (and depth) that schedule was measured with: another view measures again
  uint32_t sched_depth = 0;
  uint32_t n_packs = 0;                   // schedules packed on this handle so far
  uint32_t frames_unmeasured = 0;         // frames of a moving view since the last measured one (updateSchedule)
  uint32_t* d_chunk_time = nullptr;       // the packer's scratch: per-chunk times, sorted chunks
  uint32_t* d_sorted = nullptr;
  DevChunkShape* d_chunk_shape = nullptr; // per chunk: how its rays are spread over its pixels (for the chunks the packer cuts)
  size_t pack_capacity = 0;               // chunks
  // ---- measurements: per-pixel ray counts and per-packet times of a measuring launch
  uint32_t* d_cost = nullptr;
  size_t cost_capacity = 0;
  std::vector<uint32_t> cost_key;  // pixel map the costs / the schedule belong to
  uint32_t* d_chunk_cost = nullptr;  // per-chunk sums of d_cost (rtc_chunk_cost_kernel)
  size_t chunk_cost_capacity = 0;
  uint32_t* d_packet_time = nullptr;  // per packet of the measured schedule: the time its wave needed (DevPixelMap::packet_time)
  size_t packet_time_capacity = 0;
  // mode-2 pixel maps (rtc_render_tile_list_device): the list of the last such launch, on both sides
  std::vector<uint32_t> h_tile_list;
  uint32_t* d_tile_list = nullptr;
  size_t tile_list_capacity = 0;
  uint32_t tile_list_gen = 0;
  uint32_t measured_regions = 0, measured_chunks_per_region = 0;  // what d_chunk_time describes (rtc_get_tile_costs)
  hipStream_t last_stream = nullptr;  // the stream of the last launch (none yet: nullptr)
  bool stats_zeroed = false;          // the first launch has zeroed d_stats on its stream (initLaunchState)
  hipEvent_t launch_done = nullptr;   // recorded behind everything a launch enqueues; a launch on ANOTHER stream waits for it
  void* d_ray_stack = nullptr;     // DevPixelMap::ray_stack
  size_t ray_stack_capacity = 0;   // bytes
  // ---- host output of a large frame (rtc_render): rendered in horizontal bands one after the other, each copied to the
  // caller while the next renders.  A band is a pixel map of its own: the lower bands run on clones of this handle (made on
  // first use), which keep their band's schedule from frame to frame.
  std::vector<rtc_scene*> band;
  hipStream_t copy_stream = nullptr;
  hipEvent_t band_done[RTC_MAX_HOST_BANDS] = {};
  bool is_band = false;            // this handle is one of another's band clones (not counted in SceneTables::handles)
  uint32_t last_bands = 1;         // bands of the last frame through rtc_render (1: the last launch was an ordinary one)
};
