// rtc_host_internal.h — what the host-side translation units of librtc_hip.so share: error reporting, the
// device-buffer holder and the scene handle behind `rtc_scene*` (include/rtc.h).  Included by rtc_capi.hip only.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstddef>
#include <cstring>
#include <string>
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
      return fail(e_ == hipErrorOutOfMemory ? RTC_ERR_OUT_OF_MEMORY : RTC_ERR_NO_DEVICE,      \
                  "%s -> %s", #expr, hipGetErrorString(e_));                                  \
    }                                                                                         \
  } while (0)

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

struct rtc_scene {
  int device = 0;
  hipStream_t stream = nullptr;
  DevScene dev{};
  DevStats* d_stats = nullptr;  // two, used alternately (see DevStats)
  uint32_t stats_parity = 0;    // which of the two the last launch counted in
  double* d_frame = nullptr;  // staging for rtc_render (host output)
  size_t frame_capacity = 0;  // in doubles
  DevBuf<uint32_t> roots, kids;
  DevBuf<RootRec> root_recs;
  DevBuf<RootCullPair> root_cull;
  DevBuf<uint4> leaf_meta;
  DevBuf<double> xf, tri, trin, node_box, light;
  DevBuf<DevPattern> pat;
  DevBuf<DevCyl> cyl;
  DevBuf<DevMaterial> mat;
  DevBuf<uint2> node_kids;
  DevBuf<Bvh4Node> bvh;
  DevBuf<BvhLeafRec> bvh_leaf;
  DevBuf<uint32_t> leaf_parent, node_parent, node_info;
  DevBuf<uint2> node_range;
  DevBuf<DevTexMap> tex;
  DevBuf<DevUv> uv;
  DevBuf<DevImage> img;
  DevBuf<float> img_rgb;
  bool has_csg = false;
  bool ext_kernel = false;         // csg nodes or texture maps: the *_ext kernels
  bool simple_kernel = false;      // only top-level spheres / planes / cubes: the `simple` kernel
  void* d_csg_buf = nullptr;       // DevPixelMap::csg_buf, only for scenes with csg nodes
  size_t csg_buf_capacity = 0;     // bytes
  uint32_t max_trav_stack = 0;
  uint32_t n_cus = 0, blocks_per_cu_lds = 1, blocks_per_cu_big = 1;
  // heavy-first scheduling hint (see DevPixelMap::order)
  std::vector<Sphere> occupied;    // bounding spheres of every bounded root
  bool unbounded_nonplane = false; // a root other than a plane without a finite bound (cannot be projected)
  bool plane_spawns_rays = false;  // a top-level plane reflects or refracts: "sees only planes" does not mean cheap
  std::vector<Sphere> branching;   // bounding spheres of roots whose materials branch the ray tree
  bool branching_everywhere = false;  // such a root without a finite bound
  std::vector<uint32_t> h_order;
  std::vector<float> h_split_inflation;   // per chunk: modelled time of its parts / its time whole, for the schedule in h_order (empty: nothing is cut)
  std::vector<float> measured_inflation;  // ... for the schedule the last measuring launch ran
  uint32_t* d_order = nullptr;
  size_t order_capacity = 0;
  std::vector<double> order_key;   // camera + map the cached (heuristic) order was built for
  // measured-cost feedback: per-chunk ray counts of the previous launch with the same pixel map
  uint32_t* d_cost = nullptr;
  size_t cost_capacity = 0;
  std::vector<uint32_t> cost_key;  // pixel map the costs / the cost-sorted order belong to
  std::vector<uint32_t> h_cost;
  uint32_t* d_chunk_cost = nullptr;  // per-chunk sums of d_cost (rtc_chunk_cost_kernel)
  size_t chunk_cost_capacity = 0;
  std::vector<uint32_t> h_chunk_cost;
  uint32_t* d_packet_time = nullptr;  // per packet of the measured schedule: the time its wave needed (DevPixelMap::packet_time)
  // A measuring launch is followed, on its stream, by the per-chunk sums, two copies into pinned host memory and this
  // event: the launch that finds the event complete packs the new schedule without waiting for anything.
  hipEvent_t measure_done = nullptr;
  bool readback_enqueued = false;
  uint32_t launches_since_measure = 0;
  size_t readback_packets = 0;
  uint32_t* pin_chunk_cost = nullptr;
  size_t pin_chunk_cost_capacity = 0;
  uint32_t* pin_packet_time = nullptr;
  size_t pin_packet_time_capacity = 0;
  size_t packet_time_capacity = 0;
  std::vector<uint32_t> h_packet_time;
  std::vector<uint32_t> h_chunk_time_dbg;  // per-chunk times the schedule in use was packed from (diagnostics)
  std::vector<uint32_t> measured_order;  // the schedule (h_order) of the measuring launch; empty: packet i was chunk i
  uint64_t launches_with_key = 0;
  bool order_from_cost = false;
  bool cost_pending = false;       // the previous launch measured per-pixel costs: the next one packs from them
  rtc_camera cost_cam{};           // camera (and depth) of that measurement ...
  uint32_t cost_depth = 0;
  rtc_camera sched_cam{};          // ... and of the measurement the current schedule was packed from
  uint32_t sched_depth = 0;
  hipStream_t last_stream = nullptr;  // the stream of the last launch (or the handle's own, after create)
  hipEvent_t launch_done = nullptr;   // recorded behind everything a launch enqueues; a launch on ANOTHER stream waits for it
  void* d_ray_stack = nullptr;     // DevPixelMap::ray_stack
  size_t ray_stack_capacity = 0;   // bytes
};
