// rtc_multi.hip — librtc_multi.so: the single-process multi-GPU render of include/rtc_multi.h on top of the per-rank
// entry points of librtc_hip.so and RCCL (one ncclGather per frame; no other collective).  Host code only.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtc_multi.h"

namespace {

thread_local std::string g_multi_error;

int mfail(int status, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_multi_error = std::string(rtc_status_name(status)) + ": " + buf;
  return status;
}

#define M_HIP(expr)                                                                                        \
  do {                                                                                                     \
    hipError_t e_ = (expr);                                                                                \
    if (e_ != hipSuccess)                                                                                  \
      return mfail(e_ == hipErrorOutOfMemory ? RTC_ERR_OUT_OF_MEMORY : RTC_ERR_NO_DEVICE, "%s -> %s", #expr, \
                   hipGetErrorString(e_));                                                                 \
  } while (0)
#define M_NCCL(expr)                                                                                \
  do {                                                                                              \
    ncclResult_t r_ = (expr);                                                                       \
    if (r_ != ncclSuccess) return mfail(RTC_ERR_NO_DEVICE, "%s -> %s", #expr, ncclGetErrorString(r_)); \
  } while (0)
#define M_RTC(expr)                                                          \
  do {                                                                       \
    const int s_ = (expr);                                                   \
    if (s_ != RTC_OK) {                                                      \
      g_multi_error = rtc_last_error();                                      \
      return s_;                                                             \
    }                                                                        \
  } while (0)

constexpr uint32_t kTile = 64;

}  // namespace

struct rtc_multi {
  uint32_t n = 0;
  bool virt = false;
  std::vector<int> dev;
  std::vector<rtc_scene*> scene;
  std::vector<hipStream_t> stream;
  std::vector<hipEvent_t> shared;  // virtual mode: rank r's share has been copied into the gathered buffer
  std::vector<ncclComm_t> comm;
  // sized for one image size
  uint32_t hsize = 0, vsize = 0, n_tiles = 0, padded = 0;
  std::vector<double*> d_buf;  // [n] a rank's compact tiles [padded][64][64][3], on its device
  double* d_gathered = nullptr;  // device 0: [n][padded][64][64][3]
  double* d_canvas[2] = {nullptr, nullptr};  // device 0: the frame being produced and the one handed out before it
  uint32_t cur = 0;                          // which of the two the next frame is assembled into
  uint32_t* d_rgba = nullptr;    // device 0: the RGBA8 form of a frame (rtc_multi_render_rgba8), allocated on first use
  uint32_t* d_slot = nullptr;    // device 0: rtc_assign_tiles' slot_of_tile
  bool pending = false;          // a frame has been enqueued and its bookkeeping (overflow check, re-deal) is still due
  rtc_camera pending_cam{};
  std::vector<uint32_t> rank_of, slot_of;
  std::vector<std::vector<uint32_t>> tiles_of;
  bool balanced = false;
  uint32_t frames_since_balance = 0;
  uint32_t redeals_in_a_row = 0;   // (a re-deal that the very next frame's measurement did not confirm is done again, three times at most)
  rtc_camera balance_cam{};
  double max_over_mean = 0.0;
};

namespace {

size_t slabDoubles(const rtc_multi* m) { return static_cast<size_t>(m->padded) * kTile * kTile * 3u; }

void setLists(rtc_multi* m) {
  m->tiles_of.assign(m->n, {});
  for (uint32_t t = 0; t < m->n_tiles; ++t) m->tiles_of[m->rank_of[t]].push_back(t);  // increasing tile order = slot order
}

void freeFrameBuffers(rtc_multi* m) {
  for (uint32_t r = 0; r < m->d_buf.size(); ++r)
    if (m->d_buf[r]) {
      (void)hipSetDevice(m->dev[r]);
      (void)hipFree(m->d_buf[r]);
    }
  m->d_buf.clear();
  if (!m->dev.empty()) (void)hipSetDevice(m->dev[0]);
  if (m->d_gathered) (void)hipFree(m->d_gathered);
  for (double*& c : m->d_canvas) {
    if (c) (void)hipFree(c);
    c = nullptr;
  }
  if (m->d_rgba) (void)hipFree(m->d_rgba);
  if (m->d_slot) (void)hipFree(m->d_slot);
  m->d_gathered = nullptr;
  m->d_rgba = nullptr;
  m->d_slot = nullptr;
}

// Buffers and the first (round-robin) deal for one image size.  All or nothing: a failure half way leaves the handle
// without frame buffers and without a size, so that the next call starts over instead of rendering into what is missing.
int sizeForUnguarded(rtc_multi* m, const rtc_camera& cam) {
  const uint32_t tx = (cam.hsize + kTile - 1) / kTile, ty = (cam.vsize + kTile - 1) / kTile;
  m->n_tiles = tx * ty;
  m->padded = (m->n_tiles + m->n - 1) / m->n;
  m->d_buf.assign(m->n, nullptr);
  for (uint32_t r = 0; r < m->n; ++r) {
    M_HIP(hipSetDevice(m->dev[r]));
    M_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_buf[r]), slabDoubles(m) * sizeof(double)));
    M_HIP(hipMemset(m->d_buf[r], 0, slabDoubles(m) * sizeof(double)));  // slots and edge pixels nobody renders stay 0
  }
  M_HIP(hipSetDevice(m->dev[0]));
  M_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_gathered), slabDoubles(m) * m->n * sizeof(double)));
  for (double*& c : m->d_canvas)
    M_HIP(hipMalloc(reinterpret_cast<void**>(&c), static_cast<size_t>(cam.hsize) * cam.vsize * 3u * sizeof(double)));
  M_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_slot), m->n_tiles * sizeof(uint32_t)));
  // the first frame: tiles dealt round-robin (nothing has been measured yet)
  m->rank_of.resize(m->n_tiles);
  m->slot_of.resize(m->n_tiles);
  for (uint32_t t = 0; t < m->n_tiles; ++t) {
    m->rank_of[t] = t % m->n;
    m->slot_of[t] = (t % m->n) * m->padded + t / m->n;
  }
  setLists(m);
  M_HIP(hipMemcpy(m->d_slot, m->slot_of.data(), m->n_tiles * sizeof(uint32_t), hipMemcpyHostToDevice));
  return RTC_OK;
}

int sizeFor(rtc_multi* m, const rtc_camera& cam) {
  if (cam.hsize == m->hsize && cam.vsize == m->vsize && !m->d_buf.empty()) return RTC_OK;
  // (a frame of 2^32 pixels and more does not fit the 32-bit tile arithmetic - and no GPU)
  if (static_cast<uint64_t>(cam.hsize) * cam.vsize >= (1ull << 31))
    return mfail(RTC_ERR_INVALID_ARGUMENT, "image %ux%u is beyond what the tile split indexes", cam.hsize, cam.vsize);
  for (uint32_t r = 0; r < m->n; ++r) {
    M_HIP(hipSetDevice(m->dev[r]));
    M_HIP(hipStreamSynchronize(m->stream[r]));
  }
  freeFrameBuffers(m);
  m->hsize = m->vsize = 0;
  m->balanced = false;
  m->frames_since_balance = 0;
  m->redeals_in_a_row = 0;
  m->max_over_mean = 0.0;
  m->cur = 0;
  m->rank_of.clear();
  m->slot_of.clear();
  m->tiles_of.assign(m->n, {});
  const int st = sizeForUnguarded(m, cam);
  if (st != RTC_OK) {
    (void)hipGetLastError();
    freeFrameBuffers(m);
    m->rank_of.clear();
    m->slot_of.clear();
    m->tiles_of.assign(m->n, {});
    return st;
  }
  m->hsize = cam.hsize;
  m->vsize = cam.vsize;
  return RTC_OK;
}

// Re-deal the tiles by what the ranks measured for them in the frame just rendered.  confirm: that frame was the first
// with the current lists (new lists are new pixel maps, so every rank measured it afresh): keep them if the fresh costs
// say they are even, re-deal if not - one wild measurement (a wave that lost its CU for a millisecond in the middle of a
// packet: four virtual ranks on one GPU are four queues the hardware time-slices) must not skew the split for good.
int rebalance(rtc_multi* m, const rtc_camera& cam, bool confirm) {
  std::vector<double> cost(m->n_tiles, 0.0);
  for (uint32_t r = 0; r < m->n; ++r) {
    const std::vector<uint32_t>& mine = m->tiles_of[r];
    if (mine.empty()) continue;
    std::vector<double> c(mine.size());
    M_HIP(hipSetDevice(m->dev[r]));
    if (rtc_get_tile_costs(m->scene[r], c.data(), static_cast<uint32_t>(mine.size())) != RTC_OK) return RTC_OK;  // nothing measured: keep the split
    for (size_t k = 0; k < mine.size(); ++k) cost[mine[k]] = c[k];
  }
  auto imbalance = [&]() {
    std::vector<double> load(m->n, 0.0);
    double total = 0.0;
    for (uint32_t t = 0; t < m->n_tiles; ++t) {
      load[m->rank_of[t]] += cost[t];
      total += cost[t];
    }
    return total > 0.0 ? *std::max_element(load.begin(), load.end()) / (total / m->n) : 0.0;
  };
  if (confirm) {
    m->max_over_mean = imbalance();
    if (m->max_over_mean <= 1.25 || m->redeals_in_a_row >= 3u) return RTC_OK;
    m->redeals_in_a_row++;
  } else {
    m->redeals_in_a_row = 0;
  }
  M_RTC(rtc_assign_tiles(cost.data(), m->n_tiles, m->n, m->rank_of.data(), m->slot_of.data()));
  setLists(m);
  for (uint32_t r = m->n; r-- > 0u;) {  // (every rank's stream: the frame that used the old table is done everywhere; device 0 last)
    M_HIP(hipSetDevice(m->dev[r]));
    M_HIP(hipStreamSynchronize(m->stream[r]));
  }
  M_HIP(hipMemcpy(m->d_slot, m->slot_of.data(), m->n_tiles * sizeof(uint32_t), hipMemcpyHostToDevice));
  m->max_over_mean = imbalance();
  m->balanced = true;
  m->frames_since_balance = 0;
  m->balance_cam = cam;
  return RTC_OK;
}

}  // namespace

extern "C" {

const char* rtc_multi_last_error(void) { return g_multi_error.c_str(); }

int rtc_multi_create(const rtc_scene_desc* desc, uint32_t n_gpus, uint32_t flags, rtc_multi** out) {
  g_multi_error.clear();
  if (!desc || !out || n_gpus == 0) return mfail(RTC_ERR_INVALID_ARGUMENT, "null argument or no GPUs");
  *out = nullptr;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) return mfail(RTC_ERR_NO_DEVICE, "no HIP device is visible");
  const bool virt = (flags & RTC_MULTI_VIRTUAL) != 0u;
  if (!virt && n_gpus > static_cast<uint32_t>(n_dev))
    return mfail(RTC_ERR_INVALID_ARGUMENT, "%u GPUs asked for, %d visible", n_gpus, n_dev);
  rtc_multi* m = new (std::nothrow) rtc_multi();
  if (!m) return mfail(RTC_ERR_OUT_OF_MEMORY, "host allocation");
  struct Guard {
    rtc_multi* m;
    ~Guard() {
      if (m) rtc_multi_destroy(m);
    }
  } guard{m};
  m->n = n_gpus;
  m->virt = virt;
  m->dev.resize(n_gpus);
  m->scene.assign(n_gpus, nullptr);
  m->stream.assign(n_gpus, nullptr);
  m->shared.assign(n_gpus, nullptr);
  for (uint32_t r = 0; r < n_gpus; ++r) m->dev[r] = virt ? 0 : static_cast<int>(r);
  for (uint32_t r = 0; r < n_gpus; ++r) {
    M_HIP(hipSetDevice(m->dev[r]));
    M_RTC(rtc_scene_create(desc, &m->scene[r]));  // the scene (<= 30 MB) is replicated
    M_HIP(hipStreamCreateWithFlags(&m->stream[r], hipStreamNonBlocking));
    M_HIP(hipEventCreateWithFlags(&m->shared[r], hipEventDisableTiming));
  }
  if (!virt) {
    m->comm.assign(n_gpus, nullptr);
    M_NCCL(ncclCommInitAll(m->comm.data(), static_cast<int>(n_gpus), m->dev.data()));
  }
  guard.m = nullptr;
  *out = m;
  return RTC_OK;
}

void rtc_multi_destroy(rtc_multi* m) {
  if (!m) return;
  for (uint32_t r = 0; r < m->stream.size(); ++r) {
    (void)hipSetDevice(m->dev[r]);
    if (m->stream[r]) (void)hipStreamSynchronize(m->stream[r]);
  }
  for (ncclComm_t c : m->comm)
    if (c) (void)ncclCommDestroy(c);
  freeFrameBuffers(m);
  for (uint32_t r = 0; r < m->scene.size(); ++r) {
    (void)hipSetDevice(m->dev[r]);
    if (m->scene[r]) rtc_scene_destroy(m->scene[r]);
    if (m->shared[r]) (void)hipEventDestroy(m->shared[r]);
    if (m->stream[r]) (void)hipStreamDestroy(m->stream[r]);
  }
  delete m;
}

}  // extern "C"

namespace {

// What a frame owes once it is done: every stream drained, the overflow check, the re-deal of the tiles by measured cost
// (after the first frame of an image size, and every 16 frames while the camera is not where the split was measured - a
// moving camera measures every frame; the frame after a re-deal confirms it).  Idempotent.
int finishFrame(rtc_multi* m) {
  if (!m->pending) return RTC_OK;
  m->pending = false;
  for (uint32_t r = m->n; r-- > 0u;) {
    M_HIP(hipSetDevice(m->dev[r]));
    M_HIP(hipStreamSynchronize(m->stream[r]));
  }
  rtc_stats st;
  if (const int s = rtc_multi_get_stats(m, &st); s != RTC_OK) return s;
  if (st.overflow) return mfail(RTC_ERR_OVERFLOW, "%llu lanes overflowed a per-lane stack or csg list", (unsigned long long)st.overflow);
  m->frames_since_balance++;
  const rtc_camera& cam = m->pending_cam;
  const bool moved = std::memcmp(&cam, &m->balance_cam, sizeof cam) != 0;
  const bool confirm = m->balanced && m->frames_since_balance == 1u;
  if (m->n > 1 && (!m->balanced || (moved && m->frames_since_balance >= 16u) || confirm))
    if (const int s = rebalance(m, cam, confirm); s != RTC_OK) return s;
  return RTC_OK;
}

// One frame, enqueued: every rank renders its tiles into its compact buffer, ONE gather brings them to rank 0 (each
// rank's send is ordered behind its render on its stream), one kernel un-permutes them into the row-major canvas
// d_canvas[cur] on device 0, stream[0].  Nothing here waits for the GPUs (the frame before has been finished: the
// buffers are free).
int enqueueFrame(rtc_multi* m, const rtc_camera* cam, uint32_t max_depth, double** d_canvas) {
  if (!m || !cam) return mfail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  if (cam->hsize == 0 || cam->vsize == 0) return mfail(RTC_ERR_INVALID_ARGUMENT, "camera %ux%u", cam->hsize, cam->vsize);
  if (const int st = finishFrame(m); st != RTC_OK) return st;
  if (const int st = sizeFor(m, *cam); st != RTC_OK) return st;
  for (uint32_t r = 0; r < m->n; ++r) {
    if (m->tiles_of[r].empty()) continue;
    M_HIP(hipSetDevice(m->dev[r]));
    M_RTC(rtc_render_tile_list_device(m->scene[r], cam, max_depth, kTile, kTile, m->tiles_of[r].data(),
                                      static_cast<uint32_t>(m->tiles_of[r].size()), m->d_buf[r], m->stream[r]));
  }
  m->pending = true;  // (from here on the streams hold work of this frame)
  m->pending_cam = *cam;
  const size_t slab = slabDoubles(m);
  if (!m->virt) {
    M_NCCL(ncclGroupStart());
    for (uint32_t r = 0; r < m->n; ++r) {
      M_HIP(hipSetDevice(m->dev[r]));
      M_NCCL(ncclGather(m->d_buf[r], m->d_gathered, slab, ncclDouble, 0, m->comm[r], m->stream[r]));
    }
    M_NCCL(ncclGroupEnd());
  } else {
    M_HIP(hipSetDevice(m->dev[0]));
    for (uint32_t r = 0; r < m->n; ++r) {
      M_HIP(hipMemcpyAsync(m->d_gathered + slab * r, m->d_buf[r], slab * sizeof(double), hipMemcpyDeviceToDevice, m->stream[r]));
      M_HIP(hipEventRecord(m->shared[r], m->stream[r]));
      if (r != 0) M_HIP(hipStreamWaitEvent(m->stream[0], m->shared[r], 0));
    }
  }
  M_HIP(hipSetDevice(m->dev[0]));
  double* const canvas = m->d_canvas[m->cur];
  m->cur ^= 1u;
  M_RTC(rtc_assemble_tile_list_device(m->d_gathered, m->d_slot, kTile, kTile, cam->hsize, cam->vsize, canvas, m->stream[0]));
  *d_canvas = canvas;
  return RTC_OK;
}

}  // namespace

extern "C" {

int rtc_multi_render(rtc_multi* m, const rtc_camera* cam, uint32_t max_depth, double* rgb_out) {
  g_multi_error.clear();
  if (!rgb_out) return mfail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  double* d_canvas = nullptr;
  if (const int st = enqueueFrame(m, cam, max_depth, &d_canvas); st != RTC_OK) return st;
  M_HIP(hipMemcpyAsync(rgb_out, d_canvas, static_cast<size_t>(cam->hsize) * cam->vsize * 3u * sizeof(double),
                       hipMemcpyDeviceToHost, m->stream[0]));
  return finishFrame(m);
}

int rtc_multi_render_rgba8(rtc_multi* m, const rtc_camera* cam, uint32_t max_depth, uint8_t* rgba_out) {
  g_multi_error.clear();
  if (!rgba_out) return mfail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  double* d_canvas = nullptr;
  if (const int st = enqueueFrame(m, cam, max_depth, &d_canvas); st != RTC_OK) return st;
  const size_t n = static_cast<size_t>(cam->hsize) * cam->vsize;
  if (!m->d_rgba) M_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_rgba), n * sizeof(uint32_t)));  // (freed with the frame buffers of this size)
  M_RTC(rtc_rgba8_device(d_canvas, n, m->d_rgba, m->stream[0]));
  M_HIP(hipMemcpyAsync(rgba_out, m->d_rgba, n * sizeof(uint32_t), hipMemcpyDeviceToHost, m->stream[0]));
  return finishFrame(m);
}

int rtc_multi_render_device(rtc_multi* m, const rtc_camera* cam, uint32_t max_depth, const double** d_canvas_out) {
  g_multi_error.clear();
  if (!d_canvas_out) return mfail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  double* d_canvas = nullptr;
  if (const int st = enqueueFrame(m, cam, max_depth, &d_canvas); st != RTC_OK) return st;
  *d_canvas_out = d_canvas;
  return RTC_OK;
}

int rtc_multi_synchronize(rtc_multi* m) {
  g_multi_error.clear();
  if (!m) return mfail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  return finishFrame(m);
}

void* rtc_multi_stream(rtc_multi* m) { return m ? static_cast<void*>(m->stream[0]) : nullptr; }

int rtc_multi_get_stats(rtc_multi* m, rtc_stats* out) {
  if (!m || !out) return mfail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  std::memset(out, 0, sizeof *out);
  for (uint32_t r = 0; r < m->n; ++r) {
    if (m->tiles_of.size() > r && m->tiles_of[r].empty()) continue;
    rtc_stats s;
    M_HIP(hipSetDevice(m->dev[r]));
    M_RTC(rtc_get_stats(m->scene[r], &s));
    out->primary += s.primary;
    out->secondary += s.secondary;
    out->shadow_calls += s.shadow_calls;
    out->shadow_traced += s.shadow_traced;
    out->overflow += s.overflow;
  }
  return RTC_OK;
}

int rtc_multi_balance(rtc_multi* m, uint32_t* tiles_per_rank, double* max_over_mean) {
  if (!m || !tiles_per_rank || !max_over_mean) return mfail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  for (uint32_t r = 0; r < m->n; ++r) tiles_per_rank[r] = r < m->tiles_of.size() ? static_cast<uint32_t>(m->tiles_of[r].size()) : 0u;
  *max_over_mean = m->max_over_mean;
  return RTC_OK;
}

}  // extern "C"
