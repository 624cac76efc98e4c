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
constexpr uint32_t kMaxFrames = 8;

// One frame in flight: per rank a scene handle (slot 0: the scene itself; the others: its clones - same device copy of
// the scene, own schedule and counters), the stream it renders on, its compact tile buffer; on device 0 the gathered
// shares.  A slot runs one frame at a time; the slots take the frames in turn.
struct Slot {
  std::vector<rtc_scene*> scene;
  std::vector<hipStream_t> stream;
  std::vector<hipEvent_t> rendered;  // rank r's share is in its buffer (virtual ranks: has been copied to device 0)
  std::vector<hipEvent_t> sent;      // RCCL: rank r's part of the gather is done, its buffer may be rendered into again
  std::vector<double*> d_buf;        // [n] a rank's compact tiles [padded][64][64][3], on its device
  double* d_gathered = nullptr;      // device 0: [n][padded][64][64][3]
  std::vector<uint32_t*> d_rgba_buf; // [n] the same tiles clamped to RGBA8, [padded][64][64] (first RGBA8 frame on)
  uint32_t* d_gathered_rgba = nullptr;
  hipEvent_t gathered = nullptr;     // RCCL: the gather has arrived on device 0
  bool pending = false;              // a frame has been enqueued and its bookkeeping (overflow check, re-deal) is still due
  rtc_camera cam{};
};

}  // namespace

struct rtc_multi {
  uint32_t n = 0;
  uint32_t frames = 1;  // slots
  bool virt = false;
  std::vector<int> dev;
  std::vector<Slot> slot;
  uint32_t next_slot = 0, last_slot = 0;
  std::vector<ncclComm_t> comm;
  std::vector<hipStream_t> cstream;  // RCCL: one stream per rank that carries its gathers, frame after frame
  // sized for one image size
  uint32_t hsize = 0, vsize = 0, n_tiles = 0, padded = 0;
  bool sized = false;
  std::vector<double*> d_canvas;  // device 0: a ring of max(2, frames) canvases; a frame's stays valid while the next ones are produced
  uint32_t cur = 0;               // which of them the next frame is assembled into
  std::vector<uint32_t*> d_rgba;  // device 0: the ring of RGBA8 framebuffers (the rgba8 entry points), allocated on first use
  uint32_t cur_rgba = 0;
  uint32_t* d_slot = nullptr;     // device 0: rtc_assign_tiles' slot_of_tile
  std::vector<uint32_t> rank_of, slot_of;
  std::vector<std::vector<uint32_t>> tiles_of;
  bool balanced = false;
  uint32_t frames_since_balance = 0;
  uint32_t redeals_in_a_row = 0;   // (a re-deal that the very next frame's measurement did not confirm is done again, three times at most)
  rtc_camera balance_cam{};
  double max_over_mean = 0.0;
  bool csg_grown = false;           // finishSlot enlarged csg lists after an overflow: the synchronous entry points render again
  std::vector<uint32_t*> d_list;    // [n] a rank's tile list on its device (the direct host forms scatter by it)
};

namespace {

size_t slabDoubles(const rtc_multi* m) { return static_cast<size_t>(m->padded) * kTile * kTile * 3u; }

void setLists(rtc_multi* m) {
  m->tiles_of.assign(m->n, {});
  for (uint32_t t = 0; t < m->n_tiles; ++t) m->tiles_of[m->rank_of[t]].push_back(t);  // increasing tile order = slot order
}

// The ranks' tile lists on their devices (for rtc_scatter_tile_list_device); called with every frame finished.
int uploadLists(rtc_multi* m) {
  if (m->d_list.size() != m->n) m->d_list.assign(m->n, nullptr);
  for (uint32_t r = 0; r < m->n; ++r) {
    M_HIP(hipSetDevice(m->dev[r]));
    if (!m->d_list[r]) M_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_list[r]), std::max<size_t>(1, m->padded) * sizeof(uint32_t)));
    if (!m->tiles_of[r].empty())
      M_HIP(hipMemcpy(m->d_list[r], m->tiles_of[r].data(), m->tiles_of[r].size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  return RTC_OK;
}

void freeFrameBuffers(rtc_multi* m) {
  for (uint32_t r = 0; r < m->d_list.size(); ++r)
    if (m->d_list[r]) {
      (void)hipSetDevice(m->dev[r]);
      (void)hipFree(m->d_list[r]);
    }
  m->d_list.clear();
  for (Slot& S : m->slot) {
    for (uint32_t r = 0; r < S.d_buf.size(); ++r)
      if (S.d_buf[r]) {
        (void)hipSetDevice(m->dev[r]);
        (void)hipFree(S.d_buf[r]);
      }
    S.d_buf.clear();
    if (!m->dev.empty()) (void)hipSetDevice(m->dev[0]);
    if (S.d_gathered) (void)hipFree(S.d_gathered);
    S.d_gathered = nullptr;
    for (uint32_t r = 0; r < S.d_rgba_buf.size(); ++r)
      if (S.d_rgba_buf[r]) {
        (void)hipSetDevice(m->dev[r]);
        (void)hipFree(S.d_rgba_buf[r]);
      }
    S.d_rgba_buf.clear();
    if (!m->dev.empty()) (void)hipSetDevice(m->dev[0]);
    if (S.d_gathered_rgba) (void)hipFree(S.d_gathered_rgba);
    S.d_gathered_rgba = nullptr;
  }
  if (!m->dev.empty()) (void)hipSetDevice(m->dev[0]);
  for (double* c : m->d_canvas)
    if (c) (void)hipFree(c);
  m->d_canvas.clear();
  for (uint32_t* c : m->d_rgba)
    if (c) (void)hipFree(c);
  m->d_rgba.clear();
  if (m->d_slot) (void)hipFree(m->d_slot);
  m->d_slot = nullptr;
  m->sized = false;
}

// Everything slot f's streams hold has run (RCCL: and every rank's part of its gather).
int drainSlot(rtc_multi* m, uint32_t f) {
  Slot& S = m->slot[f];
  for (uint32_t r = m->n; r-- > 0u;) {  // (device 0 last)
    M_HIP(hipSetDevice(m->dev[r]));
    M_HIP(hipStreamSynchronize(S.stream[r]));
    if (!m->virt) M_HIP(hipEventSynchronize(S.sent[r]));
  }
  return RTC_OK;
}

// Buffers and the first (round-robin) deal for one image size.  All or nothing: a failure half way leaves the handle
// without frame buffers and without a size, so that the next call starts over instead of rendering into what is missing.
int sizeForUnguarded(rtc_multi* m, const rtc_camera& cam) {
  const uint32_t tx = (cam.hsize + kTile - 1) / kTile, ty = (cam.vsize + kTile - 1) / kTile;
  m->n_tiles = tx * ty;
  m->padded = (m->n_tiles + m->n - 1) / m->n;
  for (Slot& S : m->slot) {
    S.d_buf.assign(m->n, nullptr);
    for (uint32_t r = 0; r < m->n; ++r) {
      M_HIP(hipSetDevice(m->dev[r]));
      M_HIP(hipMalloc(reinterpret_cast<void**>(&S.d_buf[r]), slabDoubles(m) * sizeof(double)));
      M_HIP(hipMemset(S.d_buf[r], 0, slabDoubles(m) * sizeof(double)));  // slots and edge pixels nobody renders stay 0
    }
    M_HIP(hipSetDevice(m->dev[0]));
    M_HIP(hipMalloc(reinterpret_cast<void**>(&S.d_gathered), slabDoubles(m) * m->n * sizeof(double)));
  }
  M_HIP(hipSetDevice(m->dev[0]));
  m->d_canvas.assign(std::max(2u, m->frames), nullptr);
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
  return uploadLists(m);
}

// The RGBA8 side of the frame buffers (per slot and rank the clamped tiles, on device 0 the gathered ones and the ring of
// framebuffers): made by the first RGBA8 frame of an image size, freed with the rest.  All or nothing.
int ensureRgbaBuffers(rtc_multi* m) {
  if (!m->d_rgba.empty()) return RTC_OK;
  const size_t tile_pixels = static_cast<size_t>(m->padded) * kTile * kTile;
  auto undo = [&]() {
    (void)hipGetLastError();
    for (Slot& S : m->slot) {
      for (uint32_t r = 0; r < S.d_rgba_buf.size(); ++r)
        if (S.d_rgba_buf[r]) {
          (void)hipSetDevice(m->dev[r]);
          (void)hipFree(S.d_rgba_buf[r]);
        }
      S.d_rgba_buf.clear();
      (void)hipSetDevice(m->dev[0]);
      if (S.d_gathered_rgba) (void)hipFree(S.d_gathered_rgba);
      S.d_gathered_rgba = nullptr;
    }
    for (uint32_t* c : m->d_rgba)
      if (c) (void)hipFree(c);
    m->d_rgba.clear();
  };
  auto make = [&]() -> int {
    for (Slot& S : m->slot) {
      S.d_rgba_buf.assign(m->n, nullptr);
      for (uint32_t r = 0; r < m->n; ++r) {
        M_HIP(hipSetDevice(m->dev[r]));
        M_HIP(hipMalloc(reinterpret_cast<void**>(&S.d_rgba_buf[r]), tile_pixels * sizeof(uint32_t)));
      }
      M_HIP(hipSetDevice(m->dev[0]));
      M_HIP(hipMalloc(reinterpret_cast<void**>(&S.d_gathered_rgba), tile_pixels * m->n * sizeof(uint32_t)));
    }
    M_HIP(hipSetDevice(m->dev[0]));
    m->d_rgba.assign(std::max(2u, m->frames), nullptr);
    for (uint32_t*& c : m->d_rgba) M_HIP(hipMalloc(reinterpret_cast<void**>(&c), static_cast<size_t>(m->hsize) * m->vsize * sizeof(uint32_t)));
    m->cur_rgba = 0;
    return RTC_OK;
  };
  const int st = make();
  if (st != RTC_OK) undo();
  return st;
}

// Is `p` host memory the GPUs can write in place - a canvas the caller handed to rtc_canvas_register?
// All of it: the frame's first AND last byte (a canvas registered for fewer bytes than the frame has, or a pointer into
// a smaller pinned allocation, would send every rank's scatter kernel past the mapped range - a GPU memory fault, not an
// error code); anything else takes the gather path.
bool registeredHost(const void* p, size_t bytes) {
  auto mapped = [](const void* q) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, q) != hipSuccess) {
      (void)hipGetLastError();  // (plain pageable memory: not an error)
      return false;
    }
    return attr.type == hipMemoryTypeHost && attr.devicePointer != nullptr;
  };
  if (bytes == 0 || !mapped(p)) return false;
  if (!mapped(static_cast<const char*>(p) + bytes - 1)) return false;
  // one registration (or allocation) from the first byte to the last: device views of both ends are as far apart as the bytes
  void *d0 = nullptr, *d1 = nullptr;
  if (hipHostGetDevicePointer(&d0, const_cast<void*>(p), 0) != hipSuccess ||
      hipHostGetDevicePointer(&d1, const_cast<char*>(static_cast<const char*>(p)) + bytes - 1, 0) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return static_cast<char*>(d1) - static_cast<char*>(d0) == static_cast<ptrdiff_t>(bytes - 1);
}

int finishSlot(rtc_multi* m, uint32_t f, bool may_redeal);

int finishAll(rtc_multi* m, bool may_redeal) {
  int status = RTC_OK;
  for (uint32_t k = 0; k < m->frames; ++k) {  // oldest first
    const int st = finishSlot(m, (m->next_slot + k) % m->frames, may_redeal);
    if (st != RTC_OK && status == RTC_OK) status = st;
  }
  return status;
}

int sizeFor(rtc_multi* m, const rtc_camera& cam) {
  if (cam.hsize == m->hsize && cam.vsize == m->vsize && m->sized) return RTC_OK;
  // (a frame of 2^32 pixels and more does not fit the 32-bit tile arithmetic - and no GPU)
  if (static_cast<uint64_t>(cam.hsize) * cam.vsize >= (1ull << 31))
    return mfail(RTC_ERR_INVALID_ARGUMENT, "image %ux%u is beyond what the tile split indexes", cam.hsize, cam.vsize);
  if (const int st = finishAll(m, false); st != RTC_OK) return st;  // (frames of the old size still in flight)
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
  m->sized = true;
  return RTC_OK;
}

// Re-deal the tiles by what the ranks measured for them in the frame slot f just rendered.  confirm: that frame was the
// first with the current lists (new lists are new pixel maps, so every handle measures them afresh): keep them if the
// fresh costs say they are even, re-deal if not - one wild measurement (a wave that lost its CU for a millisecond in the
// middle of a packet: four virtual ranks on one GPU are four queues the hardware time-slices) must not skew the split
// for good.  The lists and the slot table are shared by the frames in flight: the others are finished first.
int rebalance(rtc_multi* m, uint32_t f, const rtc_camera& cam, bool confirm) {
  std::vector<double> cost(m->n_tiles, 0.0);
  for (uint32_t r = 0; r < m->n; ++r) {
    const std::vector<uint32_t>& mine = m->tiles_of[r];
    if (mine.empty()) continue;
    std::vector<double> c(mine.size());
    M_HIP(hipSetDevice(m->dev[r]));
    if (rtc_get_tile_costs(m->slot[f].scene[r], c.data(), static_cast<uint32_t>(mine.size())) != RTC_OK) return RTC_OK;  // nothing measured: keep the split
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
  if (const int st = finishAll(m, false); st != RTC_OK) return st;  // (every frame that uses the old table is done, everywhere)
  M_RTC(rtc_assign_tiles(cost.data(), m->n_tiles, m->n, m->rank_of.data(), m->slot_of.data()));
  setLists(m);
  M_HIP(hipSetDevice(m->dev[0]));
  M_HIP(hipMemcpy(m->d_slot, m->slot_of.data(), m->n_tiles * sizeof(uint32_t), hipMemcpyHostToDevice));
  if (const int st = uploadLists(m); st != RTC_OK) return st;
  m->max_over_mean = imbalance();
  m->balanced = true;
  m->frames_since_balance = 0;
  m->balance_cam = cam;
  return RTC_OK;
}

int slotStats(rtc_multi* m, uint32_t f, rtc_stats* out) {
  std::memset(out, 0, sizeof *out);
  for (uint32_t r = 0; r < m->n; ++r) {
    if (m->tiles_of.size() > r && m->tiles_of[r].empty()) continue;
    rtc_stats s;
    M_HIP(hipSetDevice(m->dev[r]));
    M_RTC(rtc_get_stats(m->slot[f].scene[r], &s));
    out->primary += s.primary;
    out->secondary += s.secondary;
    out->shadow_calls += s.shadow_calls;
    out->shadow_traced += s.shadow_traced;
    out->overflow += s.overflow;
  }
  return RTC_OK;
}

// What a frame owes once it is done: its streams drained, the overflow check, the re-deal of the tiles by measured cost
// (after the first frame of an image size, and every 16 frames while the camera is not where the split was measured - a
// moving camera measures every frame; the frame after a re-deal confirms it).  Idempotent.
int finishSlot(rtc_multi* m, uint32_t f, bool may_redeal) {
  Slot& S = m->slot[f];
  if (!S.pending) return RTC_OK;
  S.pending = false;
  if (const int st = drainSlot(m, f); st != RTC_OK) return st;
  rtc_stats st;
  if (const int s = slotStats(m, f, &st); s != RTC_OK) return s;
  if (st.overflow) {
    // A csg intersection list that ran out is made longer on the handles that overflowed - rtc_grow_csg_lists sizes it for
    // what the frame needed - and the other slots' handles of the same rank (clones: they share the scene's tables) take the
    // new length over at their next launch (SceneTables::csg_entries_wanted) instead of overflowing and growing one by one;
    // a rank that has not overflowed yet grows when a re-deal hands it the tiles that need it.  The synchronous entry
    // points then render the frame again; the asynchronous ones report this frame's overflow once, and the frames after
    // it have the longer lists.
    bool grown = false, stuck = false;
    for (uint32_t r = 0; r < m->n; ++r) {
      if (m->tiles_of.size() > r && m->tiles_of[r].empty()) continue;
      M_HIP(hipSetDevice(m->dev[r]));
      rtc_stats rs;
      M_RTC(rtc_get_stats(S.scene[r], &rs));
      if (rs.overflow == 0) continue;
      if (rtc_grow_csg_lists(S.scene[r]) == RTC_OK) {
        grown = true;
      } else {
        stuck = true;
      }
    }
    if (grown && !stuck) {
      // the other handles follow: a render with the same lists would overflow there as well.  (They have no overflow of
      // their own to size by: one synchronous pass over them is the frame they render next.)
      m->csg_grown = true;
    }
    return mfail(RTC_ERR_OVERFLOW, "%llu lanes overflowed a per-lane stack or csg list%s", (unsigned long long)st.overflow,
                 (grown && !stuck) ? " (the csg lists have been enlarged: render the frame again)" : "");
  }
  m->frames_since_balance++;
  const rtc_camera cam = S.cam;
  const bool moved = std::memcmp(&cam, &m->balance_cam, sizeof cam) != 0;
  const bool confirm = m->balanced && m->frames_since_balance == 1u;
  if (may_redeal && m->n > 1 && (!m->balanced || (moved && m->frames_since_balance >= 16u) || confirm))
    if (const int s = rebalance(m, f, cam, confirm); s != RTC_OK) return s;
  return RTC_OK;
}

// One frame, enqueued on the next slot: every rank renders its tiles into its compact buffer, ONE gather brings them to
// rank 0 (each rank's send is ordered behind its render), one kernel un-permutes them into the next canvas of the ring
// on device 0, on the slot's stream of rank 0.  Only the frame that last ran on this slot is waited for: with several
// slots the frames before this one are still rendering.
int enqueueFrame(rtc_multi* m, const rtc_camera* cam, uint32_t max_depth, bool rgba8, void** d_out, uint32_t* slot_out,
                 void* host_out = nullptr) {
  if (!m || !cam) return mfail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  if (cam->hsize == 0 || cam->vsize == 0) return mfail(RTC_ERR_INVALID_ARGUMENT, "camera %ux%u", cam->hsize, cam->vsize);
  const uint32_t f = m->next_slot;
  if (const int st = finishSlot(m, f, true); st != RTC_OK) return st;
  if (const int st = sizeFor(m, *cam); st != RTC_OK) return st;
  if (rgba8 && !host_out)
    if (const int st = ensureRgbaBuffers(m); st != RTC_OK) return st;
  Slot& S = m->slot[f];
  const size_t tile_pixels = static_cast<size_t>(m->padded) * kTile * kTile;
  for (uint32_t r = 0; r < m->n; ++r) {
    if (m->tiles_of[r].empty()) continue;
    M_HIP(hipSetDevice(m->dev[r]));
    M_RTC(rtc_render_tile_list_device(S.scene[r], cam, max_depth, kTile, kTile, m->tiles_of[r].data(),
                                      static_cast<uint32_t>(m->tiles_of[r].size()), S.d_buf[r], S.stream[r]));
    if (host_out) {
      // The caller's canvas is registered (pinned, mapped into every GPU): each rank writes its own tiles straight to
      // their places in it over its OWN host link - no gather, no un-permute, nothing through GPU 0's one link (at 4K:
      // 199 MB over one link ~3.8 ms; an eighth of it over each of eight ~0.5 ms).  Camera.render's consumer is a host
      // Canvas (camera.zig:80-102); the *_device forms keep the single RCCL gather.
      void* dev_view = nullptr;
      M_HIP(hipHostGetDevicePointer(&dev_view, host_out, 0));
      const uint32_t mine = static_cast<uint32_t>(m->tiles_of[r].size());
      if (rgba8) {
        M_RTC(rtc_scatter_tile_list_rgba8_device(S.d_buf[r], m->d_list[r], mine, kTile, kTile, cam->hsize, cam->vsize,
                                                 static_cast<uint32_t*>(dev_view), S.stream[r]));
      } else {
        M_RTC(rtc_scatter_tile_list_device(S.d_buf[r], m->d_list[r], mine, kTile, kTile, cam->hsize, cam->vsize,
                                           static_cast<double*>(dev_view), S.stream[r]));
      }
      continue;
    }
    // (an RGBA8 frame is clamped where it was rendered: 4 bytes per pixel go through the gather instead of 24)
    if (rgba8) M_RTC(rtc_rgba8_device(S.d_buf[r], tile_pixels, S.d_rgba_buf[r], S.stream[r]));
  }
  S.pending = true;  // (from here on the streams hold work of this frame)
  S.cam = *cam;
  m->last_slot = f;
  m->next_slot = (f + 1u) % m->frames;
  if (host_out) {  // every rank's stream ends with its own writes into the canvas: finishSlot drains them all
    if (!m->virt)
      for (uint32_t r = 0; r < m->n; ++r) {  // (the `sent` events of a gather that did not happen: nothing to wait for)
        M_HIP(hipSetDevice(m->dev[r]));
        M_HIP(hipEventRecord(S.sent[r], S.stream[r]));
      }
    *d_out = nullptr;
    *slot_out = f;
    return RTC_OK;
  }
  const size_t slab = slabDoubles(m);
  if (!m->virt) {
    for (uint32_t r = 0; r < m->n; ++r) {
      M_HIP(hipSetDevice(m->dev[r]));
      M_HIP(hipEventRecord(S.rendered[r], S.stream[r]));
      M_HIP(hipStreamWaitEvent(m->cstream[r], S.rendered[r], 0));
    }
    M_NCCL(ncclGroupStart());
    for (uint32_t r = 0; r < m->n; ++r) {
      M_HIP(hipSetDevice(m->dev[r]));
      if (rgba8) {
        M_NCCL(ncclGather(S.d_rgba_buf[r], S.d_gathered_rgba, tile_pixels, ncclUint32, 0, m->comm[r], m->cstream[r]));
      } else {
        M_NCCL(ncclGather(S.d_buf[r], S.d_gathered, slab, ncclDouble, 0, m->comm[r], m->cstream[r]));
      }
    }
    M_NCCL(ncclGroupEnd());
    for (uint32_t r = 0; r < m->n; ++r) {
      M_HIP(hipSetDevice(m->dev[r]));
      M_HIP(hipEventRecord(S.sent[r], m->cstream[r]));
    }
    M_HIP(hipSetDevice(m->dev[0]));
    M_HIP(hipEventRecord(S.gathered, m->cstream[0]));
    M_HIP(hipStreamWaitEvent(S.stream[0], S.gathered, 0));
  } else {
    M_HIP(hipSetDevice(m->dev[0]));
    for (uint32_t r = 0; r < m->n; ++r) {
      if (rgba8) {
        M_HIP(hipMemcpyAsync(S.d_gathered_rgba + tile_pixels * r, S.d_rgba_buf[r], tile_pixels * sizeof(uint32_t), hipMemcpyDeviceToDevice, S.stream[r]));
      } else {
        M_HIP(hipMemcpyAsync(S.d_gathered + slab * r, S.d_buf[r], slab * sizeof(double), hipMemcpyDeviceToDevice, S.stream[r]));
      }
      M_HIP(hipEventRecord(S.rendered[r], S.stream[r]));
      if (r != 0) M_HIP(hipStreamWaitEvent(S.stream[0], S.rendered[r], 0));
    }
  }
  M_HIP(hipSetDevice(m->dev[0]));
  if (rgba8) {
    uint32_t* const fb = m->d_rgba[m->cur_rgba];
    m->cur_rgba = (m->cur_rgba + 1u) % static_cast<uint32_t>(m->d_rgba.size());
    M_RTC(rtc_assemble_tile_list_rgba8_device(S.d_gathered_rgba, m->d_slot, kTile, kTile, cam->hsize, cam->vsize, fb, S.stream[0]));
    *d_out = fb;
  } else {
    double* const canvas = m->d_canvas[m->cur];
    m->cur = (m->cur + 1u) % static_cast<uint32_t>(m->d_canvas.size());
    M_RTC(rtc_assemble_tile_list_device(S.d_gathered, m->d_slot, kTile, kTile, cam->hsize, cam->vsize, canvas, S.stream[0]));
    *d_out = canvas;
  }
  *slot_out = f;
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
  const uint32_t frames = std::max(1u, (flags >> 8) & 15u);
  if (frames > kMaxFrames) return mfail(RTC_ERR_INVALID_ARGUMENT, "%u frames in flight asked for, at most %u", frames, kMaxFrames);
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
  m->frames = frames;
  m->virt = virt;
  m->dev.resize(n_gpus);
  for (uint32_t r = 0; r < n_gpus; ++r) m->dev[r] = virt ? 0 : static_cast<int>(r);
  m->slot.resize(frames);
  for (Slot& S : m->slot) {
    S.scene.assign(n_gpus, nullptr);
    S.stream.assign(n_gpus, nullptr);
    S.rendered.assign(n_gpus, nullptr);
    S.sent.assign(n_gpus, nullptr);
  }
  m->tiles_of.assign(n_gpus, {});
  for (uint32_t r = 0; r < n_gpus; ++r) {
    M_HIP(hipSetDevice(m->dev[r]));
    for (uint32_t f = 0; f < frames; ++f) {
      Slot& S = m->slot[f];
      if (f == 0) {
        M_RTC(rtc_scene_create(desc, &S.scene[r]));  // the scene (<= 30 MB) is replicated per GPU ...
      } else {
        M_RTC(rtc_scene_clone(m->slot[0].scene[r], &S.scene[r]));  // ... and shared by the frames in flight on it
      }
      M_HIP(hipStreamCreateWithFlags(&S.stream[r], hipStreamNonBlocking));
      M_HIP(hipEventCreateWithFlags(&S.rendered[r], hipEventDisableTiming));
      M_HIP(hipEventCreateWithFlags(&S.sent[r], hipEventDisableTiming));
      M_HIP(hipEventRecord(S.sent[r], S.stream[r]));  // (never waited for before its first gather otherwise)
      if (r == 0) M_HIP(hipEventCreateWithFlags(&S.gathered, hipEventDisableTiming));
    }
  }
  if (!virt) {
    m->comm.assign(n_gpus, nullptr);
    m->cstream.assign(n_gpus, nullptr);
    for (uint32_t r = 0; r < n_gpus; ++r) {
      M_HIP(hipSetDevice(m->dev[r]));
      M_HIP(hipStreamCreateWithFlags(&m->cstream[r], hipStreamNonBlocking));
    }
    M_NCCL(ncclCommInitAll(m->comm.data(), static_cast<int>(n_gpus), m->dev.data()));
  }
  guard.m = nullptr;
  *out = m;
  return RTC_OK;
}

void rtc_multi_destroy(rtc_multi* m) {
  if (!m) return;
  for (Slot& S : m->slot)
    for (uint32_t r = 0; r < S.stream.size(); ++r) {
      (void)hipSetDevice(m->dev[r]);
      if (S.stream[r]) (void)hipStreamSynchronize(S.stream[r]);
    }
  for (uint32_t r = 0; r < m->cstream.size(); ++r) {
    (void)hipSetDevice(m->dev[r]);
    if (m->cstream[r]) (void)hipStreamSynchronize(m->cstream[r]);
  }
  for (ncclComm_t c : m->comm)
    if (c) (void)ncclCommDestroy(c);
  freeFrameBuffers(m);
  for (uint32_t f = m->slot.size(); f-- > 0u;) {  // (clones before the scene they were made from: either order is allowed)
    Slot& S = m->slot[f];
    for (uint32_t r = 0; r < S.scene.size(); ++r) {
      (void)hipSetDevice(m->dev[r]);
      if (S.scene[r]) rtc_scene_destroy(S.scene[r]);
      if (S.rendered[r]) (void)hipEventDestroy(S.rendered[r]);
      if (S.sent[r]) (void)hipEventDestroy(S.sent[r]);
      if (S.stream[r]) (void)hipStreamDestroy(S.stream[r]);
    }
    if (S.gathered) (void)hipEventDestroy(S.gathered);
  }
  for (uint32_t r = 0; r < m->cstream.size(); ++r) {
    (void)hipSetDevice(m->dev[r]);
    if (m->cstream[r]) (void)hipStreamDestroy(m->cstream[r]);
  }
  delete m;
}

int rtc_multi_render(rtc_multi* m, const rtc_camera* cam, uint32_t max_depth, double* rgb_out) {
  g_multi_error.clear();
  if (!rgb_out) return mfail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  for (int attempt = 0;; ++attempt) {  // (again while a frame's csg lists ran out and could be enlarged: each handle at most six times)
    void* d_canvas = nullptr;
    uint32_t f = 0;
    const size_t frame_bytes = cam ? static_cast<size_t>(cam->hsize) * cam->vsize * 3u * sizeof(double) : 0u;
    void* const direct = registeredHost(rgb_out, frame_bytes) ? rgb_out : nullptr;
    if (const int st = enqueueFrame(m, cam, max_depth, false, &d_canvas, &f, direct); st != RTC_OK) return st;
    if (!direct)  // (a pageable canvas: gathered to GPU 0, one copy over its link)
      M_HIP(hipMemcpyAsync(rgb_out, d_canvas, static_cast<size_t>(cam->hsize) * cam->vsize * 3u * sizeof(double),
                           hipMemcpyDeviceToHost, m->slot[f].stream[0]));
    const int st = finishSlot(m, f, true);
    if (st != RTC_ERR_OVERFLOW || !m->csg_grown || attempt >= 6 * static_cast<int>(m->frames)) return st;
    m->csg_grown = false;
    g_multi_error.clear();
  }
}

int rtc_multi_render_rgba8(rtc_multi* m, const rtc_camera* cam, uint32_t max_depth, uint8_t* rgba_out) {
  g_multi_error.clear();
  if (!rgba_out) return mfail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  for (int attempt = 0;; ++attempt) {
    void* d_fb = nullptr;
    uint32_t f = 0;
    const size_t frame_bytes = cam ? static_cast<size_t>(cam->hsize) * cam->vsize * sizeof(uint32_t) : 0u;
    void* const direct = registeredHost(rgba_out, frame_bytes) ? rgba_out : nullptr;
    if (const int st = enqueueFrame(m, cam, max_depth, true, &d_fb, &f, direct); st != RTC_OK) return st;
    if (!direct)
      M_HIP(hipMemcpyAsync(rgba_out, d_fb, static_cast<size_t>(cam->hsize) * cam->vsize * sizeof(uint32_t), hipMemcpyDeviceToHost,
                           m->slot[f].stream[0]));
    const int st = finishSlot(m, f, true);
    if (st != RTC_ERR_OVERFLOW || !m->csg_grown || attempt >= 6 * static_cast<int>(m->frames)) return st;
    m->csg_grown = false;
    g_multi_error.clear();
  }
}

int rtc_multi_render_rgba8_device(rtc_multi* m, const rtc_camera* cam, uint32_t max_depth, const uint32_t** d_rgba_out) {
  g_multi_error.clear();
  if (!d_rgba_out) return mfail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  void* d_fb = nullptr;
  uint32_t f = 0;
  if (const int st = enqueueFrame(m, cam, max_depth, true, &d_fb, &f); st != RTC_OK) return st;
  *d_rgba_out = static_cast<const uint32_t*>(d_fb);
  return RTC_OK;
}

int rtc_multi_render_device(rtc_multi* m, const rtc_camera* cam, uint32_t max_depth, const double** d_canvas_out) {
  g_multi_error.clear();
  if (!d_canvas_out) return mfail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  void* d_canvas = nullptr;
  uint32_t f = 0;
  if (const int st = enqueueFrame(m, cam, max_depth, false, &d_canvas, &f); st != RTC_OK) return st;
  *d_canvas_out = static_cast<const double*>(d_canvas);
  return RTC_OK;
}

int rtc_multi_synchronize(rtc_multi* m) {
  g_multi_error.clear();
  if (!m) return mfail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  return finishAll(m, true);
}

void* rtc_multi_stream(rtc_multi* m) { return m ? static_cast<void*>(m->slot[m->last_slot].stream[0]) : nullptr; }

int rtc_multi_get_stats(rtc_multi* m, rtc_stats* out) {
  if (!m || !out) return mfail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  return slotStats(m, m->last_slot, out);
}

int rtc_multi_balance(rtc_multi* m, uint32_t* tiles_per_rank, double* max_over_mean) {
  if (!m || !tiles_per_rank || !max_over_mean) return mfail(RTC_ERR_INVALID_ARGUMENT, "null argument");
  for (uint32_t r = 0; r < m->n; ++r) tiles_per_rank[r] = r < m->tiles_of.size() ? static_cast<uint32_t>(m->tiles_of[r].size()) : 0u;
  *max_over_mean = m->max_over_mean;
  return RTC_OK;
}

}  // extern "C"
