// rtc_schedule.h — the host's part of the order in which the persistent waves are handed pixels (DESIGN.md section 3,
// "Schedule"): the schedule buffers and one tuning constant.  The schedule itself - the first frame's estimate, the
// packing from measured times, the cutting of heavy chunks into runs of pixels - is made on the device (rtc_kernels.hip).
// Never affects results.
#pragma once
#include "rtc_host_internal.h"
#include "rtc_bounds.h"

namespace {

// Both schedule buffers hold at least `words` words (16 per packet; maxPackets() of them).  Growing drops what the
// buffers held.
int ensureScheduleBuffers(rtc_scene* s, size_t words) {
  if (words <= s->sched_capacity) return RTC_OK;
  HIP_TRY(hipEventSynchronize(s->launch_done));  // (the last launch may still be reading a buffer)
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
inline double groupFloor(const rtc_scene* s) {
  static const double forced = getenv("RTC_SCHED_TMIN") ? atof(getenv("RTC_SCHED_TMIN")) : 0.0;
  return forced > 0.0 ? forced : (s->ext_kernel ? 12000.0 : 8000.0);
}

}  // namespace
