// rtc_schedule.h — the host's part of the order in which the persistent waves are handed pixels (DESIGN.md section 3,
// "Schedule"): the schedule buffers, and the packing with chunks cut into runs of pixels (packSchedule), for launches
// where some chunk takes much longer than a wave's fair share.  Everything else - the first frame's estimate, the packing
// of whole chunks from measured times - runs on the device (rtc_kernels.hip).  Never affects results.
#pragma once
#include "rtc_host_internal.h"
#include "rtc_bounds.h"

namespace {

inline uint32_t scheduleItem(uint32_t chunk, uint32_t start, uint32_t len) {
  return chunk | (start << 20) | ((len - 1u) << 26);
}

// Both schedule buffers hold at least `words` words (a device-packed schedule needs 16 per chunk; a host schedule with
// chunks cut into runs can be longer).  Growing drops what the buffers held.
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

// Copies s->h_order into the buffer that is not in use and makes it the current one (stream order: launches already
// enqueued keep reading the other buffer).
int uploadSchedule(rtc_scene* s, const DevPixelMap& map, hipStream_t stream) {
  const std::vector<uint32_t>& order = s->h_order;
  if (const int st = ensureScheduleBuffers(s, std::max(order.size(), static_cast<size_t>(map.n_chunks) * RTC_PACKET_ITEMS)); st != RTC_OK) return st;
  const uint32_t target = s->sched_cur ^ 1u;
  HIP_TRY(hipMemcpyAsync(s->d_sched[target], order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
  s->sched_cur = target;
  s->sched_on_device = false;
  s->sched_n_units = static_cast<uint32_t>(order.size() / RTC_PACKET_ITEMS);
  return RTC_OK;
}

// Schedule from the MEASURED per-pixel ray counts of an earlier frame with the same pixel map.
//
// One wave's fair share of the frame is F = total rays / resident waves.  A chunk that costs more than
// cap = alpha * F would BE the critical path if one wave had to run it alone (a frame split over several
// GPUs leaves each of them few chunks per wave; measured: tools/scale_sim.py), so it is cut into
// rows, and rows into shorter runs, until every run costs <= cap.  The runs are dealt longest-first into
// bins of <= 64 pixels whose cost stays near cap; a bin is then topped up to 64 pixels with rows of the
// cheapest chunks of the frame (one or two rays per pixel), so the wave that pulls it starts with all lanes
// busy and the cheap pixels' lanes become free just as the expensive pixels' ray trees fan out (the
// kernel's intra-wave sharing moves the sub-trees over).  Every other chunk stays whole: neighbouring
// pixels in one wave is what keeps the traversal coherent.  Packets go out most expensive first.
// (Host statement of step 1 of rtc_pack_kernel, which is what the library runs; the packer fuzz checks this one.)
// Per-chunk wave time from the per-packet times of a measured launch (DevPixelMap::packet_time): a packet's time is
// shared among its items in proportion to their cost (a partial item: its share of the chunk's cost by pixel count).
// Chunks that were not timed (time 0: e.g. a launch that measured nothing) fall back to their cost.
[[maybe_unused]] std::vector<uint32_t> chunkTimes(const DevPixelMap& map, const std::vector<uint32_t>& chunk_cost,
                                 const std::vector<uint32_t>& packet_time, const std::vector<uint32_t>& measured_order) {
  std::vector<double> t(map.n_chunks, 0.0);
  if (measured_order.empty()) {
    for (uint32_t c = 0; c < map.n_chunks && c < packet_time.size(); ++c) t[c] = packet_time[c];
  } else {
    const size_t n_packets = measured_order.size() / RTC_PACKET_ITEMS;
    for (size_t p = 0; p < n_packets && p < packet_time.size(); ++p) {
      double w[RTC_PACKET_ITEMS], sum = 0.0;
      uint32_t chunk[RTC_PACKET_ITEMS], n = 0;
      for (uint32_t i = 0; i < RTC_PACKET_ITEMS; ++i) {
        const uint32_t it = measured_order[p * RTC_PACKET_ITEMS + i];
        if (it == RTC_NO_ITEM) continue;
        const uint32_t c = it & 0xFFFFFu, len = (it >> 26) + 1u;
        if (c >= map.n_chunks) continue;
        chunk[n] = c;
        w[n] = (static_cast<double>(chunk_cost[c]) + 1.0) * len / 64.0;
        sum += w[n];
        ++n;
      }
      for (uint32_t i = 0; i < n; ++i) t[chunk[i]] += packet_time[p] * w[i] / sum;
    }
  }
  double total_t = 0.0, total_c = 0.0;
  for (uint32_t c = 0; c < map.n_chunks; ++c) {
    total_t += t[c];
    total_c += chunk_cost[c];
  }
  const double per_cost = total_c > 0.0 && total_t > 0.0 ? total_t / total_c : 1.0;
  std::vector<uint32_t> out(map.n_chunks);
  for (uint32_t c = 0; c < map.n_chunks; ++c) {
    const double v = t[c] > 0.0 ? t[c] : chunk_cost[c] * per_cost;
    out[c] = static_cast<uint32_t>(std::min(v, 4.0e9));
  }
  return out;
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

// The common case of packSchedule below, from per-chunk sums alone: no chunk costs more than a wave's fair share,
// so every packet is one whole chunk, most expensive first.  Returns false if some chunk has to be split.
// The library itself packs this case ON THE DEVICE (rtc_pack_kernel, same policy: classes of a quarter octave, image
// order inside a class, cheap chunks several to a packet); this host statement of it is what the packer fuzz
// (tools/sanitize/pack_fuzz.hip) and packSchedule's readers go by.
[[maybe_unused]] bool packWholeChunks(rtc_scene* s, const DevPixelMap& map, const std::vector<uint32_t>& chunk_cost, double n_waves) {
  static const double alpha = getenv("RTC_SPLIT_ALPHA") ? atof(getenv("RTC_SPLIT_ALPHA")) : 1.0;
  double total = 0.0;
  uint32_t heaviest = 0;
  for (uint32_t c : chunk_cost) {
    total += c;
    heaviest = std::max(heaviest, c);
  }
  const double cap = std::max(1.0, alpha * total / std::max(1.0, n_waves));
  if (static_cast<double>(heaviest) > cap) return false;
  std::vector<uint32_t> order(map.n_chunks);
  for (uint32_t i = 0; i < map.n_chunks; ++i) order[i] = i;
  // Longest first, but in classes of about equal length (a quarter octave) that keep the chunks' image order: waves
  // that run at the same moment then work on neighbouring chunks (measured: sorting strictly by time scatters the
  // cheap chunks of the tail over the image and they take 2-8 times longer each than in image order).
  static const bool strict = getenv("RTC_SCHED_STRICT") != nullptr;  // experiment knob
  // Below `flat` of a wave's fair share the order no longer matters for the balance of the frame, but the neighbourhood
  // does (mesh scenes: a chunk among its image neighbours finds its BVH nodes and triangles in cache): one class.
  static const double flat = getenv("RTC_SCHED_FLAT") ? atof(getenv("RTC_SCHED_FLAT")) : 0.0;
  const double flat_below = flat * total / std::max(1.0, n_waves);
  std::vector<int> klass(map.n_chunks);  // once per chunk, not once per comparison (32 400 chunks at 1080p: 8 ms -> under 2)
  for (uint32_t c = 0; c < map.n_chunks; ++c)
    klass[c] = strict ? static_cast<int>(chunk_cost[c])
                      : static_cast<int>(4.0 * std::log2(1.0 + std::max(static_cast<double>(chunk_cost[c]), flat_below)));
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return klass[a] > klass[b]; });
  // Cheap chunks travel several to a packet (up to 16, up to 1/`group` of a wave's fair share): when every wave
  // reaches the cheap end of the list at the same moment, one-chunk packets of a few microseconds each turn the
  // work counter and the memory system into the bottleneck (measured: the last 2 % of the schedule took 8 times
  // longer per chunk than the same chunks took when only a few waves were pulling them).
  static const double group = getenv("RTC_SCHED_GROUP") ? atof(getenv("RTC_SCHED_GROUP")) : 32.0;
  const double t_min = groupFloor(s);
  const double group_cap = group > 0.0 ? std::max(total / std::max(1.0, n_waves) / group, t_min) : 0.0;
  s->h_order.clear();
  s->h_order.reserve(static_cast<size_t>(map.n_chunks) * 4u);
  uint32_t n_packets = 0;
  for (uint32_t i = 0; i < map.n_chunks;) {
    double sum = 0.0;
    uint32_t n = 0;
    while (i < map.n_chunks && n < RTC_PACKET_ITEMS && (n == 0 || sum + chunk_cost[order[i]] <= group_cap)) {
      sum += chunk_cost[order[i]];
      s->h_order.push_back(scheduleItem(order[i], 0, 64));
      ++i;
      ++n;
    }
    for (; n < RTC_PACKET_ITEMS; ++n) s->h_order.push_back(RTC_NO_ITEM);
    ++n_packets;
  }
  if (getenv("RTC_PROFILE_DUMP"))
    std::fprintf(stderr, "rtc schedule: %u whole chunks in %u packets, heaviest %u, cap %.0f, total %.0f\n", map.n_chunks, n_packets,
                 heaviest, cap, total);
  if (getenv("RTC_PROFILE_DUMP")) {
    std::fprintf(stderr, "rtc schedule times by position:");
    for (double f : {0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 0.97, 0.98, 0.99, 0.995, 1.0}) {
      const uint32_t i = std::min<uint32_t>(map.n_chunks - 1, static_cast<uint32_t>(f * map.n_chunks));
      std::fprintf(stderr, " %.3f:%u(c%u)", f, chunk_cost[order[i]], order[i]);
    }
    std::fprintf(stderr, "\n");
  }
  return true;
}

void packSchedule(const rtc_scene* s, const DevPixelMap& map, const std::vector<uint32_t>& cost,
                  const std::vector<uint32_t>& chunk_cost_sum, const std::vector<uint32_t>& chunk_time, double n_waves,
                  uint32_t max_depth, std::vector<uint32_t>& out) {
  static const double alpha = getenv("RTC_SPLIT_ALPHA") ? atof(getenv("RTC_SPLIT_ALPHA")) : 1.0;
  struct Item { uint32_t cost, code; };
  const uint32_t n_chunks = map.n_chunks;
  std::vector<uint32_t> pc(static_cast<size_t>(n_chunks) * 64u);  // per chunk, its pixels' shares of the chunk's time
  std::vector<float> rays(static_cast<size_t>(n_chunks) * 64u);   // per chunk, the rays of its pixels' trees (about)
  std::vector<uint64_t> chunk_cost(n_chunks, 0);
  double total = 0.0;
  for (uint32_t c = 0; c < n_chunks; ++c) {
    const uint32_t region = c / map.chunks_per_region, cr = c - region * map.chunks_per_region;
    const uint32_t ccy = cr / map.chunks_x;
    const uint32_t rx0 = (cr - ccy * map.chunks_x) * 8u, ry0 = ccy * 8u;
    const uint32_t w = map.mode == 0u ? map.w : map.tile_w, h = map.mode == 0u ? map.h : map.tile_h;
    const size_t out0 = map.mode == 0u ? 0 : static_cast<size_t>(region) * map.tile_h * map.tile_w;
    // a pixel's share of the chunk's measured TIME, by its share of the chunk's cost
    const double scale = chunk_cost_sum[c] ? static_cast<double>(chunk_time[c]) / chunk_cost_sum[c] : 0.0;
    for (uint32_t k = 0; k < 64u; ++k) {
      const uint32_t rx = rx0 + (k & 7u), ry = ry0 + (k >> 3);
      if (rx >= w || ry >= h) continue;
      const uint32_t px_cost = cost[out0 + static_cast<size_t>(ry) * w + rx];
      const uint32_t v = static_cast<uint32_t>(px_cost * scale);
      pc[static_cast<size_t>(c) * 64u + k] = v;
      chunk_cost[c] += v;
      rays[static_cast<size_t>(c) * 64u + k] = std::max(1.0f, px_cost / 5.0f);  // cost: 2 per closest-hit or containers trace, 1 per shadow ray
    }
    total += static_cast<double>(chunk_cost[c]);
  }
  // Depth-aware splitting.  A wave renders one ray per lane per iteration, and the rays of one pixel's tree depend on
  // each other level by level: pixels [a, b) of a chunk take about L + S iterations, L = the deepest tree among them
  // (at most max_depth + 1 levels, at most the rays it has), S = their rays / 64.  A chunk measured at T > F (a wave's
  // fair share) is cut into r parts of S / r each; every part pays L again, so r stops where a part's S would fall
  // under L / 2 (a chunk that is all depth stays whole), and F includes what the cuts add (fixed point, a few rounds).
  const double level_cap = static_cast<double>(max_depth) + 1.0;
  auto iterations = [&](uint32_t c, uint32_t a, uint32_t b, double* depth = nullptr) {
    double L = 0.0, S = 0.0;
    for (uint32_t i = a; i < b; ++i) {
      const double r = rays[static_cast<size_t>(c) * 64u + i];
      L = std::max(L, std::min(r, level_cap));
      S += r / 64.0;
    }
    if (depth) *depth = L;
    return L + S;
  };
  std::vector<uint8_t> parts(n_chunks, 1);
  double fair = total / std::max(1.0, n_waves);
  for (int round = 0; round < 6; ++round) {
    double extra = 0.0;
    for (uint32_t c = 0; c < n_chunks; ++c) {
      const double T = static_cast<double>(chunk_cost[c]);
      uint32_t r = 1;
      if (T > alpha * fair) {
        double L = 0.0;
        const double it = iterations(c, 0, 64, &L), per_iteration = T / std::max(it, 1e-9), S = it - L;
        const double room = std::max(alpha * fair / per_iteration - L, std::max(0.5 * L, 0.5));
        r = std::max(1u, std::min(16u, static_cast<uint32_t>(std::ceil(S / room))));
        extra += (r - 1u) * L * per_iteration;
      }
      parts[c] = static_cast<uint8_t>(r);
    }
    fair = (total + extra) / std::max(1.0, n_waves);
  }
  const double cap = std::max(1.0, alpha * fair);
  std::vector<float> inflation(n_chunks, 1.0f);  // per chunk: modelled time of its parts / its time whole
  std::vector<Item> whole;
  struct Part { uint64_t est; uint32_t code, npx; };
  std::vector<Part> split_parts;
  for (uint32_t c = 0; c < n_chunks; ++c) {
    if (parts[c] == 1u) {
      whole.push_back({static_cast<uint32_t>(chunk_cost[c]), scheduleItem(c, 0, 64)});
      continue;
    }
    // r contiguous pixel ranges of about equal ray counts, in the chunk's row-major order: one item each
    const float* k = &rays[static_cast<size_t>(c) * 64u];
    const double T = static_cast<double>(chunk_cost[c]), per_iteration = T / std::max(iterations(c, 0, 64), 1e-9);
    double S = 0.0;
    for (uint32_t i = 0; i < 64u; ++i) S += k[i];
    double done = 0.0, est_sum = 0.0;
    uint32_t start = 0, made = 0;
    for (uint32_t i = 0; i < 64u; ++i) {
      done += k[i];
      if (i == 63u || (made + 1u < parts[c] && done >= S * (made + 1u) / parts[c])) {
        const double est = per_iteration * iterations(c, start, i + 1u);
        split_parts.push_back({static_cast<uint64_t>(est), scheduleItem(c, start, i + 1u - start), i + 1u - start});
        est_sum += est;
        start = i + 1u;
        ++made;
      }
    }
    inflation[c] = static_cast<float>(std::max(1.0, est_sum / std::max(T, 1.0)));
  }
  out.clear();
  struct Packet { uint64_t cost; uint32_t npx, n_items; uint32_t items[RTC_PACKET_ITEMS]; };
  std::vector<Packet> packets;
  std::sort(whole.begin(), whole.end(), [](const Item& a, const Item& b) { return a.cost > b.cost; });
  size_t light_end = whole.size();  // whole[light_end..] have been cut up as filler
  {
    // one part per packet, topped up with rows of the cheapest chunks: their pixels are done after an iteration or
    // two, just when the part's ray trees fan out and need the lanes
    static const bool no_fill = getenv("RTC_SPLIT_NOFILL") != nullptr;  // experiment knob
    uint32_t filler_chunk = 0, filler_row = 8;
    for (const Part& pt : split_parts) {
      Packet P{pt.est, pt.npx, 1, {}};
      P.items[0] = pt.code;
      while (!no_fill && P.npx + 8u <= 64u && P.n_items < RTC_PACKET_ITEMS) {
        if (filler_row == 8u) {
          if (light_end == 0 || static_cast<double>(whole[light_end - 1].cost) > 0.25 * cap) break;
          --light_end;
          filler_chunk = whole[light_end].code & 0xFFFFFu;
          filler_row = 0;
        }
        const uint32_t* k = &pc[static_cast<size_t>(filler_chunk) * 64u + filler_row * 8u];
        uint32_t rc = 0;
        for (int i = 0; i < 8; ++i) rc += k[i];
        P.items[P.n_items++] = scheduleItem(filler_chunk, filler_row * 8u, 8);
        P.npx += 8u;
        P.cost += rc;
        ++filler_row;
      }
      packets.push_back(P);
    }
    if (filler_row < 8u) {  // rows of a filler chunk that no packet took
      Packet P{0, 0, 0, {}};
      for (; filler_row < 8u; ++filler_row) {
        P.items[P.n_items++] = scheduleItem(filler_chunk, filler_row * 8u, 8);
        P.npx += 8u;
        const uint32_t* k = &pc[static_cast<size_t>(filler_chunk) * 64u + filler_row * 8u];
        for (int i = 0; i < 8; ++i) P.cost += k[i];
      }
      packets.push_back(P);
    }
  }
  {  // the chunks that stay whole: cheap ones several to a packet, as in packWholeChunks
    static const double group = getenv("RTC_SCHED_GROUP") ? atof(getenv("RTC_SCHED_GROUP")) : 32.0;
    const double t_min = groupFloor(s);
    const double group_cap = group > 0.0 ? std::max(total / std::max(1.0, n_waves) / group, t_min) : 0.0;
    for (size_t i = 0; i < light_end;) {
      Packet P{0, 0u, 0, {}};
      while (i < light_end && P.n_items < RTC_PACKET_ITEMS && (P.n_items == 0 || static_cast<double>(P.cost + whole[i].cost) <= group_cap)) {
        P.items[P.n_items++] = whole[i].code;
        P.cost += whole[i].cost;
        P.npx += 64u;
        ++i;
      }
      packets.push_back(P);
    }
  }
  packets.erase(std::remove_if(packets.begin(), packets.end(), [](const Packet& P) { return P.n_items == 0; }), packets.end());
  std::stable_sort(packets.begin(), packets.end(), [](const Packet& a, const Packet& b) { return a.cost > b.cost; });
  out.assign(packets.size() * RTC_PACKET_ITEMS, RTC_NO_ITEM);
  for (size_t i = 0; i < packets.size(); ++i)
    for (uint32_t j = 0; j < packets[i].n_items; ++j) out[i * RTC_PACKET_ITEMS + j] = packets[i].items[j];
  // every pixel of every chunk exactly once, whatever the packing did: otherwise fall back to whole chunks
  {
    std::vector<uint8_t> seen(static_cast<size_t>(n_chunks) * 64u, 0);
    bool ok = true;
    size_t covered = 0;
    for (uint32_t it : out) {
      if (it == RTC_NO_ITEM) continue;
      const uint32_t c = it & 0xFFFFFu, start = (it >> 20) & 63u, len = (it >> 26) + 1u;
      if (c >= n_chunks || start + len > 64u) {
        ok = false;
        break;
      }
      for (uint32_t k = start; k < start + len; ++k) {
        if (seen[static_cast<size_t>(c) * 64u + k]++) ok = false;
        ++covered;
      }
    }
    if (!ok || covered != seen.size()) {
      std::fprintf(stderr, "rtc: schedule packing lost or duplicated pixels (%zu of %zu): using whole chunks\n", covered, seen.size());
      out.assign(static_cast<size_t>(n_chunks) * RTC_PACKET_ITEMS, RTC_NO_ITEM);
      for (uint32_t c = 0; c < n_chunks; ++c) out[static_cast<size_t>(c) * RTC_PACKET_ITEMS] = scheduleItem(c, 0, 64);
    }
  }
  if (getenv("RTC_PROFILE_DUMP")) {
    std::vector<uint64_t> pcst;
    uint64_t hpx[5] = {0, 0, 0, 0, 0}, hit[5] = {0, 0, 0, 0, 0};
    for (const Packet& P : packets) {
      pcst.push_back(P.cost);
      hpx[std::min(4u, P.npx / 16u)]++;
      hit[P.n_items == 1 ? 0 : 1 + (P.n_items - 1) / 5]++;
    }
    std::sort(pcst.begin(), pcst.end());
    auto q = [&](double f) { return pcst.empty() ? 0ull : (unsigned long long)pcst[std::min(pcst.size() - 1, (size_t)(f * pcst.size()))]; };
    std::fprintf(stderr, "rtc packets: cost min %llu p10 %llu med %llu p90 %llu p99 %llu max %llu | px<16 %llu <32 %llu <48 %llu <64 %llu =64 %llu | items 1: %llu 2-5: %llu 6-10: %llu 11-15: %llu 16: %llu\n",
                 q(0), q(0.1), q(0.5), q(0.9), q(0.99), q(1.0), (unsigned long long)hpx[0], (unsigned long long)hpx[1], (unsigned long long)hpx[2], (unsigned long long)hpx[3], (unsigned long long)hpx[4],
                 (unsigned long long)hit[0], (unsigned long long)hit[1], (unsigned long long)hit[2], (unsigned long long)hit[3], (unsigned long long)hit[4]);
  }
  if (getenv("RTC_PROFILE_DUMP"))
    std::fprintf(stderr, "rtc schedule: %zu packets (%zu parts of %zu split chunks, %zu filler chunks), cap %.0f, total %.0f\n",
                 packets.size(), split_parts.size(), static_cast<size_t>(n_chunks) - whole.size(), whole.size() - light_end, cap, total);
}

}  // namespace
