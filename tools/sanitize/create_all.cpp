// ASan/UBSan driver for the host half of librtc_hip (validateScene + buildTables: bounds, BVH build, four-wide collapse):
// no GPU here, so rtc_scene_create must come back with NoDevice after all of that has run.
#include <cstdio>
#include <fstream>
#include <sstream>
#include <vector>
#include "rtc.h"
#include "rtc_host.h"
int main(int argc, char** argv) {
  int bad = 0;
  for (int i = 2; i < argc; ++i) {
    std::ifstream f(argv[i]);
    std::stringstream ss; ss << f.rdbuf();
    void* h = nullptr;
    if (rtch_scene_load(ss.str().c_str(), argv[1], &h) != 0) { std::printf("%s: %s\n", argv[i], rtch_last_error()); ++bad; continue; }
    rtc_scene* sc = nullptr;
    const int st = rtc_scene_create(rtch_scene_desc(h), &sc);
    std::printf("%s: rtc_scene_create -> %s (%s)\n", argv[i], rtc_status_name(st), rtc_last_error());
    if (st != RTC_ERR_NO_DEVICE) ++bad;
    rtch_scene_free(h);
  }
  // rtc_assign_tiles (host only): skewed costs, every world size; every tile once, no rank above its buffer
  for (unsigned world = 1; world <= 9; ++world)
    for (unsigned n : {1u, 5u, 64u, 510u}) {
      std::vector<double> cost(n);
      for (unsigned t = 0; t < n; ++t) cost[t] = (t * 2654435761u % 97u) + (t % 13u == 0 ? 4000.0 : 1.0);
      std::vector<unsigned> rank_of(n), slot_of(n), seen((n + world - 1) / world * world, 0);
      if (rtc_assign_tiles(cost.data(), n, world, rank_of.data(), slot_of.data()) != 0) ++bad;
      for (unsigned t = 0; t < n; ++t) {
        if (rank_of[t] >= world || slot_of[t] >= seen.size() || seen[slot_of[t]]++) ++bad;
      }
    }
  std::printf("rtc_assign_tiles: %s\n", bad ? "FAIL" : "ok");
  return bad;
}
