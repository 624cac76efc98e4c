// ASan/UBSan driver for the host half of librtc_hip (validateScene + buildTables: bounds, BVH build, four-wide collapse):
// no GPU here, so rtc_scene_create must come back with NoDevice after all of that has run.
#include <cstdio>
#include <fstream>
#include <sstream>
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
  return bad;
}
