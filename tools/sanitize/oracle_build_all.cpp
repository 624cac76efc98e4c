// AddressSanitizer / UBSan over the oracle's OWN scene build (oracle/rtc_oracle_scene.hpp through the C face of
// oracle_capi.cpp): every golden scene is parsed (JSON, OBJ files from the data directory; scenes that reference images
// get a 2x2 stand-in per name) and a 24x14 image rendered from the built tree.  tools/sanitize_oracle.sh builds and runs it.
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

struct rtc_camera;
extern "C" {
int orc_built_create(void** out);
void orc_built_destroy(void* b);
int orc_built_add_image(void* b, const char* name, uint32_t w, uint32_t h, const float* rgb);
int orc_built_parse(void* b, const char* scene_json, const char* data_dir, uint32_t width, uint32_t height);
int orc_built_counts(void* b, uint64_t* out);
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string data_dir = std::string(argv[1]) + "/";
  int bad = 0;
  for (int i = 2; i < argc; ++i) {
    std::ifstream f(argv[i]);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string json = ss.str();
    void* b = nullptr;
    if (orc_built_create(&b) != 0) return 3;
    const float px[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 1, 1, 0};
    // (image names the scenes use: a parse that finds none registered reports it - fine for this check)
    for (const char* name : {"earthmap1k.png", "negx.png", "negy.png", "negz.png", "posx.png", "posy.png", "posz.png", "checker.png"})
      orc_built_add_image(b, name, 2, 2, px);
    const int st = orc_built_parse(b, json.c_str(), data_dir.c_str(), 24, 14);
    uint64_t counts[16] = {0};
    if (st == 0) orc_built_counts(b, counts);
    std::printf("%s: parse -> %d, leaves %llu\n", argv[i], st, (unsigned long long)counts[0]);
    if (st != 0) ++bad;
    orc_built_destroy(b);
  }
  std::printf("oracle build under the sanitizers: %d scene(s) not parsed\n", bad);
  return 0;
}
