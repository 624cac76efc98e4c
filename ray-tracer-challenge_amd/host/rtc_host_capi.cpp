// rtc_host_capi.cpp — C entry points of librtc_host.so for non-C++ callers (the
// Python tests / bench harness use them through ctypes).  Everything here is
// scene loading and output formatting, i.e. the steps either side of the hot
// path; the per-pixel work is reached only through include/rtc.h.
#include <cstring>
#include <memory>
#include <string>

#include "../../include/rtc_host.h"
#include "rtc_api.hpp"
#include "rtc_loader.hpp"

namespace {

thread_local std::string g_error;

struct HostScene {
  rtc::SceneInfo info;
  rtc::FlatScene flat;
  rtc_scene_desc desc;
};

template <typename F>
int guarded(F&& f) {
  try {
    g_error.clear();
    f();
    return 0;
  } catch (const rtc::Error& e) {
    g_error = e.what();
    return 1;
  } catch (const std::exception& e) {
    g_error = std::string("Unexpected: ") + e.what();
    return 2;
  }
}

}  // namespace

extern "C" {

// Error text of the last failing rtch_* call on this thread; begins with the Zig-style error name.
const char* rtch_last_error(void) { return g_error.c_str(); }

// parseScene (scene.zig:612-661) + flatten.  `data_dir` is where from-obj files are read
// from (the reference CLI reads "data/<file>", main.zig:14-21).
int rtch_scene_load(const char* scene_json, const char* data_dir, void** out) {
  return guarded([&] {
    auto hs = std::make_unique<HostScene>();
    hs->info = rtc::parseScene(scene_json, rtc::directoryLoader(data_dir ? data_dir : ""));
    hs->flat = rtc::flattenWorld(hs->info.world);
    hs->desc = hs->flat.desc();
    *out = hs.release();
  });
}

// Threads parseScene may build a scene's objects on (0: what the process may use, at most 16; 1: one loop, as the
// reference).  The scene description does not depend on it.
void rtch_set_loader_threads(uint32_t threads) { rtc::setLoaderThreads(threads); }

void rtch_scene_free(void* h) { delete static_cast<HostScene*>(h); }

const rtc_scene_desc* rtch_scene_desc(void* h) { return &static_cast<HostScene*>(h)->desc; }

// Camera of the scene file; width/height 0 keep the file's values, otherwise they replace
// camera.width/height before Camera.new runs (the reference has no such override).
int rtch_scene_camera(void* h, uint32_t width, uint32_t height, rtc_camera* out) {
  return guarded([&] {
    const rtc::Camera& c0 = static_cast<HostScene*>(h)->info.camera;
    rtc::Camera c = rtc::Camera::create(width ? width : c0.hsize, height ? height : c0.vsize, c0.fov);
    c.setTransform(c0.transform);
    *out = rtc::flattenCamera(c);
  });
}

// lib.zig:166-190 on the scene's own camera (the "preheated" interactive mode): the next rtch_scene_camera
// returns the moved camera; the GPU scene handle is untouched.
int rtch_camera_rotate(void* h, double angle) {
  return guarded([&] { rtc::rotateCamera(static_cast<HostScene*>(h)->info.camera, angle); });
}
int rtch_camera_move(void* h, double distance) {
  return guarded([&] { rtc::moveCamera(static_cast<HostScene*>(h)->info.camera, distance); });
}

// Camera.new + viewTransform for callers that build cameras themselves (camera.zig:33-61).
int rtch_camera_make(uint32_t hsize, uint32_t vsize, double fov, const double from[3], const double to[3],
                     const double up[3], rtc_camera* out) {
  return guarded([&] {
    rtc::Camera c = rtc::Camera::create(hsize, vsize, fov);
    c.setTransform(rtc::Matrix4::viewTransform(rtc::Tuple::point(from[0], from[1], from[2]),
                                               rtc::Tuple::point(to[0], to[1], to[2]),
                                               rtc::Tuple::vec3(up[0], up[1], up[2])));
    *out = rtc::flattenCamera(c);
  });
}

// Canvas.ppm (canvas.zig:181-254) of an [h][w][3] f64 image.  Returns the number of bytes
// needed; writes at most `cap` bytes to `buf`.
size_t rtch_canvas_ppm(const double* rgb, uint32_t w, uint32_t h, char* buf, size_t cap) {
  rtc::Canvas c = rtc::Canvas::create(w, h);
  std::memcpy(static_cast<void*>(c.pixels.data()), rgb, sizeof(double) * 3 * w * h);
  const std::string s = c.ppm();
  if (buf && cap) std::memcpy(buf, s.data(), s.size() < cap ? s.size() : cap);
  return s.size();
}

// lib.zig:146-153: RGBA8 framebuffer, clamp()'d channels, alpha 255.
void rtch_canvas_rgba8(const double* rgb, uint32_t w, uint32_t h, uint8_t* out) {
  for (size_t i = 0; i < static_cast<size_t>(w) * h; ++i) {
    out[4 * i + 0] = rtc::clampChannel(rgb[3 * i + 0]);
    out[4 * i + 1] = rtc::clampChannel(rgb[3 * i + 1]);
    out[4 * i + 2] = rtc::clampChannel(rgb[3 * i + 2]);
    out[4 * i + 3] = 255;
  }
}

// Whole-path convenience for C callers: Camera.render(world) through the GPU library.
int rtch_scene_render(void* h, uint32_t width, uint32_t height, uint32_t max_depth, double* rgb_out) {
  return guarded([&] {
    HostScene* hs = static_cast<HostScene*>(h);
    rtc_camera cam;
    const rtc::Camera& c0 = hs->info.camera;
    rtc::Camera c = rtc::Camera::create(width ? width : c0.hsize, height ? height : c0.vsize, c0.fov);
    c.setTransform(c0.transform);
    cam = rtc::flattenCamera(c);
    rtc_scene* scene = nullptr;
    int st = rtc_scene_create(&hs->desc, &scene);
    if (st == RTC_OK) {
      st = rtc_render(scene, &cam, max_depth, 0, 0, cam.hsize, cam.vsize, rgb_out);
      rtc_scene_destroy(scene);
    }
    if (st != RTC_OK) throw rtc::Error(rtc_status_name(st), rtc_last_error());
  });
}

}  // extern "C"
