// rtc_api.hpp — host-side mirror of the reference's public interface for the path:
//
//   Camera(T).render(self, allocator, world) !Canvas(T)     camera.zig:80-102
//   Canvas(T).new / getPixelPointer / ppm / fromPpm         canvas.zig:25-46, 132-147, 181-254, 48-121
//   Renderer (scene kept alive between renders)             lib.zig:41-190 ("preheated")
//
// Same names and argument meaning as the reference; the body of render() is a
// call through the C ABI (include/rtc.h) into the HIP kernels.  There is no CPU
// implementation of the per-pixel path in this library: if the HIP library or a
// GPU is missing, render() throws.
#pragma once
#include <string>
#include <vector>

#include "rtc_flatten.hpp"
#include "rtc_scene.hpp"

struct rtc_scene;

namespace rtc {

struct Canvas {  // canvas.zig:16-22
  size_t width = 0, height = 0;
  std::vector<Color> pixels;  // row-major, y*width + x (canvas.zig:132-137)

  static Canvas create(size_t width, size_t height) {  // Canvas.new: zero-filled
    Canvas c;
    c.width = width;
    c.height = height;
    c.pixels.assign(width * height, Color{0.0, 0.0, 0.0});
    return c;
  }
  const Color* getPixelPointer(size_t x, size_t y) const {  // canvas.zig:132-139: null when out of range
    if (x >= width || y >= height) return nullptr;
    return &pixels[y * width + x];
  }
  Color* getPixelPointerMut(size_t x, size_t y) {
    if (x >= width || y >= height) return nullptr;
    return &pixels[y * width + x];
  }
  std::string ppm() const;                 // canvas.zig:181-254 — P3, 70-column wrap
  // canvas.zig:48-121 — a P3 file back into a canvas; throws Error named like the reference's ParseError /
  // parseInt / parseFloat errors (InvalidMagicNumber, InvalidDimensions, InvalidScale, InvalidCharacter, Overflow)
  static Canvas fromPpm(const std::string& ppm);
  std::vector<uint8_t> rgba8() const;      // lib.zig:146-153 — clamp()'d RGBA, alpha 255
};

// Scene resident in HBM; many renders per upload (lib.zig Renderer / "WASM preheated").
class Renderer {
 public:
  explicit Renderer(const World& world);
  ~Renderer();
  Renderer(const Renderer&) = delete;
  Renderer& operator=(const Renderer&) = delete;

  // recursion depth: the reference hard-codes 5 (camera.zig:118, lib.zig:144)
  Canvas render(const Camera& camera, unsigned max_depth = 5);
  const FlatScene& flat() const { return flat_; }

 private:
  FlatScene flat_;
  rtc_scene* scene_ = nullptr;
};

// The interactive mode of the WASM library (lib.zig:166-190): the scene stays loaded (one Renderer, one
// rtc_scene on the GPU) while the camera orbits its target or moves along its line of sight; each call
// updates the camera's transform and its saved from/to/up exactly like Renderer.rotateCamera / moveCamera.
void rotateCamera(Camera& camera, double angle);   // lib.zig:166-178
void moveCamera(Camera& camera, double distance);  // lib.zig:180-190

// Camera.render(world): upload, render at depth 5, download.
Canvas render(const Camera& camera, const World& world);

}  // namespace rtc
