// rtc_api.cpp — Camera.render / Renderer / Canvas on top of the C ABI.
#include "rtc_api.hpp"

#include <cstdio>

namespace rtc {

namespace {
[[noreturn]] void raise(int status) { throw Error(rtc_status_name(status), rtc_last_error()); }
}  // namespace

Renderer::Renderer(const World& world) : flat_(flattenWorld(world)) {
  const rtc_scene_desc d = flat_.desc();
  const int st = rtc_scene_create(&d, &scene_);
  if (st != RTC_OK) raise(st);
}

Renderer::~Renderer() { rtc_scene_destroy(scene_); }

Canvas Renderer::render(const Camera& camera, unsigned max_depth) {
  Canvas image = Canvas::create(camera.hsize, camera.vsize);  // camera.zig:81
  const rtc_camera cam = flattenCamera(camera);
  static_assert(sizeof(Color) == 3 * sizeof(double), "Canvas pixels are packed rgb f64");
  const int st = rtc_render(scene_, &cam, max_depth, 0, 0, cam.hsize, cam.vsize,
                            reinterpret_cast<double*>(image.pixels.data()));
  if (st != RTC_OK) raise(st);
  return image;
}

void rotateCamera(Camera& camera, double angle) {  // lib.zig:166-178
  Tuple from = camera.saved_from;
  const Tuple to = camera.saved_to, up = camera.saved_up;
  const Tuple delta = Tuple::point(0.0, 0.0, 0.0).sub(to);
  from = from.add(delta);
  from = Matrix4::identity().rotate(up, angle).tupleMul(from);
  from = from.sub(delta);
  camera.setTransform(Matrix4::viewTransform(from, to, up));
  camera.saved_from = from;
}

void moveCamera(Camera& camera, double distance) {  // lib.zig:180-190
  Tuple from = camera.saved_from;
  const Tuple to = camera.saved_to, up = camera.saved_up;
  const Tuple delta = to.sub(from).mul(distance);
  from = from.add(delta);
  camera.setTransform(Matrix4::viewTransform(from, to, up));
  camera.saved_from = from;
}

Canvas render(const Camera& camera, const World& world) {
  Renderer r(world);
  return r.render(camera, 5);
}

// canvas.zig:181-254, restated literally (including how the 70-column rule treats
// the red channel differently from green and blue).
std::string Canvas::ppm() const {
  std::string str;
  str.reserve(width * height * 12);
  char scratch[32];
  str += "P3\n";
  std::snprintf(scratch, sizeof scratch, "%zu %zu\n", width, height);
  str += scratch;
  str += "255\n";
  size_t col = 0;
  for (size_t i = 0; i < pixels.size(); ++i) {
    const Color& pixel = pixels[i];
    int len = std::snprintf(scratch, sizeof scratch, "%u", static_cast<unsigned>(clampChannel(pixel.r)));
    if (col + len >= 70) {
      str += '\n';
      col = 0;
    } else if (i % width != 0) {
      str += ' ';
      col += 1;
    }
    str += scratch;
    col += len;

    len = std::snprintf(scratch, sizeof scratch, "%u", static_cast<unsigned>(clampChannel(pixel.g)));
    if (col + len >= 70) {
      str += '\n';
      col = 0;
    } else {
      str += ' ';
      col += 1;
    }
    str += scratch;
    col += len;

    len = std::snprintf(scratch, sizeof scratch, "%u", static_cast<unsigned>(clampChannel(pixel.b)));
    if (col + len >= 70) {
      str += '\n';
      col = 0;
    } else {
      str += ' ';
      col += 1;
    }
    str += scratch;
    col += len;

    if ((i + 1) % width == 0) {
      str += '\n';
      col = 0;
    }
  }
  return str;
}

std::vector<uint8_t> Canvas::rgba8() const {  // lib.zig:146-153
  std::vector<uint8_t> out(pixels.size() * 4);
  for (size_t i = 0; i < pixels.size(); ++i) {
    out[4 * i + 0] = clampChannel(pixels[i].r);
    out[4 * i + 1] = clampChannel(pixels[i].g);
    out[4 * i + 2] = clampChannel(pixels[i].b);
    out[4 * i + 3] = 255;
  }
  return out;
}

}  // namespace rtc
