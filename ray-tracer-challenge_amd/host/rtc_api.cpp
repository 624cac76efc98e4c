// rtc_api.cpp — Camera.render / Renderer / Canvas on top of the C ABI.
#include "rtc_api.hpp"

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <string_view>

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

// canvas.zig:48-121.  Lines and fields are std.mem.tokenizeScalar's (one delimiter byte, empty tokens skipped); a line
// that begins with '#' is skipped before the dimensions and before the scale, and among the pixel data when its first
// field begins with '#'; a colour may span lines; the sample count must come out as width x height.
namespace {
struct Fields {  // tokenizeScalar
  std::string_view rest;
  char delim;
  bool next(std::string_view& out) {
    size_t i = 0;
    while (i < rest.size() && rest[i] == delim) ++i;
    size_t j = i;
    while (j < rest.size() && rest[j] != delim) ++j;
    if (j == i) return false;
    out = rest.substr(i, j - i);
    rest = rest.substr(j);
    return true;
  }
};
size_t ppmUsize(std::string_view tok) {  // std.fmt.parseInt(usize, tok, 10)
  size_t i = (!tok.empty() && tok[0] == '+') ? 1 : 0, v = 0;
  if (i >= tok.size()) throw Error("InvalidCharacter", "ppm number");
  for (; i < tok.size(); ++i) {
    if (tok[i] < '0' || tok[i] > '9') throw Error("InvalidCharacter", "ppm number");
    const size_t nv = v * 10 + static_cast<size_t>(tok[i] - '0');
    if (nv < v) throw Error("Overflow", "ppm number");
    v = nv;
  }
  return v;
}
double ppmFloat(std::string_view tok) {  // std.fmt.parseFloat
  const std::string text(tok);
  char* end = nullptr;
  errno = 0;
  const double v = std::strtod(text.c_str(), &end);
  if (text.empty() || end != text.c_str() + text.size()) throw Error("InvalidCharacter", "ppm sample");
  return v;
}
}  // namespace

Canvas Canvas::fromPpm(const std::string& ppm) {
  Fields lines{ppm, '\n'};
  std::string_view line;
  if (!lines.next(line) || line != "P3") throw Error("InvalidMagicNumber");
  auto header_line = [&](const char* error) {
    do {
      if (!lines.next(line)) throw Error(error);
    } while (!line.empty() && line[0] == '#');
    return Fields{line, ' '};
  };
  std::string_view tok;
  Fields dims = header_line("InvalidDimensions");
  if (!dims.next(tok)) throw Error("InvalidDimensions");
  const size_t width = ppmUsize(tok);
  if (!dims.next(tok)) throw Error("InvalidDimensions");
  const size_t height = ppmUsize(tok);
  if (dims.next(tok)) throw Error("InvalidDimensions");
  Fields scale_fields = header_line("InvalidScale");
  if (!scale_fields.next(tok)) throw Error("InvalidScale");
  const double scale = static_cast<double>(ppmUsize(tok));
  if (scale_fields.next(tok)) throw Error("InvalidScale");

  Canvas c;
  c.width = width;
  c.height = height;
  Color current{0.0, 0.0, 0.0};
  unsigned filled = 0;
  while (lines.next(line)) {
    Fields samples{line, ' '};
    bool first = true;
    while (samples.next(tok)) {
      if (first && tok[0] == '#') break;  // a comment among the pixel data
      first = false;
      const double val = ppmFloat(tok) / scale;
      if (filled == 0) {
        current.r = val;
      } else if (filled == 1) {
        current.g = val;
      } else {
        current.b = val;
        c.pixels.push_back(current);
      }
      filled = (filled + 1) % 3;
    }
  }
  if (c.pixels.size() != width * height) throw Error("InvalidDimensions");
  return c;
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
