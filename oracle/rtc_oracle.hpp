// rtc_oracle.hpp — CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A plain C++ restatement (not an optimisation) of the reference's per-pixel
// path, used only by tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg to check and time-compare the HIP path.  Nothing under
// ray-tracer-challenge_amd/ may include, link or call anything in oracle/.
//
// Parity pinning: the reference (Zig) cannot be built in this environment (no
// Zig toolchain), so this restatement is pinned by the reference's
// own unit-test known answers — every numeric KAT of SURVEY §4 is restated in
// oracle/kat_main.cpp and must pass (tests/test_oracle_kats.py).
//
// Follows, function by function and in evaluation order (compile with
// -ffp-contract=off; Zig's default float mode is strict):
//   src/raytracer/tuple.zig, matrix.zig, ray.zig, color.zig, light.zig
//   src/raytracer/shapes/shape.zig, sphere.zig, plane.zig, cube.zig,
//       cylinder.zig, cone.zig, triangle.zig, bounding_box.zig, group.zig
//   src/raytracer/patterns/pattern.zig, solid.zig, stripes.zig, checkers.zig,
//       rings.zig, gradient.zig, blend.zig
//   src/raytracer/material.zig, world.zig, camera.zig
// All types are f64 (src/main.zig:71).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <limits>
#include <new>
#include <memory>
#include <stdexcept>
#include <vector>

namespace orc {

constexpr double INF = std::numeric_limits<double>::infinity();

// Per-thread work counters (ray = one World.intersect call).
struct Counters {
  uint64_t primary = 0, secondary = 0, shadow = 0;
  uint64_t bbox_tests = 0, tri_tests = 0, smooth_hits = 0, xforms = 0, leaf_tests = 0;
  void add(const Counters& o) {
    primary += o.primary; secondary += o.secondary; shadow += o.shadow;
    bbox_tests += o.bbox_tests; tri_tests += o.tri_tests; smooth_hits += o.smooth_hits;
    xforms += o.xforms; leaf_tests += o.leaf_tests;
  }
};
inline Counters& counters() {
  static thread_local Counters c;
  return c;
}

// ------------------------------------------------------------------ allocation
// The reference gives every row job one ArenaAllocator and resets it after every pixel with
// `.retain_capacity` (camera.zig:113-120): all the intersection lists of a pixel's ray tree are bump allocations
// that are never freed one by one, and from the second pixel on the arena asks the system for nothing.  Same here:
// a per-thread bump arena behind a stateless std allocator (deallocate is a no-op), reset by the render loop after
// every pixel.  (With std::allocator the oracle spent its time in glibc malloc: 256 threads ran 18 times slower per
// thread than one.)  Users that never reset (the KAT binary) just keep bumping.
class Arena {
 public:
  ~Arena() {
    for (Block& b : blocks_) std::free(b.p);
  }
  void* allocate(size_t bytes, size_t align) {
    for (;;) {
      if (!blocks_.empty()) {
        Block& b = blocks_.back();
        const size_t at = (b.used + align - 1) & ~(align - 1);
        if (at + bytes <= b.size) {
          b.used = at + bytes;
          return b.p + at;
        }
      }
      const size_t size = std::max<size_t>(std::max<size_t>(bytes + align, 1u << 16), 2 * total_);
      char* p = static_cast<char*>(std::malloc(size));
      if (!p) throw std::bad_alloc();
      blocks_.push_back({p, size, 0});
      total_ += size;
    }
  }
  void reset() {  // ArenaAllocator.reset(.retain_capacity): one block of the whole capacity, nothing returned
    if (blocks_.size() > 1) {
      for (Block& b : blocks_) std::free(b.p);
      blocks_.clear();
      char* p = static_cast<char*>(std::malloc(total_));
      if (!p) throw std::bad_alloc();
      blocks_.push_back({p, total_, 0});
    } else if (!blocks_.empty()) {
      blocks_.back().used = 0;
    }
  }
  static Arena& mine() {
    static thread_local Arena a;
    return a;
  }

 private:
  struct Block {
    char* p;
    size_t size, used;
  };
  std::vector<Block> blocks_;
  size_t total_ = 0;
};
template <class T>
struct ArenaAlloc {
  using value_type = T;
  ArenaAlloc() = default;
  template <class U>
  ArenaAlloc(const ArenaAlloc<U>&) {}
  T* allocate(size_t n) { return static_cast<T*>(Arena::mine().allocate(n * sizeof(T), alignof(T) < 8 ? 8 : alignof(T))); }
  void deallocate(T*, size_t) {}
  template <class U>
  bool operator==(const ArenaAlloc<U>&) const { return true; }
  template <class U>
  bool operator!=(const ArenaAlloc<U>&) const { return false; }
};

// ------------------------------------------------------------------ tuple.zig
struct Tuple {
  double x, y, z, w;
};
inline Tuple point(double x, double y, double z) { return {x, y, z, 1.0}; }
inline Tuple vec3(double x, double y, double z) { return {x, y, z, 0.0}; }
inline Tuple add(Tuple a, Tuple b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }        // :54
inline Tuple sub(Tuple a, Tuple b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }        // :62
inline Tuple negate(Tuple a) { return {-a.x, -a.y, -a.z, -a.w}; }                                  // :72
inline Tuple mul(Tuple a, double v) { return {a.x * v, a.y * v, a.z * v, a.w * v}; }               // :82
inline Tuple div(Tuple a, double v) { return {a.x / v, a.y / v, a.z / v, a.w / v}; }               // :92
inline double magnitude(Tuple a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w); }  // :102
inline Tuple normalized(Tuple a) {                                                                 // :109
  const double mag = magnitude(a);
  if (mag == 0.0) return a;
  return div(a, mag);
}
inline double dot(Tuple a, Tuple b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }      // :121
inline Tuple cross(Tuple a, Tuple b) {                                                             // :128
  return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline Tuple reflect(Tuple a, Tuple n) { return sub(a, mul(n, 2.0 * dot(a, n))); }                 // :139
inline bool approxEqual(Tuple a, Tuple b, double tol = 1e-5) {                                     // :42
  return std::fabs(a.x - b.x) < tol && std::fabs(a.y - b.y) < tol && std::fabs(a.z - b.z) < tol &&
         std::fabs(a.w - b.w) < tol;
}

// ------------------------------------------------------------------ color.zig
struct Color {
  double r, g, b;
};
inline Color cadd(Color a, Color b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
inline Color cmul(Color a, double v) { return {a.r * v, a.g * v, a.b * v}; }
inline Color cemul(Color a, Color b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }
inline bool approxEqual(Color a, Color b, double tol = 1e-5) {
  return std::fabs(a.r - b.r) < tol && std::fabs(a.g - b.g) < tol && std::fabs(a.b - b.b) < tol;
}

// ------------------------------------------------------------------ matrix.zig (N = 4 and its minors)
inline double det2(const double m[2][2]) { return m[0][0] * m[1][1] - m[0][1] * m[1][0]; }  // :192
inline double det3(const double m[3][3]) {                                                  // :194-196 with N=3
  double det = 0.0;
  for (int col = 0; col < 3; ++col) {
    double s[2][2];
    for (int r = 1; r < 3; ++r)
      for (int c = 0, k = 0; c < 3; ++c)
        if (c != col) s[r - 1][k++] = m[r][c];
    const double minor = det2(s);
    det += m[0][col] * ((col % 2 == 0) ? minor : -minor);
  }
  return det;
}
struct Matrix {
  double d[4][4];

  static Matrix identity() {
    Matrix m{};
    for (int i = 0; i < 4; ++i) m.d[i][i] = 1.0;
    return m;
  }
  double cofactor(int row, int col) const {  // :157-185
    double s[3][3];
    for (int r = 0, rr = 0; r < 4; ++r) {
      if (r == row) continue;
      for (int c = 0, cc = 0; c < 4; ++c) {
        if (c == col) continue;
        s[rr][cc++] = d[r][c];
      }
      ++rr;
    }
    const double minor = det3(s);
    return ((row + col) % 2 == 0) ? minor : -minor;
  }
  double det() const {  // :188-199
    double det_ = 0.0;
    for (int col = 0; col < 4; ++col) det_ += d[0][col] * cofactor(0, col);
    return det_;
  }
  Matrix mul(const Matrix& o) const {  // :106-120
    Matrix r;
    for (int row = 0; row < 4; ++row)
      for (int col = 0; col < 4; ++col) {
        double sum = 0;
        for (int i = 0; i < 4; ++i) sum += d[row][i] * o.d[i][col];
        r.d[row][col] = sum;
      }
    return r;
  }
  Tuple tupleMul(Tuple t) const {  // :124-140
    Tuple r;
    r.x = dot(Tuple{d[0][0], d[0][1], d[0][2], d[0][3]}, t);
    r.y = dot(Tuple{d[1][0], d[1][1], d[1][2], d[1][3]}, t);
    r.z = dot(Tuple{d[2][0], d[2][1], d[2][2], d[2][3]}, t);
    r.w = dot(Tuple{d[3][0], d[3][1], d[3][2], d[3][3]}, t);
    return r;
  }
  Matrix transpose() const {
    Matrix t;
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) t.d[r][c] = d[c][r];
    return t;
  }
  Matrix inverse() const {  // :202-217
    const double det_ = det();
    if (std::fabs(det_) < 1e-5) throw std::runtime_error("NotInvertible");
    Matrix inv;
    for (int row = 0; row < 4; ++row)
      for (int col = 0; col < 4; ++col) inv.d[col][row] = cofactor(row, col) / det_;
    return inv;
  }
  static Matrix make(std::initializer_list<double> v) {
    Matrix m;
    int i = 0;
    for (double x : v) {
      m.d[i / 4][i % 4] = x;
      ++i;
    }
    return m;
  }
  Matrix translate(double x, double y, double z) const {  // :222
    return make({1, 0, 0, x, 0, 1, 0, y, 0, 0, 1, z, 0, 0, 0, 1}).mul(*this);
  }
  Matrix scale(double x, double y, double z) const {  // :236
    return make({x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1}).mul(*this);
  }
  Matrix rotateX(double a) const {  // :249
    return make({1, 0, 0, 0, 0, std::cos(a), -std::sin(a), 0, 0, std::sin(a), std::cos(a), 0, 0, 0, 0, 1}).mul(*this);
  }
  Matrix rotateY(double a) const {  // :262
    return make({std::cos(a), 0, std::sin(a), 0, 0, 1, 0, 0, -std::sin(a), 0, std::cos(a), 0, 0, 0, 0, 1}).mul(*this);
  }
  Matrix rotateZ(double a) const {  // :275
    return make({std::cos(a), -std::sin(a), 0, 0, std::sin(a), std::cos(a), 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}).mul(*this);
  }
  Matrix shear(double xy, double xz, double yx, double yz, double zx, double zy) const {  // :311
    return make({1, xy, xz, 0, yx, 1, yz, 0, zx, zy, 1, 0, 0, 0, 0, 1}).mul(*this);
  }
  static Matrix viewTransform(Tuple from, Tuple to, Tuple up) {  // :54-67
    const Tuple forward = normalized(sub(to, from));
    const Tuple left = cross(forward, normalized(up));
    const Tuple true_up = cross(left, forward);
    const Matrix orientation = make({left.x, left.y, left.z, 0, true_up.x, true_up.y, true_up.z, 0,
                                     -forward.x, -forward.y, -forward.z, 0, 0, 0, 0, 1});
    return orientation.mul(identity().translate(-from.x, -from.y, -from.z));
  }
  bool approxEqual(const Matrix& o, double tol = 1e-5) const {
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j)
        if (std::fabs(d[i][j] - o.d[i][j]) > tol) return false;
    return true;
  }
};

// ------------------------------------------------------------------ ray.zig
struct Ray {
  Tuple origin, direction;
  Tuple position(double t) const { return add(origin, mul(direction, t)); }                           // :25
  Ray transform(const Matrix& m) const { return {m.tupleMul(origin), m.tupleMul(direction)}; }        // :30
};

// ------------------------------------------------------------------ patterns/*.zig
// Zig @mod(x, y) for floats as the LLVM backend lowers it: r = fmod(x,y); if x < 0 then
// fmod(r + y, y) else r  (result has the sign of the divisor; stripes.zig:50 expects
// @mod(-0.1, 2) = 1.9).
inline double zigMod(double x, double y) {
  const double a = std::fmod(x, y);
  if (x < 0.0) return std::fmod(a + y, y);
  return a;
}

enum PatternKind : uint8_t {  // == RTC_PAT_* (include/rtc.h)
  PAT_SOLID = 0, PAT_STRIPES = 1, PAT_RINGS = 2, PAT_GRADIENT = 3, PAT_RADIAL_GRADIENT = 4,
  PAT_CHECKERS = 5, PAT_BLEND = 6, PAT_PERTURB = 7, PAT_TEXTURE_MAP = 8, PAT_TEST = 9
};

// ------------------------------------------------------------------ noise.zig (Perlin's improved noise)
// noise.zig:6-23: Ken Perlin's reference permutation (public constant of the algorithm), doubled (noise.zig:25-33)
inline const uint8_t* perlinPermutation() {
  static const uint8_t permutation[256] = {
      151, 160, 137, 91,  90,  15,  131, 13,  201, 95,  96,  53,  194, 233, 7,   225, 140, 36,  103, 30,  69,  142,
      8,   99,  37,  240, 21,  10,  23,  190, 6,   148, 247, 120, 234, 75,  0,   26,  197, 62,  94,  252, 219, 203,
      117, 35,  11,  32,  57,  177, 33,  88,  237, 149, 56,  87,  174, 20,  125, 136, 171, 168, 68,  175, 74,  165,
      71,  134, 139, 48,  27,  166, 77,  146, 158, 231, 83,  111, 229, 122, 60,  211, 133, 230, 220, 105, 92,  41,
      55,  46,  245, 40,  244, 102, 143, 54,  65,  25,  63,  161, 1,   216, 80,  73,  209, 76,  132, 187, 208, 89,
      18,  169, 200, 196, 135, 130, 116, 188, 159, 86,  164, 100, 109, 198, 173, 186, 3,   64,  52,  217, 226, 250,
      124, 123, 5,   202, 38,  147, 118, 126, 255, 82,  85,  212, 207, 206, 59,  227, 47,  16,  58,  17,  182, 189,
      28,  42,  223, 183, 170, 213, 119, 248, 152, 2,   44,  154, 163, 70,  221, 153, 101, 155, 167, 43,  172, 9,
      129, 22,  39,  253, 19,  98,  108, 110, 79,  113, 224, 232, 178, 185, 112, 104, 218, 246, 97,  228, 251, 34,
      242, 193, 238, 210, 144, 12,  191, 179, 162, 241, 81,  51,  145, 235, 249, 14,  239, 107, 49,  192, 214, 31,
      181, 199, 106, 157, 184, 84,  204, 176, 115, 121, 50,  45,  127, 4,   150, 254, 138, 236, 205, 93,  222, 114,
      67,  29,  24,  72,  243, 141, 128, 195, 78,  66,  215, 61,  156, 180};
  return permutation;
}
// noise.zig:51-97.  The reference indexes a doubled table with u8 sums (overflow is undefined there);
// p[k] == p[k & 255] for every k < 512, so every reading of those sums gives the value computed here.
inline double perlinNoise(double x, double y, double z) {
  const uint8_t* p = perlinPermutation();
  auto P = [&](int i) { return static_cast<int>(p[i & 255]); };
  const int X = static_cast<int>(static_cast<long long>(std::floor(x)) & 255);
  const int Y = static_cast<int>(static_cast<long long>(std::floor(y)) & 255);
  const int Z = static_cast<int>(static_cast<long long>(std::floor(z)) & 255);
  x -= std::floor(x);
  y -= std::floor(y);
  z -= std::floor(z);
  auto fade = [](double t) { return t * t * t * (t * (t * 6.0 - 15.0) + 10.0); };
  auto lerp = [](double t, double a, double b) { return a + t * (b - a); };
  auto grad = [](int hash, double gx, double gy, double gz) {
    const int h = hash & 15;
    const double u = h < 8 ? gx : gy;
    const double v = h < 4 ? gy : ((h == 12 || h == 14) ? gx : gz);
    return ((h & 1) == 0 ? u : -u) + ((h & 2) == 0 ? v : -v);
  };
  const double u = fade(x), v = fade(y), w = fade(z);
  const int A = P(X) + Y, AA = P(A) + Z, AB = P(A + 1) + Z;
  const int B = P(X + 1) + Y, BA = P(B) + Z, BB = P(B + 1) + Z;
  return lerp(w,
              lerp(v, lerp(u, grad(P(AA), x, y, z), grad(P(BA), x - 1, y, z)),
                   lerp(u, grad(P(AB), x, y - 1, z), grad(P(BB), x - 1, y - 1, z))),
              lerp(v, lerp(u, grad(P(AA + 1), x, y, z - 1), grad(P(BA + 1), x - 1, y, z - 1)),
                   lerp(u, grad(P(AB + 1), x, y - 1, z - 1), grad(P(BB + 1), x - 1, y - 1, z - 1))));
}
inline double octaveNoise(double x, double y, double z, unsigned octaves, double persistence) {  // noise.zig:35-49
  double total = 0.0, frequency = 1.0, amplitude = 1.0, max_value = 0.0;
  for (unsigned i = 0; i < octaves; ++i) {
    total += perlinNoise(x * frequency, y * frequency, z * frequency) * amplitude;
    max_value += amplitude;
    amplitude *= persistence;
    frequency *= 2.0;
  }
  return total / max_value;
}

// ------------------------------------------------------------------ texture_map.zig
struct Pattern;
struct UvImage {  // Canvas(T) behind a UvImage (texture_map.zig:66-105)
  size_t width = 0, height = 0;
  std::vector<double> rgb;  // [height][width][3]
  Color at(size_t x, size_t y) const {  // Canvas.getPixelPointer(x, y).?.*; out of bounds is a panic there
    if (x >= width) x = width - 1;
    if (y >= height) y = height - 1;
    const double* p = &rgb[3 * (y * width + x)];
    return {p[0], p[1], p[2]};
  }
};
enum UvKind : uint8_t { UV_ALIGN_CHECK = 0, UV_CHECKERS = 1, UV_IMAGE = 2, UV_TEST = 3 };     // == RTC_UV_*
enum TexMapping : uint8_t { TEX_SPHERICAL = 0, TEX_PLANAR = 1, TEX_CYLINDRICAL = 2, TEX_CUBIC = 3 };  // == RTC_TEX_*
struct UvPattern {
  UvKind kind = UV_TEST;
  double width = 0.0, height = 0.0;
  const Pattern* sub[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  const UvImage* image = nullptr;
  bool bilinear = false;
  Color uvPatternAt(double u, double v, Tuple object_point) const;
};
struct TextureMap {
  TexMapping mapping = TEX_SPHERICAL;
  UvPattern faces[6];  // Cubic.Face order: front, back, left, right, up, down (texture_map.zig:216)
  Color patternAt(Tuple pattern_point, Tuple object_point) const;
};

struct Pattern {
  Matrix transform = Matrix::identity();
  Matrix inverse = Matrix::identity();
  PatternKind kind = PAT_SOLID;
  Color rgb{1, 1, 1};
  const Pattern* a = nullptr;
  const Pattern* b = nullptr;
  const TextureMap* texture_map = nullptr;

  void setTransform(const Matrix& m) {  // pattern.zig:103
    transform = m;
    inverse = m.inverse();
  }
  // pattern.zig:112-124 — sub-patterns are evaluated at the OBJECT point with their own inverse.
  Color patternAt(Tuple object_point) const {
    const Tuple pp = inverse.tupleMul(object_point);
    switch (kind) {
      case PAT_SOLID: return rgb;                                   // solid.zig:20
      case PAT_TEST: return {pp.x, pp.y, pp.z};                     // pattern.zig:144
      case PAT_STRIPES:                                             // stripes.zig:27
        return (zigMod(pp.x, 2.0) < 1.0) ? a->patternAt(object_point) : b->patternAt(object_point);
      case PAT_CHECKERS:                                            // checkers.zig:23
        return (zigMod(std::floor(pp.x) + std::floor(pp.y) + std::floor(pp.z), 2.0) < 1.0)
                   ? a->patternAt(object_point)
                   : b->patternAt(object_point);
      case PAT_RINGS:                                               // rings.zig
        return (zigMod(std::floor(std::sqrt(pp.x * pp.x + pp.z * pp.z)), 2.0) < 1.0)
                   ? a->patternAt(object_point)
                   : b->patternAt(object_point);
      case PAT_GRADIENT: {                                          // gradient.zig
        const Color ca = a->patternAt(object_point), cb = b->patternAt(object_point);
        const Color distance{cb.r - ca.r, cb.g - ca.g, cb.b - ca.b};
        const double fraction = pp.x - std::floor(pp.x);
        return cadd(ca, cmul(distance, fraction));
      }
      case PAT_RADIAL_GRADIENT: {
        const Color ca = a->patternAt(object_point), cb = b->patternAt(object_point);
        const Color distance{cb.r - ca.r, cb.g - ca.g, cb.b - ca.b};
        const double mag = std::sqrt(pp.x * pp.x + pp.z * pp.z);
        const double fraction = mag - std::floor(mag);
        return cadd(ca, cmul(distance, fraction));
      }
      case PAT_BLEND: {                                             // blend.zig
        const Color ca = a->patternAt(object_point), cb = b->patternAt(object_point);
        return cmul(cadd(ca, cb), 0.5);
      }
      case PAT_TEXTURE_MAP: return texture_map->patternAt(pp, object_point);  // texture_map.zig:322-330
      case PAT_PERTURB: {                                           // perturb.zig:31-46; rgb = PerturbInfo
        const unsigned octaves = static_cast<unsigned>(rgb.g);
        const Tuple offset = vec3(octaveNoise(object_point.x, object_point.y, object_point.z, octaves, rgb.b),
                                  octaveNoise(object_point.x, object_point.y, object_point.z + 1.0, octaves, rgb.b),
                                  octaveNoise(object_point.x, object_point.y, object_point.z + 2.0, octaves, rgb.b));
        return a->patternAt(add(object_point, mul(offset, rgb.r)));
      }
      default: throw std::runtime_error("oracle: unsupported pattern kind");
    }
  }
};

inline Color UvPattern::uvPatternAt(double u, double v, Tuple object_point) const {
  switch (kind) {
    case UV_TEST: return {u, v, 0.0};  // texture_map.zig:13-17
    case UV_ALIGN_CHECK:               // texture_map.zig:30-39
      if (v > 0.8) {
        if (u < 0.2) return sub[1]->patternAt(object_point);
        if (u > 0.8) return sub[2]->patternAt(object_point);
      } else if (v < 0.2) {
        if (u < 0.2) return sub[3]->patternAt(object_point);
        if (u > 0.8) return sub[4]->patternAt(object_point);
      }
      return sub[0]->patternAt(object_point);
    case UV_CHECKERS: {  // texture_map.zig:52-60
      const double u_adj = std::floor(u * width), v_adj = std::floor(v * height);
      return (zigMod(u_adj + v_adj, 2.0) < 1.0) ? sub[0]->patternAt(object_point) : sub[1]->patternAt(object_point);
    }
    case UV_IMAGE: {  // texture_map.zig:74-103
      const double v_flip = 1.0 - v;
      const double x = u * static_cast<double>(image->width - 1);
      const double y = v_flip * static_cast<double>(image->height - 1);
      auto idx = [](double f) { return f <= 0.0 ? size_t{0} : static_cast<size_t>(f); };  // @intFromFloat; negative is UB there
      if (!bilinear) return image->at(idx(std::round(x)), idx(std::round(y)));  // @round: half away from zero
      const double x1 = std::floor(x), x2 = std::ceil(x), y1 = std::floor(y), y2 = std::ceil(y);
      const Color c11 = image->at(idx(x1), idx(y1)), c21 = image->at(idx(x2), idx(y1));
      const Color c12 = image->at(idx(x1), idx(y2)), c22 = image->at(idx(x2), idx(y2));
      // on an integer coordinate x1 == x2 and both weights are 0: the reference returns black there
      const Color cx1 = cadd(cmul(c11, x2 - x), cmul(c21, x - x1));
      const Color cx2 = cadd(cmul(c12, x2 - x), cmul(c22, x - x1));
      return cadd(cmul(cx1, y2 - y), cmul(cx2, y - y1));
    }
  }
  return {0.0, 0.0, 0.0};
}

inline Color TextureMap::patternAt(Tuple p, Tuple object_point) const {
  const double kPi = 3.14159265358979323846264338327950288;  // std.math.pi
  switch (mapping) {
    case TEX_SPHERICAL: {  // texture_map.zig:180-195
      const double theta = std::atan2(p.x, p.z);
      const double radius = magnitude(vec3(p.x, p.y, p.z));
      const double phi = std::acos(p.y / radius);
      const double raw_u = theta / (2.0 * kPi);
      const double u = 1.0 - (raw_u + 0.5);
      const double v = 1.0 - phi / kPi;
      return faces[0].uvPatternAt(u, v, object_point);
    }
    case TEX_PLANAR:  // texture_map.zig:201-205
      return faces[0].uvPatternAt(zigMod(p.x, 1.0), zigMod(p.z, 1.0), object_point);
    case TEX_CYLINDRICAL: {  // texture_map.zig:210-217
      const double theta = std::atan2(p.x, p.z);
      const double raw_u = theta / (2.0 * kPi);
      const double u = 1.0 - (raw_u + 0.5);
      return faces[0].uvPatternAt(u, zigMod(p.y, 1.0), object_point);
    }
    case TEX_CUBIC: {  // texture_map.zig:219-303
      const double coord = std::fmax(std::fabs(p.x), std::fmax(std::fabs(p.y), std::fabs(p.z)));
      enum { FRONT = 0, BACK = 1, LEFT = 2, RIGHT = 3, UP = 4, DOWN = 5 };
      int face = BACK;
      if (coord == p.x) face = RIGHT;
      else if (coord == -p.x) face = LEFT;
      else if (coord == p.y) face = UP;
      else if (coord == -p.y) face = DOWN;
      else if (coord == p.z) face = FRONT;
      double u = 0.0, v = 0.0;
      switch (face) {
        case FRONT: u = zigMod(p.x + 1.0, 2.0) / 2.0; v = zigMod(p.y + 1.0, 2.0) / 2.0; break;
        case BACK: u = zigMod(1.0 - p.x, 2.0) / 2.0; v = zigMod(p.y + 1.0, 2.0) / 2.0; break;
        case LEFT: u = zigMod(p.z + 1.0, 2.0) / 2.0; v = zigMod(p.y + 1.0, 2.0) / 2.0; break;
        case RIGHT: u = zigMod(1.0 - p.z, 2.0) / 2.0; v = zigMod(p.y + 1.0, 2.0) / 2.0; break;
        case UP: u = zigMod(p.x + 1.0, 2.0) / 2.0; v = zigMod(1.0 - p.z, 2.0) / 2.0; break;
        default: u = zigMod(p.x + 1.0, 2.0) / 2.0; v = zigMod(p.z + 1.0, 2.0) / 2.0; break;
      }
      return faces[face].uvPatternAt(u, v, object_point);
    }
  }
  return {0.0, 0.0, 0.0};
}

struct Light {  // light.zig
  Tuple position;
  Color intensity;
};

struct Shape;

struct Material {  // material.zig:18-25
  Pattern pattern;
  double ambient = 0.1, diffuse = 0.9, specular = 0.9, shininess = 200.0;
  double reflective = 0.0, transparency = 0.0, refractive_index = 1.0;
  Color lighting(const Light& light, const Shape* object, Tuple point, Tuple point_to_eye, Tuple normal,
                 bool in_shadow) const;
};

// ------------------------------------------------------------------ shapes
enum ShapeKind : uint8_t {
  SPHERE = 0, PLANE = 1, CUBE = 2, CYLINDER = 3, TRIANGLE = 4, SMOOTH_TRIANGLE = 5, CONE = 6,  // == RTC_*
  GROUP = 100, BOUNDING_BOX = 101, TEST_SHAPE = 102, CSG = 103
};
enum CsgOp : uint8_t { CSG_UNION = 1, CSG_INTERSECTION = 2, CSG_DIFFERENCE = 3 };  // csg.zig:16-20, == RTC_CSG_*

struct Intersection {  // shape.zig:23-47
  double t;
  const Shape* object;
  double u = 0.0, v = 0.0;
};
using Intersections = std::vector<Intersection, ArenaAlloc<Intersection>>;

// shape.zig:64-66 — std.mem.sort is a stable, in-place sort that allocates nothing (SURVEY assumption A1).  Any stable
// sort yields the same order; short lists (all but pathological ones) take the in-place insertion sort, because
// std::stable_sort asks the heap for a merge buffer on every call.
inline void sortIntersections(Intersections& xs) {
  if (xs.size() <= 64) {
    for (size_t i = 1; i < xs.size(); ++i) {
      const Intersection v = xs[i];
      size_t j = i;
      while (j > 0 && v.t < xs[j - 1].t) {
        xs[j] = xs[j - 1];
        --j;
      }
      xs[j] = v;
    }
    return;
  }
  std::stable_sort(xs.begin(), xs.end(), [](const Intersection& a, const Intersection& b) { return a.t < b.t; });
}
// shape.zig:71-80
inline long hit(const Intersections& xs, size_t from = 0) {
  for (size_t i = from; i < xs.size(); ++i)
    if (xs[i].t >= 0.0) return static_cast<long>(i);
  return -1;
}

inline size_t nextId() {
  static size_t id = 0;
  return id++;
}

struct Shape {
  size_t id = 0;
  Matrix transform = Matrix::identity();
  Matrix inverse = Matrix::identity();
  Matrix inverse_transpose = Matrix::identity();
  Material material;
  ShapeKind kind = SPHERE;
  bool casts_shadow = true;
  // cylinder / cone
  double ymin = -INF, ymax = INF;
  bool closed = false;
  // triangles
  Tuple p1{}, e1{}, e2{}, normal{}, n1{}, n2{}, n3{};
  Tuple p2{}, p3{};  // the other two corners: only Triangle.bounds() reads them (triangle.zig:72-79; rtc_oracle_scene.hpp)
  // bounding box / group
  Tuple bmin = point(INF, INF, INF), bmax = point(-INF, -INF, -INF);
  std::vector<Shape> children;  // a csg: {left, right} (csg.zig:28-29)
  CsgOp csg_op = CSG_UNION;

  static Shape make(ShapeKind k) {
    Shape s;
    s.id = nextId();
    s.kind = k;
    return s;
  }
  static Shape triangle(Tuple p1, Tuple p2, Tuple p3) {  // shape.zig:186-204
    Shape s = make(TRIANGLE);
    s.p1 = p1;
    s.p2 = p2;
    s.p3 = p3;
    s.e1 = sub(p2, p1);
    s.e2 = sub(p3, p1);
    s.normal = normalized(cross(s.e2, s.e1));
    return s;
  }
  static Shape smoothTriangle(Tuple p1, Tuple p2, Tuple p3, Tuple n1, Tuple n2, Tuple n3) {  // shape.zig:207-227
    Shape s = make(SMOOTH_TRIANGLE);
    s.p1 = p1;
    s.p2 = p2;
    s.p3 = p3;
    s.e1 = sub(p2, p1);
    s.e2 = sub(p3, p1);
    s.n1 = n1; s.n2 = n2; s.n3 = n3;
    return s;
  }
  static Shape glassSphere() {
    Shape s = make(SPHERE);
    s.material.transparency = 1.0;
    s.material.refractive_index = 1.5;
    return s;
  }

  void setTransform(const Matrix& m) {  // shape.zig:303-308 (leaf branch; the oracle gets groups pre-built)
    transform = m;
    inverse = m.inverse();
    inverse_transpose = inverse.transpose();
  }
  Tuple worldToObject(Tuple p) const { return inverse.tupleMul(p); }  // shape.zig:133
  Tuple normalToWorld(Tuple normal_) const {                          // shape.zig:139
    Tuple n = inverse_transpose.tupleMul(normal_);
    n.w = 0.0;
    n = normalized(n);
    return n;
  }

  Intersections intersect(const Ray& ray) const;           // shape.zig:313-335
  Intersections localIntersect(const Ray& ray) const;      // per-kind
  Tuple localNormalAt(Tuple p, const Intersection& h) const;
  Tuple normalAt(Tuple p, const Intersection& h) const {   // shape.zig:338-350
    const Tuple local_point = worldToObject(p);
    const Tuple local_normal = localNormalAt(local_point, h);
    return normalToWorld(local_normal);
  }
};

// cube.zig:24-47 / bounding_box.zig:112-137 (same body with min/max as parameters)
inline void checkAxis(double origin, double direction, double mn, double mx, double& tmin, double& tmax) {
  const double epsilon = 1e-5;
  const double tmin_numerator = mn - origin;
  const double tmax_numerator = mx - origin;
  if (std::fabs(direction) >= epsilon) {
    tmin = tmin_numerator / direction;
    tmax = tmax_numerator / direction;
  } else {
    tmin = tmin_numerator * INF;
    tmax = tmax_numerator * INF;
  }
  if (tmin > tmax) {
    const double save = tmax;
    tmax = tmin;
    tmin = save;
  }
}

// cube.zig:49-79 and bounding_box.zig:139-165 share this body (min/max differ).
inline void slabIntersect(Tuple mn, Tuple mx, const Ray& ray, const Shape* self, Intersections& xs) {
  double xtmin, xtmax, ytmin, ytmax, ztmin, ztmax;
  checkAxis(ray.origin.x, ray.direction.x, mn.x, mx.x, xtmin, xtmax);
  checkAxis(ray.origin.y, ray.direction.y, mn.y, mx.y, ytmin, ytmax);
  checkAxis(ray.origin.z, ray.direction.z, mn.z, mx.z, ztmin, ztmax);
  // Zig @max/@min return the non-NaN operand == fmax/fmin.
  const double tmin = std::fmax(xtmin, std::fmax(ytmin, ztmin));
  const double tmax = std::fmin(xtmax, std::fmin(ytmax, ztmax));
  if (tmin > tmax) return;
  xs.push_back({tmin, self});
  xs.push_back({tmax, self});
}

inline bool cylCheckCap(const Ray& ray, double t) {  // cylinder.zig:30-35
  const double x = ray.origin.x + t * ray.direction.x;
  const double z = ray.origin.z + t * ray.direction.z;
  return x * x + z * z <= 1.0;
}
inline bool coneCheckCap(const Ray& ray, double t, double radius) {  // cone.zig:29-34
  const double x = ray.origin.x + t * ray.direction.x;
  const double z = ray.origin.z + t * ray.direction.z;
  return x * x + z * z <= radius * radius;
}

// csg.zig:112-118
inline bool intersectionAllowed(CsgOp op, bool lhit, bool inl, bool inr) {
  switch (op) {
    case CSG_UNION: return (lhit && !inr) || !(lhit || inl);
    case CSG_INTERSECTION: return (lhit && inr) || (!lhit && inl);
    case CSG_DIFFERENCE: return (lhit && !inr) || (!lhit && inl);
  }
  return false;
}
// csg.zig:120-139
inline bool csgIncludes(const Shape& a, const Shape& b) {
  if (a.kind == GROUP) {
    for (const Shape& child : a.children)
      if (csgIncludes(child, b)) return true;
    return false;
  }
  if (a.kind == CSG) return csgIncludes(a.children[0], b) || csgIncludes(a.children[1], b);
  return a.id == b.id;
}
// csg.zig:51-72
inline Intersections csgFilter(const Shape& csg, const Intersections& xs) {
  bool inl = false, inr = false;
  Intersections result;
  for (const Intersection& i : xs) {
    const bool lhit = csgIncludes(csg.children[0], *i.object);
    if (intersectionAllowed(csg.csg_op, lhit, inl, inr)) result.push_back(i);
    if (lhit) {
      inl = !inl;
    } else {
      inr = !inr;
    }
  }
  return result;
}

inline Intersections Shape::localIntersect(const Ray& ray) const {
  Intersections xs;
  switch (kind) {
    case SPHERE: {  // sphere.zig:24-46
      const Tuple sphere_to_ray = sub(ray.origin, point(0.0, 0.0, 0.0));
      const double a = dot(ray.direction, ray.direction);
      const double b = 2.0 * dot(sphere_to_ray, ray.direction);
      const double c = dot(sphere_to_ray, sphere_to_ray) - 1.0;
      const double discriminant = b * b - 4.0 * a * c;
      if (discriminant >= 0.0) {
        const double t1 = (-b - std::sqrt(discriminant)) / (2.0 * a);
        const double t2 = (-b + std::sqrt(discriminant)) / (2.0 * a);
        xs.push_back({t1, this});
        xs.push_back({t2, this});
        sortIntersections(xs);
      }
      return xs;
    }
    case PLANE: {  // plane.zig:25-36
      if (std::fabs(ray.direction.y) > 1e-5) xs.push_back({-ray.origin.y / ray.direction.y, this});
      return xs;
    }
    case CUBE:  // cube.zig:49-79
      slabIntersect(point(-1.0, -1.0, -1.0), point(1.0, 1.0, 1.0), ray, this, xs);
      return xs;
    case BOUNDING_BOX:  // bounding_box.zig:139-165
      counters().bbox_tests++;
      slabIntersect(bmin, bmax, ray, this, xs);
      return xs;
    case CYLINDER: {  // cylinder.zig:53-98
      auto intersectCaps = [&]() {  // cylinder.zig:37-51
        if (!closed || std::fabs(ray.direction.y) < 1e-5) return;
        double t = (ymin - ray.origin.y) / ray.direction.y;
        if (cylCheckCap(ray, t)) xs.push_back({t, this});
        t = (ymax - ray.origin.y) / ray.direction.y;
        if (cylCheckCap(ray, t)) xs.push_back({t, this});
      };
      const double a = ray.direction.x * ray.direction.x + ray.direction.z * ray.direction.z;
      if (std::fabs(a) < 1e-5) {
        intersectCaps();
        return xs;
      }
      const double b = 2.0 * ray.origin.x * ray.direction.x + 2.0 * ray.origin.z * ray.direction.z;
      const double c = ray.origin.x * ray.origin.x + ray.origin.z * ray.origin.z - 1.0;
      const double discriminant = b * b - 4.0 * a * c;
      if (discriminant < 0.0) return xs;
      double t0 = (-b - std::sqrt(discriminant)) / (2.0 * a);
      double t1 = (-b + std::sqrt(discriminant)) / (2.0 * a);
      if (t0 > t1) std::swap(t0, t1);
      const double y0 = ray.origin.y + t0 * ray.direction.y;
      if (ymin < y0 && y0 < ymax) xs.push_back({t0, this});
      const double y1 = ray.origin.y + t1 * ray.direction.y;
      if (ymin < y1 && y1 < ymax) xs.push_back({t1, this});
      intersectCaps();
      return xs;
    }
    case CONE: {  // cone.zig:52-113
      const double tol = 1e-4;
      auto intersectCaps = [&]() {  // cone.zig:36-50
        if (!closed || std::fabs(ray.direction.y) < tol) return;
        double t = (ymin - ray.origin.y) / ray.direction.y;
        if (coneCheckCap(ray, t, ymin)) xs.push_back({t, this});
        t = (ymax - ray.origin.y) / ray.direction.y;
        if (coneCheckCap(ray, t, ymax)) xs.push_back({t, this});
      };
      const double a = ray.direction.x * ray.direction.x - ray.direction.y * ray.direction.y +
                       ray.direction.z * ray.direction.z;
      const double b = 2.0 * ray.origin.x * ray.direction.x - 2.0 * ray.origin.y * ray.direction.y +
                       2.0 * ray.origin.z * ray.direction.z;
      if (std::fabs(a) < tol && std::fabs(b) < tol) {
        intersectCaps();
        return xs;
      }
      const double c = ray.origin.x * ray.origin.x - ray.origin.y * ray.origin.y + ray.origin.z * ray.origin.z;
      if (std::fabs(a) < tol) {
        xs.push_back({-c / (2.0 * b), this});
        intersectCaps();
        return xs;
      }
      const double discriminant = b * b - 4.0 * a * c;
      if (discriminant < 0.0) return xs;
      double t0 = (-b - std::sqrt(discriminant)) / (2.0 * a);
      double t1 = (-b + std::sqrt(discriminant)) / (2.0 * a);
      if (t0 > t1) std::swap(t0, t1);
      const double y0 = ray.origin.y + t0 * ray.direction.y;
      if (ymin < y0 && y0 < ymax) xs.push_back({t0, this});
      const double y1 = ray.origin.y + t1 * ray.direction.y;
      if (ymin < y1 && y1 < ymax) xs.push_back({t1, this});
      intersectCaps();
      return xs;
    }
    case TRIANGLE:           // triangle.zig:29-63
    case SMOOTH_TRIANGLE: {  // triangle.zig:225-259
      counters().tri_tests++;
      const Tuple dir_cross_e2 = cross(ray.direction, e2);
      const double det = dot(e1, dir_cross_e2);
      if (std::fabs(det) < 1e-5) return xs;
      const double f = 1.0 / det;
      const Tuple p1_to_origin = sub(ray.origin, p1);
      const double u = f * dot(p1_to_origin, dir_cross_e2);
      if (u < 0.0 || u > 1.0) return xs;
      const Tuple p1_to_origin_cross_e1 = cross(p1_to_origin, e1);
      const double v = f * dot(ray.direction, p1_to_origin_cross_e1);
      if (v < 0.0 || (u + v) > 1.0) return xs;
      const double t = f * dot(e2, p1_to_origin_cross_e1);
      if (kind == SMOOTH_TRIANGLE)
        xs.push_back({t, this, u, v});
      else
        xs.push_back({t, this});
      return xs;
    }
    case GROUP: {  // group.zig:39-62
      // self._bbox.intersect(ray): the bbox is a Shape with identity transform, so the ray
      // goes through ray.transform(identity) first (shape.zig:314-318).
      static const Matrix kIdentity = Matrix::identity();
      counters().bbox_tests++;
      Intersections bbox_xs;
      slabIntersect(bmin, bmax, ray.transform(kIdentity), this, bbox_xs);
      if (bbox_xs.empty()) return xs;
      for (const Shape& child : children) {
        const Intersections cx = child.intersect(ray);
        xs.insert(xs.end(), cx.begin(), cx.end());
      }
      sortIntersections(xs);
      return xs;
    }
    case CSG: {  // csg.zig:74-95
      static const Matrix kIdentity = Matrix::identity();
      counters().bbox_tests++;
      Intersections bbox_xs;
      slabIntersect(bmin, bmax, ray.transform(kIdentity), this, bbox_xs);
      if (bbox_xs.empty()) return xs;
      xs = children[0].intersect(ray);
      const Intersections rightxs = children[1].intersect(ray);
      xs.insert(xs.end(), rightxs.begin(), rightxs.end());
      sortIntersections(xs);
      return csgFilter(*this, xs);
    }
    case TEST_SHAPE: return xs;  // shape.zig:411-420
  }
  return xs;
}

inline Intersections Shape::intersect(const Ray& ray) const {
  if (kind == GROUP || kind == CSG) return localIntersect(ray);
  if (kind != BOUNDING_BOX) {
    counters().leaf_tests++;
    counters().xforms++;
  }
  return localIntersect(ray.transform(inverse));
}

inline double zigSign(double v) { return v > 0 ? 1.0 : (v < 0 ? -1.0 : v); }  // std.math.sign

inline Tuple Shape::localNormalAt(Tuple p, const Intersection& h) const {
  switch (kind) {
    case SPHERE: return sub(p, point(0.0, 0.0, 0.0));  // sphere.zig:48-53
    case PLANE: return vec3(0.0, 1.0, 0.0);            // plane.zig:38-43
    case CUBE: {                                       // cube.zig:81-97
      const double abs_x = std::fabs(p.x), abs_y = std::fabs(p.y), abs_z = std::fabs(p.z);
      const double maxc = std::fmax(abs_x, std::fmax(abs_y, abs_z));
      if (maxc == abs_x) return vec3(p.x, 0.0, 0.0);
      if (maxc == abs_y) return vec3(0.0, p.y, 0.0);
      return vec3(0.0, 0.0, p.z);
    }
    case CYLINDER: {  // cylinder.zig:100-112
      const double dist = p.x * p.x + p.z * p.z;
      if (dist < 1.0 && p.y >= ymax - 1e-5) return vec3(0.0, 1.0, 0.0);
      if (dist < 1.0 && p.y <= ymin + 1e-5) return vec3(0.0, -1.0, 0.0);
      return vec3(p.x, 0.0, p.z);
    }
    case CONE: {  // cone.zig:115-132
      const double dist = p.x * p.x + p.z * p.z;
      if (dist < ymax * ymax && p.y >= ymax - 1e-4) return vec3(0.0, 1.0, 0.0);
      if (dist < ymin * ymin && p.y <= ymin + 1e-4) return vec3(0.0, -1.0, 0.0);
      const double y = -zigSign(p.y) * std::sqrt(p.x * p.x + p.z * p.z);
      return vec3(p.x, y, p.z);
    }
    case TRIANGLE: return normal;  // triangle.zig:65-70
    case SMOOTH_TRIANGLE:          // triangle.zig:261-265
      counters().smooth_hits++;
      return add(add(mul(n2, h.u), mul(n3, h.v)), mul(n1, 1.0 - h.u - h.v));
    case TEST_SHAPE: return point(0.0, 0.0, 0.0);
    default: throw std::runtime_error("localNormalAt not implemented for this kind");
  }
}

// ------------------------------------------------------------------ std.math.pow
// material.zig:69 and world.zig:288 call Zig's std.math.pow(f64, x, y).  That function is part of the Zig
// standard library (the toolchain, `zig-version: master` in .github/workflows/deploy.yml), not of the
// reference tree; its published algorithm (lib/std/math/pow.zig, "ported from Go's math.Pow") is restated
// here: special cases, then x^y = x^frac(y) * x^int(y) with the integer power by binary exponentiation on
// the frexp mantissa (exponents tracked separately) and one final scalbn.  It is NOT libm's correctly
// rounded-ish pow: for shininess = 300 the two differ in the last few ulps.
inline double zig_pow(double x, double y) {
  const double inf = std::numeric_limits<double>::infinity();
  if (y == 0.0 || x == 1.0) return 1.0;
  if (std::isnan(x) || std::isnan(y)) return std::numeric_limits<double>::quiet_NaN();
  if (y == 1.0) return x;
  auto is_odd_integer = [](double v) {
    if (std::fabs(v) >= 9007199254740992.0) return false;  // 2^53: every such double is even
    double ip;
    const double fp = std::modf(v, &ip);
    return fp == 0.0 && (static_cast<long long>(ip) & 1) == 1;
  };
  if (x == 0.0) {
    if (y < 0.0) return is_odd_integer(y) ? std::copysign(inf, x) : inf;
    return is_odd_integer(y) ? x : 0.0;
  }
  if (std::isinf(y)) {
    if (x == -1.0) return 1.0;
    if ((std::fabs(x) < 1.0) == (y > 0.0)) return 0.0;
    return inf;
  }
  if (std::isinf(x)) {
    if (x < 0.0) {
      if (y < 0.0) return is_odd_integer(y) ? -0.0 : 0.0;
      return is_odd_integer(y) ? -inf : inf;
    }
    return y < 0.0 ? 0.0 : inf;
  }
  if (y == 0.5) return std::sqrt(x);
  if (y == -0.5) return 1.0 / std::sqrt(x);
  double yi;
  double yf = std::modf(std::fabs(y), &yi);
  if (yf != 0.0 && x < 0.0) return std::numeric_limits<double>::quiet_NaN();
  if (yi >= 9223372036854775808.0) return std::exp(y * std::log(x));  // 1 << 63
  double a1 = 1.0;  // the result is a1 * 2^ae
  int ae = 0;
  if (yf != 0.0) {
    if (yf > 0.5) {
      yf -= 1.0;
      yi += 1.0;
    }
    a1 = std::exp(yf * std::log(x));
  }
  int xe;
  double x1 = std::frexp(x, &xe);
  for (long long i = static_cast<long long>(yi); i != 0; i >>= 1) {
    if (xe < -(1 << 12) || (1 << 12) < xe) {  // catastrophic overflow
      ae += xe;
      break;
    }
    if (i & 1) {
      a1 *= x1;
      ae += xe;
    }
    x1 *= x1;
    xe <<= 1;
    if (x1 < 0.5) {
      x1 += x1;
      xe -= 1;
    }
  }
  if (y < 0.0) {
    a1 = 1.0 / a1;
    ae = -ae;
  }
  return std::ldexp(a1, ae);
}

// ------------------------------------------------------------------ material.zig:40-74
inline Color Material::lighting(const Light& light, const Shape* object, Tuple pt, Tuple point_to_eye, Tuple normal,
                                bool in_shadow) const {
  const Color color = pattern.patternAt(object->worldToObject(pt));  // pattern.zig:128-131
  const Color effective_color = cemul(color, light.intensity);
  const Tuple point_to_light = normalized(sub(light.position, pt));
  const Color ambient_ = cmul(effective_color, ambient);
  if (in_shadow) return ambient_;
  Color diffuse_{0.0, 0.0, 0.0};
  Color specular_{0.0, 0.0, 0.0};
  const double light_dot_normal = dot(point_to_light, normal);
  if (light_dot_normal >= 0.0) {
    diffuse_ = cmul(effective_color, diffuse * light_dot_normal);
    const Tuple reflected = reflect(point_to_light, normal);
    const double reflect_dot_eye = dot(negate(reflected), point_to_eye);
    if (reflect_dot_eye > 0.0) {
      specular_ = cmul(light.intensity, specular * zig_pow(reflect_dot_eye, shininess));
    }
  }
  return cadd(cadd(ambient_, diffuse_), specular_);
}

// ------------------------------------------------------------------ world.zig
struct PreComputations {  // world.zig:194-210
  Intersection intersection;
  Tuple point, over_point, under_point, eyev, normal;
  bool inside;
  Tuple reflectv;
  double n1, n2;

  static PreComputations make(const Intersection& hit_, const Ray& ray, const Intersections& xs) {  // :212-270
    const double epsilon = 1e-5;
    PreComputations c;
    const Tuple pt = ray.position(hit_.t);
    const Tuple eyev = negate(ray.direction);
    Tuple normal = hit_.object->normalAt(pt, hit_);
    bool inside = false;
    if (dot(normal, eyev) < 0) {
      normal = negate(normal);
      inside = true;
    }
    const Tuple over_point = add(pt, mul(normal, epsilon));
    const Tuple under_point = sub(pt, mul(normal, epsilon));
    const Tuple reflectv = reflect(ray.direction, normal);

    std::vector<const Shape*, ArenaAlloc<const Shape*>> containers;  // (allocator.alloc in world.zig:229)
    containers.reserve(xs.size());
    double n1 = 1.0, n2 = 1.0;
    for (const Intersection& item : xs) {
      const bool is_hit = item.t == hit_.t && item.object->id == hit_.object->id;
      if (is_hit && !containers.empty()) n1 = containers.back()->material.refractive_index;
      bool removed = false;
      for (size_t i = 0; i < containers.size(); ++i) {
        if (containers[i]->id == item.object->id) {
          containers.erase(containers.begin() + static_cast<long>(i));  // orderedRemove
          removed = true;
          break;
        }
      }
      if (!removed) containers.push_back(item.object);
      if (is_hit && !containers.empty()) {
        n2 = containers.back()->material.refractive_index;
        break;
      }
    }
    c.intersection = hit_;
    c.point = pt;
    c.over_point = over_point;
    c.under_point = under_point;
    c.eyev = eyev;
    c.normal = normal;
    c.inside = inside;
    c.reflectv = reflectv;
    c.n1 = n1;
    c.n2 = n2;
    return c;
  }

  double schlick() const {  // world.zig:272-289
    double cos = dot(eyev, normal);
    if (n1 > n2) {
      const double n_ratio = n1 / n2;
      const double sin2_t = n_ratio * n_ratio * (1.0 - cos * cos);
      if (sin2_t > 1.0) return 1.0;
      const double cos_t = std::sqrt(1.0 - sin2_t);
      cos = cos_t;
    }
    const double frac = (n1 - n2) / (n1 + n2);
    const double r0 = frac * frac;
    return r0 + (1.0 - r0) * zig_pow(1 - cos, 5);
  }
};

struct World {
  std::vector<Shape> objects;
  std::vector<Light> lights;

  static World defaultWorld() {  // world.zig:40-62
    World w;
    Shape s1 = Shape::make(SPHERE);
    s1.material.pattern.rgb = {0.8, 1.0, 0.6};
    s1.material.diffuse = 0.7;
    s1.material.specular = 0.2;
    Shape s2 = Shape::make(SPHERE);
    s2.setTransform(Matrix::identity().scale(0.5, 0.5, 0.5));
    w.objects.push_back(s1);
    w.objects.push_back(s2);
    w.lights.push_back({point(-10.0, 10.0, -10.0), {1.0, 1.0, 1.0}});
    return w;
  }

  Intersections intersect(const Ray& ray) const {  // world.zig:71-83
    Intersections all;
    for (const Shape& object : objects) {
      const Intersections xs = object.intersect(ray);
      all.insert(all.end(), xs.begin(), xs.end());
    }
    sortIntersections(all);
    return all;
  }

  bool isShadowed(Tuple pt, const Light& light) const {  // world.zig:126-154
    counters().shadow++;
    const Tuple direction = sub(light.position, pt);
    const double distance = magnitude(direction);
    const Ray shadow_ray{pt, normalized(direction)};
    const Intersections xs = intersect(shadow_ray);
    bool is_shadowed = false;
    long i = hit(xs);
    while (i >= 0) {
      if (xs[i].t < distance && xs[i].object->casts_shadow) {
        is_shadowed = true;
        break;
      }
      i = hit(xs, static_cast<size_t>(i) + 1);
    }
    return is_shadowed;
  }

  Color reflectedColor(const PreComputations& comps, size_t remaining) const {  // world.zig:157-167
    if (remaining == 0 || comps.intersection.object->material.reflective == 0.0) return {0.0, 0.0, 0.0};
    counters().secondary++;
    const Ray reflected{comps.over_point, comps.reflectv};
    const Color color = colorAt(reflected, remaining - 1);
    return cmul(color, comps.intersection.object->material.reflective);
  }

  Color refractedColor(const PreComputations& comps, size_t remaining) const {  // world.zig:171-189
    const double n_ratio = comps.n1 / comps.n2;
    const double cos_i = dot(comps.eyev, comps.normal);
    const double sin2_t = n_ratio * n_ratio * (1.0 - cos_i * cos_i);
    if (sin2_t > 1.0) return {0.0, 0.0, 0.0};
    if (remaining == 0 || comps.intersection.object->material.transparency == 0.0) return {0.0, 0.0, 0.0};
    const double cos_t = std::sqrt(1.0 - sin2_t);
    const Tuple direction = sub(mul(comps.normal, n_ratio * cos_i - cos_t), mul(comps.eyev, n_ratio));
    counters().secondary++;
    const Ray refracted{comps.under_point, direction};
    const Color color = colorAt(refracted, remaining - 1);
    return cmul(color, comps.intersection.object->material.transparency);
  }

  Color shadeHit(const PreComputations& comps, size_t remaining) const {  // world.zig:86-108
    Color surface{0.0, 0.0, 0.0};
    const Material& m = comps.intersection.object->material;
    for (const Light& light : lights) {
      const bool shadowed = isShadowed(comps.over_point, light);
      surface = cadd(surface, m.lighting(light, comps.intersection.object, comps.over_point, comps.eyev,
                                         comps.normal, shadowed));
    }
    const Color reflected = reflectedColor(comps, remaining);
    const Color refracted = refractedColor(comps, remaining);
    if (m.reflective > 0.0 && m.transparency > 0.0) {
      const double reflectance = comps.schlick();
      return cadd(cadd(surface, cmul(reflected, reflectance)), cmul(refracted, 1.0 - reflectance));
    }
    return cadd(cadd(surface, reflected), refracted);
  }

  Color colorAt(const Ray& ray, size_t remaining) const {  // world.zig:111-121
    const Intersections xs = intersect(ray);
    const long h = hit(xs);
    if (h >= 0) {
      const PreComputations comps = PreComputations::make(xs[h], ray, xs);
      return shadeHit(comps, remaining);
    }
    return {0.0, 0.0, 0.0};
  }
};

// ------------------------------------------------------------------ camera.zig
struct Camera {
  size_t hsize, vsize;
  double fov, half_width, half_height, pixel_size;
  Matrix transform = Matrix::identity();
  Matrix inverse = Matrix::identity();

  static Camera make(size_t hsize, size_t vsize, double fov) {  // :33-52
    Camera c;
    const double half_view = std::tan(fov / 2.0);
    const double aspect = static_cast<double>(hsize) / static_cast<double>(vsize);
    double half_width = half_view * aspect;
    double half_height = half_view;
    if (aspect >= 1.0) {
      half_width = half_view;
      half_height = half_view / aspect;
    }
    c.hsize = hsize;
    c.vsize = vsize;
    c.fov = fov;
    c.half_width = half_width;
    c.half_height = half_height;
    c.pixel_size = (half_width * 2.0) / static_cast<double>(hsize);
    return c;
  }
  void setTransform(const Matrix& m) {
    transform = m;
    inverse = m.inverse();
  }
  Ray rayForPixel(size_t x, size_t y) const {  // :64-76
    const double xoffset = (static_cast<double>(x) + 0.5) * pixel_size;
    const double yoffset = (static_cast<double>(y) + 0.5) * pixel_size;
    const double world_x = half_width - xoffset;
    const double world_y = half_height - yoffset;
    const Tuple pixel = inverse.tupleMul(point(world_x, world_y, -1.0));
    const Tuple origin = inverse.tupleMul(point(0.0, 0.0, 0.0));
    const Tuple direction = normalized(sub(pixel, origin));
    return {origin, direction};
  }
};

}  // namespace orc
