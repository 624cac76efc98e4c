// rtc_math.hpp — host-side Tuple / Matrix / Color used to BUILD scenes.
//
// Mirrors the result-affecting arithmetic of the reference's
//   src/raytracer/tuple.zig, matrix.zig, color.zig
// in f64 (the reference instantiates every type with f64 for scene renders,
// src/main.zig:71).  Every expression keeps the reference's evaluation order
// and this file must be compiled with -ffp-contract=off: the Zig default float
// mode is strict IEEE (no FMA contraction).  The matrices built here end up,
// bit for bit, in the flat scene handed to the GPU.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>

namespace rtc {

constexpr double kInf = std::numeric_limits<double>::infinity();

// Error names follow the Zig error sets (matrix.zig:7, scene.zig:212, obj.zig:14-20).
struct Error : std::runtime_error {
  std::string name;
  Error(const std::string& n, const std::string& detail = "")
      : std::runtime_error(detail.empty() ? n : n + ": " + detail), name(n) {}
};

// tuple.zig:12-19 — xyzw; w==1 point, w==0 vector.
struct Tuple {
  double x = 0, y = 0, z = 0, w = 0;
  static Tuple point(double x, double y, double z) { return {x, y, z, 1.0}; }
  static Tuple vec3(double x, double y, double z) { return {x, y, z, 0.0}; }
  Tuple add(const Tuple& o) const { return {x + o.x, y + o.y, z + o.z, w + o.w}; }   // tuple.zig:54
  Tuple sub(const Tuple& o) const { return {x - o.x, y - o.y, z - o.z, w - o.w}; }   // tuple.zig:62
  Tuple negate() const { return {-x, -y, -z, -w}; }                                   // tuple.zig:72
  Tuple mul(double v) const { return {x * v, y * v, z * v, w * v}; }                  // tuple.zig:82
  Tuple div(double v) const { return {x / v, y / v, z / v, w / v}; }                  // tuple.zig:92
  double magnitude() const { return std::sqrt(x * x + y * y + z * z + w * w); }       // tuple.zig:102
  Tuple normalized() const {                                                          // tuple.zig:109
    const double mag = magnitude();
    if (mag == 0.0) return *this;
    return div(mag);
  }
  double dot(const Tuple& o) const { return x * o.x + y * o.y + z * o.z + w * o.w; }  // tuple.zig:121
  Tuple cross(const Tuple& o) const {                                                 // tuple.zig:128 (left-handed)
    return vec3(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x);
  }
  Tuple reflect(const Tuple& n) const { return sub(n.mul(2.0 * dot(n))); }            // tuple.zig:139
  bool bitEqual(const Tuple& o) const { return x == o.x && y == o.y && z == o.z && w == o.w; }
};

struct Color {
  double r = 0, g = 0, b = 0;
};

// color.zig:61-71 — channel -> u8 with round-half-away and clamping.
inline uint8_t clampChannel(double channel) {
  double t = std::round(channel * 255);
  if (!(t >= 0)) return 0;  // negative (or NaN, which Zig's @intFromFloat would trap on)
  if (t > 255) return 255;
  return static_cast<uint8_t>(t);
}

// matrix.zig — square row-major matrices; only N=4 (and the 3,2 minors) are needed.
template <int N>
struct Mat {
  double d[N][N];

  double det() const;
  Mat<N - 1> submatrix(int row, int col) const {  // matrix.zig:157
    Mat<N - 1> s;
    for (int r = 0; r < N; ++r) {
      if (r == row) continue;
      for (int c = 0; c < N; ++c) {
        if (c == col) continue;
        s.d[r - (r > row)][c - (c > col)] = d[r][c];
      }
    }
    return s;
  }
  double minor_(int row, int col) const { return submatrix(row, col).det(); }   // matrix.zig:174
  double cofactor(int row, int col) const {                                     // matrix.zig:179
    return ((row + col) % 2 == 0) ? minor_(row, col) : -minor_(row, col);
  }
};

template <>
struct Mat<1> {
  double d[1][1];
  double det() const { return d[0][0]; }
};

template <int N>
double Mat<N>::det() const {  // matrix.zig:188 — cofactor expansion along row 0
  double det_ = 0.0;
  if constexpr (N == 2) {
    det_ = d[0][0] * d[1][1] - d[0][1] * d[1][0];
  } else {
    for (int col = 0; col < N; ++col) det_ += d[0][col] * cofactor(0, col);
  }
  return det_;
}

struct ShearArgs {  // matrix.zig:296-303
  double xy = 0, xz = 0, yx = 0, yz = 0, zx = 0, zy = 0;
};

struct Matrix4 : Mat<4> {
  static constexpr double tolerance = 1e-5;  // matrix.zig:13

  static Matrix4 zero() {
    Matrix4 m;
    for (auto& row : m.d)
      for (double& v : row) v = 0.0;
    return m;
  }
  static Matrix4 identity() {
    Matrix4 m = zero();
    for (int i = 0; i < 4; ++i) m.d[i][i] = 1.0;
    return m;
  }
  static Matrix4 rows(const double (&r)[4][4]) {
    Matrix4 m;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) m.d[i][j] = r[i][j];
    return m;
  }

  Matrix4 mul(const Matrix4& o) const {  // matrix.zig:106
    Matrix4 r;
    for (int row = 0; row < 4; ++row)
      for (int col = 0; col < 4; ++col) {
        double sum = 0;
        for (int i = 0; i < 4; ++i) sum += d[row][i] * o.d[i][col];
        r.d[row][col] = sum;
      }
    return r;
  }
  Tuple tupleMul(const Tuple& t) const {  // matrix.zig:124 — four row dots, w included
    auto rowdot = [&](int r) { return d[r][0] * t.x + d[r][1] * t.y + d[r][2] * t.z + d[r][3] * t.w; };
    return {rowdot(0), rowdot(1), rowdot(2), rowdot(3)};
  }
  Matrix4 transpose() const {  // matrix.zig:143
    Matrix4 t;
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) t.d[r][c] = d[c][r];
    return t;
  }
  Matrix4 add(const Matrix4& o) const {
    Matrix4 r;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) r.d[i][j] = d[i][j] + o.d[i][j];
    return r;
  }
  Matrix4 scalarMul(double v) const {
    Matrix4 r;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) r.d[i][j] = d[i][j] * v;
    return r;
  }
  // matrix.zig:202-217 — cofactor inverse; NotInvertible iff |det| < 1e-5.
  Matrix4 inverse() const {
    const double det_ = det();
    if (std::fabs(det_) < tolerance) throw Error("NotInvertible");
    Matrix4 inv;
    for (int row = 0; row < 4; ++row)
      for (int col = 0; col < 4; ++col) inv.d[col][row] = cofactor(row, col) / det_;
    return inv;
  }
  // Fluent transforms LEFT-multiply the running matrix (matrix.zig:222-325).
  Matrix4 translate(double x, double y, double z) const {
    return rows({{1, 0, 0, x}, {0, 1, 0, y}, {0, 0, 1, z}, {0, 0, 0, 1}}).mul(*this);
  }
  Matrix4 scale(double x, double y, double z) const {
    return rows({{x, 0, 0, 0}, {0, y, 0, 0}, {0, 0, z, 0}, {0, 0, 0, 1}}).mul(*this);
  }
  Matrix4 rotateX(double a) const {
    return rows({{1, 0, 0, 0}, {0, std::cos(a), -std::sin(a), 0}, {0, std::sin(a), std::cos(a), 0}, {0, 0, 0, 1}}).mul(*this);
  }
  Matrix4 rotateY(double a) const {
    return rows({{std::cos(a), 0, std::sin(a), 0}, {0, 1, 0, 0}, {-std::sin(a), 0, std::cos(a), 0}, {0, 0, 0, 1}}).mul(*this);
  }
  Matrix4 rotateZ(double a) const {
    return rows({{std::cos(a), -std::sin(a), 0, 0}, {std::sin(a), std::cos(a), 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}).mul(*this);
  }
  Matrix4 rotate(const Tuple& axis, double angle) const {  // matrix.zig:283-294 (Rodrigues)
    Matrix4 C = rows({{0, -axis.z, axis.y, 0}, {axis.z, 0, -axis.x, 0}, {-axis.y, axis.x, 0, 0}, {0, 0, 0, 0}});
    Matrix4 rot = identity().add(C.scalarMul(std::sin(angle))).add(C.mul(C).scalarMul(1.0 - std::cos(angle)));
    rot.d[3][3] = 1.0;
    return rot.mul(*this);
  }
  Matrix4 shear(const ShearArgs& a) const {
    return rows({{1, a.xy, a.xz, 0}, {a.yx, 1, a.yz, 0}, {a.zx, a.zy, 1, 0}, {0, 0, 0, 1}}).mul(*this);
  }
  // matrix.zig:54-67 — uses the left-handed cross.
  static Matrix4 viewTransform(const Tuple& from, const Tuple& to, const Tuple& up) {
    const Tuple forward = to.sub(from).normalized();
    const Tuple left = forward.cross(up.normalized());
    const Tuple true_up = left.cross(forward);
    Matrix4 orientation = rows({{left.x, left.y, left.z, 0},
                                {true_up.x, true_up.y, true_up.z, 0},
                                {-forward.x, -forward.y, -forward.z, 0},
                                {0, 0, 0, 1}});
    return orientation.mul(identity().translate(-from.x, -from.y, -from.z));
  }
  bool bitEqual(const Matrix4& o) const {
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j)
        if (!(d[i][j] == o.d[i][j])) return false;
    return true;
  }
  bool approxEqual(const Matrix4& o) const {  // matrix.zig:70
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j)
        if (std::fabs(d[i][j] - o.d[i][j]) > tolerance) return false;
    return true;
  }
};

}  // namespace rtc
