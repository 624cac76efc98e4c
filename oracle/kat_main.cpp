// kat_main.cpp — pins the CPU oracle (rtc_oracle.hpp) to the reference's own
// known-answer tests.  TEST INFRASTRUCTURE.
//
// Every numeric KAT the reference's inline `test` blocks hold for the hot path
//  is restated here against the oracle, in f64.  The
// reference runs most of them in f32 with tolerance 1e-5; expected values are
// the reference's literals and the tolerance is the reference's 1e-5, with ONE
// exception noted at the case (cylinder.zig:175: the literals 6.80800/7.08869
// are f32 round-off; f64 gives the book's 6.80798/7.08872).
//
// Output: one line per case, "KAT <file>:<line> <name> PASS|FAIL [detail]";
// exit status 1 if any case failed.  tests/test_oracle_kats.py parses it.
#include <cstdio>
#include <string>

#include "rtc_oracle.hpp"
#include "rtc_oracle_scene.hpp"

using namespace orc;

static int g_failed = 0, g_total = 0;

static void report(const char* where, const std::string& name, bool ok, const std::string& detail = "") {
  ++g_total;
  if (!ok) ++g_failed;
  std::printf("KAT %s %s %s%s%s\n", where, name.c_str(), ok ? "PASS" : "FAIL", detail.empty() ? "" : " ",
              detail.c_str());
}
static std::string fmt(Tuple t) {
  char b[160];
  std::snprintf(b, sizeof b, "(%.8g,%.8g,%.8g,%.8g)", t.x, t.y, t.z, t.w);
  return b;
}
static std::string fmt(Color c) {
  char b[160];
  std::snprintf(b, sizeof b, "(%.8g,%.8g,%.8g)", c.r, c.g, c.b);
  return b;
}
// tol == 0 demands exact equality (the reference uses expectEqual there).
static bool within(double a, double b, double tol) { return std::fabs(a - b) <= tol; }
static void expectTuple(const char* where, const std::string& name, Tuple got, Tuple want, double tol = 1e-5) {
  const bool ok = within(got.x, want.x, tol) && within(got.y, want.y, tol) && within(got.z, want.z, tol) &&
                  within(got.w, want.w, tol);
  report(where, name, ok, "got " + fmt(got) + " want " + fmt(want));
}
static void expectColor(const char* where, const std::string& name, Color got, Color want, double tol = 1e-5) {
  const bool ok = within(got.r, want.r, tol) && within(got.g, want.g, tol) && within(got.b, want.b, tol);
  report(where, name, ok, "got " + fmt(got) + " want " + fmt(want));
}
static void expectNear(const char* where, const std::string& name, double got, double want, double tol = 1e-5) {
  char b[96];
  std::snprintf(b, sizeof b, "got %.10g want %.10g", got, want);
  report(where, name, std::fabs(got - want) <= tol, b);
}
static void expectTrue(const char* where, const std::string& name, bool v) { report(where, name, v); }

static const double PI = 3.14159265358979323846;
static const double RS2 = 1.0 / std::sqrt(2.0);

static Matrix M(std::initializer_list<double> v) { return Matrix::make(v); }

// ------------------------------------------------------------------------------------------
static void tupleKats() {  // tuple.zig:146-216
  const Tuple a = vec3(1, 2, 3), b = vec3(2, 3, 4);
  expectNear("tuple.zig:187", "dot", dot(a, b), 20.0);
  expectTuple("tuple.zig:193", "cross_ab", cross(a, b), vec3(-1, 2, -1));
  expectTuple("tuple.zig:194", "cross_ba", cross(b, a), vec3(1, -2, 1));
  expectNear("tuple.zig:175", "magnitude", magnitude(vec3(1, 2, 3)), std::sqrt(14.0));
  expectTuple("tuple.zig:181", "normalize", normalized(vec3(1, 2, 3)), vec3(0.26726, 0.53452, 0.80178));
  expectTuple("tuple.zig:203", "reflect_45", reflect(vec3(1, -1, 0), vec3(0, 1, 0)), vec3(1, 1, 0));
  expectTuple("tuple.zig:209", "reflect_slanted", reflect(vec3(0, -1, 0), vec3(RS2, RS2, 0)), vec3(1, 0, 0));
}

static void matrixKats() {  // matrix.zig:330-683
  const Matrix a = M({1, 2, 3, 4, 5, 6, 7, 8, 9, 8, 7, 6, 5, 4, 3, 2});
  const Matrix b = M({-2, 1, 2, 3, 3, 2, 1, -1, 4, 3, 6, 5, 1, 2, 7, 8});
  const Matrix axb = M({20, 22, 50, 48, 44, 54, 114, 108, 40, 58, 110, 102, 16, 26, 46, 42});
  expectTrue("matrix.zig:352", "mul", a.mul(b).approxEqual(axb));
  const Matrix a2 = M({1, 2, 3, 4, 2, 4, 4, 2, 8, 6, 4, 1, 0, 0, 0, 1});
  expectTuple("matrix.zig:378", "tupleMul", a2.tupleMul(Tuple{1, 2, 3, 1}), Tuple{18, 24, 33, 1});
  const Matrix t = M({0, 9, 3, 0, 9, 8, 0, 8, 1, 8, 5, 3, 0, 0, 5, 8});
  expectTrue("matrix.zig:395", "transpose", t.transpose().approxEqual(M({0, 9, 1, 0, 9, 8, 8, 0, 3, 0, 5, 5, 0, 8, 3, 8})));
  const double m3[3][3] = {{1, 2, 6}, {-5, 8, -4}, {2, 6, 4}};
  expectNear("matrix.zig:437", "det3", det3(m3), -196.0);
  const Matrix d4 = M({-2, -8, 3, 5, -3, 1, 7, 3, 1, 2, -9, 6, -6, 7, 7, -9});
  expectNear("matrix.zig:446", "cofactor00", d4.cofactor(0, 0), 690.0);
  expectNear("matrix.zig:447", "cofactor01", d4.cofactor(0, 1), 447.0);
  expectNear("matrix.zig:448", "cofactor02", d4.cofactor(0, 2), 210.0);
  expectNear("matrix.zig:449", "cofactor03", d4.cofactor(0, 3), 51.0);
  expectNear("matrix.zig:450", "det4", d4.det(), -4071.0);
  const Matrix ia = M({8, -5, 9, 2, 7, 5, 6, 1, -6, 0, 9, 6, -3, 0, -9, -4});
  expectTrue("matrix.zig:468", "inverse_a",
             ia.inverse().approxEqual(M({-0.15385, -0.15385, -0.28205, -0.53846, -0.07692, 0.12308, 0.02564, 0.03077,
                                         0.35897, 0.35897, 0.43590, 0.92308, -0.69231, -0.69231, -0.76923, -1.92308})));
  const Matrix ib = M({9, 3, 0, 9, -5, -2, -6, -3, -4, 9, 6, 4, -7, 6, 6, 2});
  expectTrue("matrix.zig:484", "inverse_b",
             ib.inverse().approxEqual(M({-0.04074, -0.07778, 0.14444, -0.22222, -0.07778, 0.03333, 0.36667, -0.33333,
                                         -0.02901, -0.14630, -0.10926, 0.12963, 0.17778, 0.06667, -0.26667, 0.33333})));
  const Matrix c = M({3, -9, 7, 3, 3, -8, 2, -9, -4, 4, 4, 1, -6, 5, -1, 1});
  const Matrix d = M({8, 2, 2, 2, 3, -1, 7, 0, 7, 0, 5, 4, 6, -2, 0, 5});
  expectTrue("matrix.zig:500", "mul_by_inverse", c.mul(d).mul(d.inverse()).approxEqual(c));
  bool threw = false;
  try {
    M({-4, 2, -2, -3, 9, 6, 2, 6, 0, -5, 1, -5, 0, 0, 0, 0}).inverse();
  } catch (const std::exception&) {
    threw = true;
  }
  expectTrue("matrix.zig:204", "not_invertible", threw);

  const Matrix I = Matrix::identity();
  Matrix tr = I.translate(5, -3, 2);
  expectTuple("matrix.zig:507", "translate", tr.tupleMul(point(-3, 4, 5)), point(2, 1, 7));
  expectTuple("matrix.zig:510", "translate_inv", tr.inverse().tupleMul(point(-3, 4, 5)), point(-8, 7, 3));
  expectTuple("matrix.zig:513", "translate_vec", tr.inverse().tupleMul(vec3(3, 4, 5)), vec3(3, 4, 5));
  tr = I.scale(2, 3, 4);
  expectTuple("matrix.zig:518", "scale_pt", tr.tupleMul(point(-4, 6, 8)), point(-8, 18, 32));
  expectTuple("matrix.zig:523", "scale_inv", tr.inverse().tupleMul(vec3(-4, 6, 8)), vec3(-2, 2, 2));
  expectTuple("matrix.zig:531", "rotx_half", I.rotateX(PI / 4).tupleMul(point(0, 1, 0)), point(0, RS2, RS2));
  expectTuple("matrix.zig:535", "rotx_full", I.rotateX(PI / 4).rotateX(PI / 4).tupleMul(point(0, 1, 0)), point(0, 0, 1));
  expectTuple("matrix.zig:538", "rotx_inv", I.rotateX(PI / 4).inverse().tupleMul(point(0, 1, 0)), point(0, RS2, -RS2));
  expectTuple("matrix.zig:543", "roty_half", I.rotateY(PI / 4).tupleMul(point(0, 0, 1)), point(RS2, 0, RS2));
  expectTuple("matrix.zig:547", "roty_full", I.rotateY(PI / 4).rotateY(PI / 4).tupleMul(point(0, 0, 1)), point(1, 0, 0));
  expectTuple("matrix.zig:551", "rotz_half", I.rotateZ(PI / 4).tupleMul(point(0, 1, 0)), point(-RS2, RS2, 0));
  expectTuple("matrix.zig:555", "rotz_full", I.rotateZ(PI / 4).rotateZ(PI / 4).tupleMul(point(0, 1, 0)), point(-1, 0, 0));
  const Tuple p = point(2, 3, 4);
  expectTuple("matrix.zig:560", "shear_xy", I.shear(1, 0, 0, 0, 0, 0).tupleMul(p), point(5, 3, 4));
  expectTuple("matrix.zig:563", "shear_xz", I.shear(0, 1, 0, 0, 0, 0).tupleMul(p), point(6, 3, 4));
  expectTuple("matrix.zig:566", "shear_yx", I.shear(0, 0, 1, 0, 0, 0).tupleMul(p), point(2, 5, 4));
  expectTuple("matrix.zig:569", "shear_yz", I.shear(0, 0, 0, 1, 0, 0).tupleMul(p), point(2, 7, 4));
  expectTuple("matrix.zig:572", "shear_zx", I.shear(0, 0, 0, 0, 1, 0).tupleMul(p), point(2, 3, 6));
  expectTuple("matrix.zig:575", "shear_zy", I.shear(0, 0, 0, 0, 0, 1).tupleMul(p), point(2, 3, 7));
  expectTuple("matrix.zig:581", "chained", I.rotateX(PI / 2).scale(5, 5, 5).translate(10, 5, 7).tupleMul(point(1, 0, 1)),
              point(15, 0, 7));

  expectTrue("matrix.zig:590", "view_default", Matrix::viewTransform(point(0, 0, 0), point(0, 0, -1), vec3(0, 1, 0)).approxEqual(I));
  expectTrue("matrix.zig:602", "view_positive_z",
             Matrix::viewTransform(point(0, 0, 0), point(0, 0, 1), vec3(0, 1, 0)).approxEqual(I.scale(-1, 1, -1)));
  expectTrue("matrix.zig:614", "view_moves_world",
             Matrix::viewTransform(point(0, 0, 8), point(0, 0, 0), vec3(0, 1, 0)).approxEqual(I.translate(0, 0, -8)));
  expectTrue("matrix.zig:633", "view_arbitrary",
             Matrix::viewTransform(point(1, 3, 2), point(4, -2, 8), vec3(1, 1, 0))
                 .approxEqual(M({-0.50709, 0.50709, 0.67612, -2.36643, 0.76772, 0.60609, 0.12122, -2.82843, -0.35857,
                                 0.59761, -0.71714, 0.00000, 0.00000, 0.00000, 0.00000, 1.00000})));
}

static void rayKats() {  // ray.zig:37-64
  const Ray r{point(2, 3, 4), vec3(1, 0, 0)};
  expectTuple("ray.zig:45", "position_0", r.position(0), point(2, 3, 4));
  expectTuple("ray.zig:46", "position_1", r.position(1), point(3, 3, 4));
  expectTuple("ray.zig:47", "position_m1", r.position(-1), point(1, 3, 4));
  expectTuple("ray.zig:48", "position_2.5", r.position(2.5), point(4.5, 3, 4));
  const Ray r2{point(1, 2, 3), vec3(0, 1, 0)};
  const Ray t = r2.transform(Matrix::identity().translate(3, 4, 5));
  expectTuple("ray.zig:55", "translate_origin", t.origin, point(4, 6, 8));
  expectTuple("ray.zig:56", "translate_dir", t.direction, vec3(0, 1, 0));
  const Ray s = r2.transform(Matrix::identity().scale(2, 3, 4));
  expectTuple("ray.zig:61", "scale_origin", s.origin, point(2, 6, 12));
  expectTuple("ray.zig:62", "scale_dir", s.direction, vec3(0, 3, 0));
}

// ------------------------------------------------------------------------------------------
static void expectTs(const char* where, const std::string& name, const Intersections& xs, std::initializer_list<double> ts,
                     double tol = 1e-5) {
  bool ok = xs.size() == ts.size();
  std::string detail = "n=" + std::to_string(xs.size());
  size_t i = 0;
  for (double t : ts) {
    if (ok) ok = std::fabs(xs[i].t - t) <= tol;
    ++i;
  }
  for (const auto& x : xs) detail += " " + std::to_string(x.t);
  report(where, name, ok, detail);
}

static void sphereKats() {  // sphere.zig:67-184
  Shape s = Shape::make(SPHERE);
  expectTs("sphere.zig:71", "through_center", s.intersect({point(0, 0, -5), vec3(0, 0, 1)}), {4.0, 6.0});
  expectTs("sphere.zig:87", "tangent", s.intersect({point(0, 1, -5), vec3(0, 0, 1)}), {5.0, 5.0});
  expectTs("sphere.zig:103", "miss", s.intersect({point(0, 2, -5), vec3(0, 0, 1)}), {});
  expectTs("sphere.zig:112", "inside", s.intersect({point(0, 0, 0), vec3(0, 0, 1)}), {-1.0, 1.0});
  expectTs("sphere.zig:128", "behind", s.intersect({point(0, 0, 5), vec3(0, 0, 1)}), {-6.0, -4.0});
  Shape s2 = Shape::make(SPHERE);
  s2.setTransform(Matrix::identity().scale(2, 2, 2));
  expectTs("sphere.zig:144", "scaled", s2.intersect({point(0, 0, -5), vec3(0, 0, 1)}), {3.0, 7.0});
  Shape s3 = Shape::make(SPHERE);
  s3.setTransform(Matrix::identity().translate(5, 0, 0));
  expectTs("sphere.zig:161", "translated_miss", s3.intersect({point(0, 0, -5), vec3(0, 0, 1)}), {});
  Shape n1 = Shape::make(SPHERE);
  n1.setTransform(Matrix::identity().translate(0, 1, 0));
  expectTuple("sphere.zig:173", "normal_translated", n1.normalAt(point(0, 1.70711, -0.70711), {}), vec3(0, 0.70711, -0.70711));
  Shape n2 = Shape::make(SPHERE);
  n2.setTransform(Matrix::identity().rotateZ(PI / 5.0).scale(1, 0.5, 1));
  expectTuple("sphere.zig:179", "normal_transformed", n2.normalAt(point(0, RS2, -RS2), {}), vec3(0, 0.97014, -0.24254));
}

static void planeKats() {  // plane.zig:58-107
  Shape p = Shape::make(PLANE);
  expectTs("plane.zig:62", "parallel", p.intersect({point(0, 10, 0), vec3(0, 0, 1)}), {});
  expectTs("plane.zig:70", "coplanar", p.intersect({point(0, 0, 0), vec3(0, 0, 1)}), {});
  expectTs("plane.zig:78", "from_above", p.intersect({point(0, 1, 0), vec3(0, -1, 0)}), {1.0});
  expectTs("plane.zig:88", "from_below", p.intersect({point(0, -1, 0), vec3(0, 1, 0)}), {1.0});
  expectTuple("plane.zig:100", "normal_a", p.normalAt(point(0, 0, 0), {}), vec3(0, 1, 0));
  expectTuple("plane.zig:101", "normal_b", p.normalAt(point(10, 0, -10), {}), vec3(0, 1, 0));
  expectTuple("plane.zig:102", "normal_c", p.normalAt(point(-5, 0, 150), {}), vec3(0, 1, 0));
}

static void cubeKats() {  // cube.zig:126-209
  Shape c = Shape::make(CUBE);
  struct Hit { Tuple o, d; double t1, t2; };
  const Hit hits[] = {{point(5, 0.5, 0), vec3(-1, 0, 0), 4, 6},  {point(-5, 0.5, 0), vec3(1, 0, 0), 4, 6},
                      {point(0.5, 5, 0), vec3(0, -1, 0), 4, 6},  {point(0.5, -5, 0), vec3(0, 1, 0), 4, 6},
                      {point(0.5, 0, -5), vec3(0, 0, 1), 4, 6},  {point(0, 0.5, 0), vec3(0, 0, 1), -1, 1}};
  int i = 0;
  for (const Hit& h : hits) expectTs("cube.zig:129", "hit_" + std::to_string(i++), c.intersect({h.o, h.d}), {h.t1, h.t2});
  const Ray misses[] = {{point(-2, 0, 0), vec3(0.2673, 0.5345, 0.8018)}, {point(0, -2, 0), vec3(0.8018, 0.2673, 0.5345)},
                        {point(0, 0, -2), vec3(0.5345, 0.8018, 0.2673)}, {point(2, 0, 2), vec3(0, 0, -1)},
                        {point(0, 2, 2), vec3(0, -1, 0)},                {point(2, 2, 0), vec3(-1, 0, 0)}};
  i = 0;
  for (const Ray& r : misses) expectTs("cube.zig:167", "miss_" + std::to_string(i++), c.intersect(r), {});
  struct N { Tuple p, n; };
  const N normals[] = {{point(1, 0.5, -0.8), vec3(1, 0, 0)},  {point(-1, -0.2, 0.9), vec3(-1, 0, 0)},
                       {point(-0.4, 1, -0.1), vec3(0, 1, 0)}, {point(0.3, -1, -0.7), vec3(0, -1, 0)},
                       {point(-0.6, 0.3, 1), vec3(0, 0, 1)},  {point(0.4, 0.4, -1), vec3(0, 0, -1)},
                       {point(1, 1, 1), vec3(1, 0, 0)},       {point(-1, -1, -1), vec3(-1, 0, 0)}};
  i = 0;
  for (const N& n : normals) expectTuple("cube.zig:200", "normal_" + std::to_string(i++), c.normalAt(n.p, {}), n.n);
}

static void cylinderKats() {  // cylinder.zig:136-332
  Shape cyl = Shape::make(CYLINDER);
  const Ray misses[] = {{point(1, 0, 0), vec3(0, 1, 0)}, {point(0, 0, 0), vec3(0, 1, 0)}, {point(0, 0, -5), vec3(1, 1, 1)}};
  int i = 0;
  for (const Ray& r : misses)
    expectTs("cylinder.zig:136", "miss_" + std::to_string(i++), cyl.intersect({r.origin, normalized(r.direction)}), {});
  expectTs("cylinder.zig:169", "tangent", cyl.intersect({point(1, 0, -5), normalized(vec3(0, 0, 1))}), {5.0, 5.0});
  expectTs("cylinder.zig:172", "through", cyl.intersect({point(0, 0, -5), normalized(vec3(0, 0, 1))}), {4.0, 6.0});
  expectTs("cylinder.zig:175", "skewed", cyl.intersect({point(0.5, 0, -5), normalized(vec3(0.1, 1, 1))}), {6.80798, 7.08872},
           1e-4);  // reference literals 6.80800 / 7.08869 are the f32 results; book (f64): 6.80798 / 7.08872
  struct N { Tuple p, n; };
  const N normals[] = {{point(1, 0, 0), vec3(1, 0, 0)}, {point(0, 5, -1), vec3(0, 0, -1)},
                       {point(0, -2, 1), vec3(0, 0, 1)}, {point(-1, 1, 0), vec3(-1, 0, 0)}};
  i = 0;
  for (const N& n : normals) expectTuple("cylinder.zig:187", "normal_" + std::to_string(i++), cyl.normalAt(n.p, {}), n.n);
  expectTrue("cylinder.zig:205", "default_min_max", cyl.ymin == -INF && cyl.ymax == INF);

  Shape tc = Shape::make(CYLINDER);
  tc.ymin = 1.0;
  tc.ymax = 2.0;
  struct C { Tuple o, d; size_t count; };
  const C trunc[] = {{point(0, 1.5, 0), vec3(0.1, 1, 0), 0}, {point(0, 3, -5), vec3(0, 0, 1), 0},
                     {point(0, 0, -5), vec3(0, 0, 1), 0},    {point(0, 2, -5), vec3(0, 0, 1), 0},
                     {point(0, 1, -5), vec3(0, 0, 1), 0},    {point(0, 1.5, -2), vec3(0, 0, 1), 2}};
  i = 0;
  for (const C& c : trunc)
    expectTrue("cylinder.zig:228", "truncated_" + std::to_string(i++), tc.intersect({c.o, normalized(c.d)}).size() == c.count);
  Shape cc = tc;
  cc.closed = true;
  const C caps[] = {{point(0, 3, 0), vec3(0, -1, 0), 2},  {point(0, 3, -2), vec3(0, -1, 2), 2},
                    {point(0, 4, -2), vec3(0, -1, 1), 2}, {point(0, 0, -2), vec3(0, 1, 2), 2},
                    {point(0, -1, -2), vec3(0, 1, 1), 2}};
  i = 0;
  for (const C& c : caps)
    expectTrue("cylinder.zig:273", "closed_caps_" + std::to_string(i++), cc.intersect({c.o, normalized(c.d)}).size() == c.count);
  const N capn[] = {{point(0, 1, 0), vec3(0, -1, 0)},   {point(0.5, 1, 0), vec3(0, -1, 0)}, {point(0, 1, 0.5), vec3(0, -1, 0)},
                    {point(0, 2, 0), vec3(0, 1, 0)},    {point(0.5, 2, 0), vec3(0, 1, 0)},  {point(0, 2, 0.5), vec3(0, 1, 0)}};
  i = 0;
  for (const N& n : capn) expectTuple("cylinder.zig:308", "cap_normal_" + std::to_string(i++), cc.normalAt(n.p, {}), n.n);
}

static void coneKats() {  // cone.zig:155-233 (Cone(T).tolerance = 1e-4 where the reference compares t)
  Shape cone = Shape::make(CONE);
  expectTs("cone.zig:158", "through_apex", cone.intersect({point(0, 0, -5), normalized(vec3(0, 0, 1))}), {5.0, 5.0}, 1e-4);
  expectTs("cone.zig:162", "diagonal", cone.intersect({point(0, 0, -5), normalized(vec3(1, 1, 1))}), {8.66025, 8.66025}, 1e-4);
  expectTs("cone.zig:166", "two_roots", cone.intersect({point(1, 1, -5), normalized(vec3(-0.5, -1, 1))}), {4.55006, 49.44994}, 1e-4);
  // a ray parallel to one half: ONE entry, appended without the min < y < max filter (cone.zig:79-86; the quirk the
  // kernel's root cull and the fuzz seed 410 are about)
  expectTs("cone.zig:173", "parallel_to_a_half", cone.intersect({point(0, 0, -1), normalized(vec3(0, 1, 1))}), {0.35355}, 1e-4);
  {
    Shape trunc = Shape::make(CONE);  // the same ray against a cone truncated far away from the root: still reported
    trunc.ymin = 5.0;
    trunc.ymax = 6.0;
    expectTs("cone.zig:79", "parallel_root_ignores_truncation", trunc.intersect({point(0, 0, -1), normalized(vec3(0, 1, 1))}), {0.35355}, 1e-4);
  }
  Shape cc = Shape::make(CONE);
  cc.ymin = -0.5;
  cc.ymax = 0.5;
  cc.closed = true;
  struct C { Tuple o, d; size_t count; };
  const C caps[] = {{point(0, 0, -5), vec3(0, 1, 0), 0}, {point(0, 0, -0.25), vec3(0, 1, 1), 2}, {point(0, 0, -0.25), vec3(0, 1, 0), 4}};
  int i = 0;
  for (const C& c : caps)
    expectTrue("cone.zig:204", "end_caps_" + std::to_string(i++), cc.intersect({c.o, normalized(c.d)}).size() == c.count);
  struct N { Tuple p, n; };
  const N normals[] = {{point(0, 0, 0), vec3(0, 0, 0)}, {point(1, 1, 1), normalized(vec3(1, -std::sqrt(2.0), 1))},
                       {point(-1, -1, 0), normalized(vec3(-1, 1, 0))}};
  i = 0;
  for (const N& n : normals) expectTuple("cone.zig:227", "normal_" + std::to_string(i++), cone.normalAt(n.p, {}), n.n);
  expectTrue("cone.zig:241", "default_min_max", cone.ymin == -INF && cone.ymax == INF && !cone.closed);
}

static void triangleKats() {  // triangle.zig:83-196, 289-342
  const Shape t = Shape::triangle(point(0, 1, 0), point(-1, 0, 0), point(1, 0, 0));
  expectTuple("triangle.zig:94", "e1", t.e1, vec3(-1, -1, 0));
  expectTuple("triangle.zig:95", "e2", t.e2, vec3(1, -1, 0));
  expectTuple("triangle.zig:96", "normal", t.normal, vec3(0, 0, -1));
  expectTuple("triangle.zig:107", "normalAt", t.normalAt(point(-0.5, 0.75, 0), {}), t.normal);
  expectTs("triangle.zig:118", "parallel", t.intersect({point(0, -1, -2), vec3(0, 1, 0)}), {});
  expectTs("triangle.zig:133", "miss_p1p3", t.intersect({point(1, 1, -2), vec3(0, 0, 1)}), {});
  expectTs("triangle.zig:149", "miss_p1p2", t.intersect({point(-1, 1, -2), vec3(0, 0, 1)}), {});
  expectTs("triangle.zig:165", "miss_p2p3", t.intersect({point(0, -1, -2), vec3(0, 0, 1)}), {});
  expectTs("triangle.zig:181", "strike", t.intersect({point(0, 0.5, -2), vec3(0, 0, 1)}), {2.0}, 0.0);
  const Shape st = Shape::smoothTriangle(point(0, 1, 0), point(-1, 0, 0), point(1, 0, 0), vec3(0, 1, 0), vec3(-1, 0, 0),
                                         vec3(1, 0, 0));
  const Intersections xs = st.intersect({point(-0.2, 0.3, -2), vec3(0, 0, 1)});
  expectTrue("triangle.zig:305", "smooth_hit", xs.size() == 1);
  if (xs.size() == 1) {
    expectNear("triangle.zig:315", "smooth_u", xs[0].u, 0.45);
    expectNear("triangle.zig:315", "smooth_v", xs[0].v, 0.25);
  }
  Intersection i{1.0, &st, 0.45, 0.25};
  expectTuple("triangle.zig:323", "smooth_normal", st.normalAt(point(0, 0, 0), i), vec3(-0.5547, 0.83205, 0));
  Intersections one{i};
  const PreComputations comps = PreComputations::make(i, {point(-0.2, 0.3, -2), vec3(0, 0, 1)}, one);
  expectTuple("triangle.zig:341", "smooth_prepared_normal", comps.normal, vec3(-0.5547, 0.83205, 0));
}

static Shape groupOf(std::vector<Shape> kids, Tuple mn, Tuple mx) {
  Shape g = Shape::make(GROUP);
  g.children = std::move(kids);
  g.bmin = mn;
  g.bmax = mx;
  return g;
}

static void groupAndBoxKats() {  // group.zig:163-223, bounding_box.zig:254-360
  expectTs("group.zig:163", "empty_group", Shape::make(GROUP).intersect({point(0, 0, 0), vec3(0, 0, 1)}), {});
  Shape s1 = Shape::make(SPHERE);
  Shape s2 = Shape::make(SPHERE);
  s2.setTransform(Matrix::identity().translate(0, 0, -3));
  Shape s3 = Shape::make(SPHERE);
  s3.setTransform(Matrix::identity().translate(5, 0, 0));
  // box = merge of the three children's parent-space bounds (group.zig:75-78)
  const Shape g = groupOf({s1, s2, s3}, point(-1, -1, -4), point(6, 1, 1));
  const Intersections xs = g.intersect({point(0, 0, -5), vec3(0, 0, 1)});
  bool ok = xs.size() == 4;
  if (ok) ok = xs[0].object->id == s2.id && xs[1].object->id == s2.id && xs[2].object->id == s1.id && xs[3].object->id == s1.id;
  expectTrue("group.zig:177", "nonempty_group_order", ok);
  // transformed group: group scale(2) pushed onto child translate(5,0,0) (group.zig:201-216)
  Shape ts = Shape::make(SPHERE);
  ts.setTransform(Matrix::identity().scale(2, 2, 2).mul(Matrix::identity().translate(5, 0, 0)));
  const Shape tg = groupOf({ts}, point(8, -2, -2), point(12, 2, 2));
  expectTrue("group.zig:206", "transformed_group", tg.intersect({point(10, 0, -10), vec3(0, 0, 1)}).size() == 2);

  auto boxHit = [](Tuple mn, Tuple mx, Tuple o, Tuple d) {
    Shape b = Shape::make(BOUNDING_BOX);
    b.bmin = mn;
    b.bmax = mx;
    return !b.intersect({o, normalized(d)}).empty();
  };
  struct B { Tuple o, d; bool r; };
  const B unit[] = {{point(5, 0.5, 0), vec3(-1, 0, 0), true},  {point(-5, 0.5, 0), vec3(1, 0, 0), true},
                    {point(0.5, -5, 0), vec3(0, -1, 0), true}, {point(0.5, -5, 0), vec3(0, 1, 0), true},
                    {point(0.5, 0, 5), vec3(0, 0, -1), true},  {point(0.5, 0, -5), vec3(0, 0, 1), true},
                    {point(0, 0.5, 0), vec3(0, 0, 1), true},   {point(-2, 0, 0), vec3(2, 4, 6), false},
                    {point(0, -2, 0), vec3(6, 2, 4), false},   {point(0, 0, -2), vec3(4, 6, 2), false},
                    {point(2, 0, 2), vec3(0, 0, -1), false},   {point(0, 2, 2), vec3(0, -1, 0), false},
                    {point(2, 2, 0), vec3(-1, 0, 0), false}};
  int i = 0;
  for (const B& b : unit)
    expectTrue("bounding_box.zig:289", "aabb_unit_" + std::to_string(i++),
               boxHit(point(-1, -1, -1), point(1, 1, 1), b.o, b.d) == b.r);
  const B nc[] = {{point(15, 1, 2), vec3(-1, 0, 0), true}, {point(-5, -1, 4), vec3(1, 0, 0), true},
                  {point(7, 6, 5), vec3(0, -1, 0), true},  {point(9, -5, 6), vec3(0, 1, 0), true},
                  {point(8, 2, 12), vec3(0, 0, -1), true}, {point(6, 0, -5), vec3(0, 0, 1), true},
                  {point(8, 1, 3.5), vec3(0, 0, 1), true}, {point(9, -1, -8), vec3(2, 4, 6), false},
                  {point(8, 3, -4), vec3(6, 2, 4), false}, {point(9, -1, -2), vec3(4, 6, 2), false},
                  {point(4, 0, 9), vec3(0, 0, -1), false}, {point(8, 6, -1), vec3(0, -1, 0), false},
                  {point(12, 5, 4), vec3(-1, 0, 0), false}};
  i = 0;
  for (const B& b : nc)
    expectTrue("bounding_box.zig:347", "aabb_noncubic_" + std::to_string(i++),
               boxHit(point(5, -2, 0), point(11, 4, 7), b.o, b.d) == b.r);
}

static void nestedGroupKats() {  // shape.zig:560-617 (transforms pushed to the leaf)
  // g1.rotateY(pi/2) o g2.scale(2,2,2) o s.translate(5,0,0)
  Shape s = Shape::make(SPHERE);
  s.setTransform(Matrix::identity().rotateY(PI / 2).mul(Matrix::identity().scale(2, 2, 2).mul(Matrix::identity().translate(5, 0, 0))));
  expectTuple("shape.zig:560", "world_to_object", s.worldToObject(point(-2, 0, -10)), point(0, 0, -1));
  Shape s2 = Shape::make(SPHERE);
  s2.setTransform(Matrix::identity().rotateY(PI / 2).mul(Matrix::identity().scale(1, 2, 3).mul(Matrix::identity().translate(5, 0, 0))));
  const double r3 = 1.0 / std::sqrt(3.0);
  expectTuple("shape.zig:583", "normal_to_world", s2.normalToWorld(vec3(r3, r3, r3)), vec3(0.28571, 0.42857, -0.85714));
  expectTuple("shape.zig:607", "normal_on_child", s2.normalAt(point(1.7321, 1.1547, -5.5774), {}),
              vec3(0.2857, 0.42854, -0.85716));
}

static void refractionIndexKats() {  // shape.zig:520-558
  Shape a = Shape::glassSphere();
  a.setTransform(Matrix::identity().scale(2, 2, 2));
  a.material.refractive_index = 1.5;
  Shape b = Shape::glassSphere();
  b.setTransform(Matrix::identity().translate(0, 0, -0.25));
  b.material.refractive_index = 2.0;
  Shape c = Shape::glassSphere();
  c.setTransform(Matrix::identity().translate(0, 0, 0.25));
  c.material.refractive_index = 2.5;
  const Ray r{point(0, 0, -4), vec3(0, 0, 1)};
  const Intersections xs{{2.0, &a}, {2.75, &b}, {3.25, &c}, {4.75, &b}, {5.25, &c}, {6.0, &a}};
  const double want[6][2] = {{1.0, 1.5}, {1.5, 2.0}, {2.0, 2.5}, {2.5, 2.5}, {2.5, 1.5}, {1.5, 1.0}};
  for (int i = 0; i < 6; ++i) {
    const PreComputations comps = PreComputations::make(xs[i], r, xs);
    report("shape.zig:551", "n1_n2_" + std::to_string(i), comps.n1 == want[i][0] && comps.n2 == want[i][1]);
  }
  // hit(): shape.zig:464-518
  Shape s = Shape::make(SPHERE);
  Intersections h1{{1, &s}, {2, &s}};
  expectTrue("shape.zig:467", "hit_all_positive", hit(h1) == 0);
  Intersections h2{{-1, &s}, {1, &s}};
  expectTrue("shape.zig:480", "hit_some_negative", hit(h2) == 1);
  Intersections h3{{-2, &s}, {-1, &s}};
  expectTrue("shape.zig:493", "hit_all_negative", hit(h3) == -1);
  Intersections h4{{5, &s}, {7, &s}, {-3, &s}, {2, &s}};
  sortIntersections(h4);
  expectTrue("shape.zig:504", "hit_lowest_nonnegative", hit(h4) >= 0 && h4[hit(h4)].t == 2.0);
}

static void materialKats() {  // material.zig:78-163
  Material m;
  const Shape obj = Shape::make(SPHERE);
  const Tuple position = point(0, 0, 0);
  expectColor("material.zig:83", "eye_between", m.lighting({point(0, 0, -10), {1, 1, 1}}, &obj, position, vec3(0, 0, -1), vec3(0, 0, -1), false), {1.9, 1.9, 1.9});
  expectColor("material.zig:93", "eye_offset_45", m.lighting({point(0, 0, -10), {1, 1, 1}}, &obj, position, vec3(0, RS2, -RS2), vec3(0, 0, -1), false), {1.0, 1.0, 1.0});
  expectColor("material.zig:103", "light_offset_45", m.lighting({point(0, 10, -10), {1, 1, 1}}, &obj, position, vec3(0, 0, -1), vec3(0, 0, -1), false), {0.7364, 0.7364, 0.7364});
  expectColor("material.zig:113", "eye_in_reflection", m.lighting({point(0, 10, -10), {1, 1, 1}}, &obj, position, vec3(0, -RS2, -RS2), vec3(0, 0, -1), false), {1.63639, 1.63639, 1.63639});
  expectColor("material.zig:123", "light_behind", m.lighting({point(0, 0, 10), {1, 1, 1}}, &obj, position, vec3(0, 0, -1), vec3(0, 0, -1), false), {0.1, 0.1, 0.1});
  expectColor("material.zig:133", "in_shadow", m.lighting({point(0, 0, -10), {1, 1, 1}}, &obj, position, vec3(0, 0, -1), vec3(0, 0, -1), true), {0.1, 0.1, 0.1});
  // Lighting with pattern (material.zig:140-163)
  Pattern white, black;
  white.rgb = {1, 1, 1};
  black.rgb = {0, 0, 0};
  Material pm;
  pm.pattern.kind = PAT_STRIPES;
  pm.pattern.a = &white;
  pm.pattern.b = &black;
  pm.ambient = 1.0;
  pm.diffuse = 0.0;
  pm.specular = 0.0;
  const Light l{point(0, 0, -10), {1, 1, 1}};
  expectColor("material.zig:160", "stripes_c1", pm.lighting(l, &obj, point(0.9, 0, 0), vec3(0, 0, -1), vec3(0, 0, -1), false), {1, 1, 1}, 0.0);
  expectColor("material.zig:161", "stripes_c2", pm.lighting(l, &obj, point(1.1, 0, 0), vec3(0, 0, -1), vec3(0, 0, -1), false), {0, 0, 0}, 0.0);
}

static void patternKats() {  // pattern.zig:152-177, checkers.zig:33-54, stripes.zig:37-53, solid.zig:28-39
  Pattern white, black;
  white.rgb = {1, 1, 1};
  black.rgb = {0, 0, 0};
  const Color W{1, 1, 1}, B{0, 0, 0};
  Pattern st;
  st.kind = PAT_STRIPES;
  st.a = &white;
  st.b = &black;
  struct P { double x; Color c; };
  const P sp[] = {{0.0, W}, {0.9, W}, {1.0, B}, {-0.1, B}, {-1.0, B}, {-1.1, W}};
  int i = 0;
  for (const P& p : sp) expectColor("stripes.zig:47", "stripes_" + std::to_string(i++), st.patternAt(point(p.x, 0, 0)), p.c, 0.0);
  Pattern ck;
  ck.kind = PAT_CHECKERS;
  ck.a = &white;
  ck.b = &black;
  struct Q { Tuple p; Color c; };
  const Q cp[] = {{point(0, 0, 0), W},    {point(0.99, 0, 0), W}, {point(1.01, 0, 0), B}, {point(0, 0.99, 0), W},
                  {point(0, 1.01, 0), B}, {point(0, 0, 0.99), W}, {point(0, 0, 1.01), B}};
  i = 0;
  for (const Q& q : cp) expectColor("checkers.zig:43", "checkers_" + std::to_string(i++), ck.patternAt(q.p), q.c, 0.0);
  expectColor("solid.zig:34", "solid", white.patternAt(point(3, 4, 5)), W, 0.0);
  // object + pattern transforms compose (pattern.zig:152-177)
  Pattern tp;
  tp.kind = PAT_TEST;
  Shape s = Shape::make(SPHERE);
  s.setTransform(Matrix::identity().scale(2, 2, 2));
  expectColor("pattern.zig:157", "object_transform", tp.patternAt(s.worldToObject(point(2, 3, 4))), {1, 1.5, 2});
  Pattern tp2 = tp;
  tp2.setTransform(Matrix::identity().scale(2, 2, 2));
  const Shape s0 = Shape::make(SPHERE);
  expectColor("pattern.zig:165", "pattern_transform", tp2.patternAt(s0.worldToObject(point(2, 3, 4))), {1, 1.5, 2});
  Pattern tp3 = tp;
  tp3.setTransform(Matrix::identity().translate(0.5, 1, 1.5));
  expectColor("pattern.zig:174", "both_transforms", tp3.patternAt(s.worldToObject(point(2.5, 3, 3.5))), {0.75, 0.5, 0.25});
  // gradient.zig, rings.zig, blend.zig
  Pattern gr;
  gr.kind = PAT_GRADIENT;
  gr.a = &white;
  gr.b = &black;
  expectColor("gradient.zig:75", "gradient_0.25", gr.patternAt(point(0.25, 0, 0)), {0.75, 0.75, 0.75});
  expectColor("gradient.zig:77", "gradient_0.5", gr.patternAt(point(0.5, 0, 0)), {0.5, 0.5, 0.5});
  Pattern rg;
  rg.kind = PAT_RINGS;
  rg.a = &white;
  rg.b = &black;
  expectColor("rings.zig:45", "rings_x1", rg.patternAt(point(1, 0, 0)), B, 0.0);
  expectColor("rings.zig:47", "rings_diag", rg.patternAt(point(0.708, 0, 0.708)), B, 0.0);
  {  // blend.zig:31-48: stripes and the same stripes rotated by pi/2 about y, averaged
    Pattern st;
    st.kind = PAT_STRIPES;
    st.a = &white;
    st.b = &black;
    Pattern rot = st;
    rot.setTransform(Matrix::identity().rotateY(PI / 2.0));
    Pattern bl;
    bl.kind = PAT_BLEND;
    bl.a = &st;
    bl.b = &rot;
    const Color gray{0.5, 0.5, 0.5};
    expectColor("blend.zig:44", "blend_both_white", bl.patternAt(point(0, 0, 0)), W, 0.0);
    expectColor("blend.zig:45", "blend_gray_1", bl.patternAt(point(0.5, 0, 0.5)), gray, 0.0);
    expectColor("blend.zig:46", "blend_both_black", bl.patternAt(point(-0.5, 0, 0.5)), B, 0.0);
    expectColor("blend.zig:47", "blend_gray_2", bl.patternAt(point(-0.5, 0, -0.5)), gray, 0.0);
  }
}

static Pattern solidPattern(Color c) {
  Pattern p;
  p.kind = PAT_SOLID;
  p.rgb = c;
  return p;
}

static void textureMapKats() {  // texture_map.zig:336-568
  const Tuple nowhere = point(0, 0, 0);
  {  // :350-361 checker pattern in 2D
    const Pattern black = solidPattern({0, 0, 0}), white = solidPattern({1, 1, 1});
    UvPattern ck;
    ck.kind = UV_CHECKERS;
    ck.width = ck.height = 2;
    ck.sub[0] = &black;
    ck.sub[1] = &white;
    const struct { double u, v; Color want; } rows[] = {{0.0, 0.0, {0, 0, 0}}, {0.5, 0.0, {1, 1, 1}}, {0.0, 0.5, {1, 1, 1}},
                                                        {0.5, 0.5, {0, 0, 0}}, {1.0, 1.0, {0, 0, 0}}};
    int k = 0;
    for (const auto& r : rows) expectColor("texture_map.zig:356-360", "uv_checkers_" + std::to_string(k++), ck.uvPatternAt(r.u, r.v, nowhere), r.want);
  }
  auto mapping = [&](const char* where, const std::string& name, TexMapping m, Tuple p, double u, double v) {
    TextureMap tm;
    tm.mapping = m;  // faces default to the uv test pattern: colour = (u, v, 0)
    const Color c = tm.patternAt(p, nowhere);
    expectNear(where, name + "_u", c.r, u);
    expectNear(where, name + "_v", c.g, v);
  };
  {  // :371-381 spherical
    const double r = 1.0 / std::sqrt(2.0);
    const struct { Tuple p; double u, v; } rows[] = {{point(0, 0, -1), 0.0, 0.5}, {point(1, 0, 0), 0.25, 0.5}, {point(0, 0, 1), 0.5, 0.5},
                                                     {point(-1, 0, 0), 0.75, 0.5}, {point(0, 1, 0), 0.5, 1.0}, {point(0, -1, 0), 0.5, 0.0},
                                                     {point(r, r, 0), 0.25, 0.75}};
    int k = 0;
    for (const auto& row : rows) mapping("texture_map.zig:374-380", "spherical_" + std::to_string(k++), TEX_SPHERICAL, row.p, row.u, row.v);
  }
  {  // :383-393 planar
    const struct { Tuple p; double u, v; } rows[] = {{point(0.25, 0, 0.5), 0.25, 0.5}, {point(0.25, 0, -0.25), 0.25, 0.75},
                                                     {point(0.25, 0.5, -0.25), 0.25, 0.75}, {point(1.25, 0, 0.5), 0.25, 0.5},
                                                     {point(0.25, 0, -1.75), 0.25, 0.25}, {point(1, 0, -1), 0.0, 0.0}, {point(0, 0, 0), 0.0, 0.0}};
    int k = 0;
    for (const auto& row : rows) mapping("texture_map.zig:386-392", "planar_" + std::to_string(k++), TEX_PLANAR, row.p, row.u, row.v);
  }
  {  // :395-408 cylindrical
    const struct { Tuple p; double u, v; } rows[] = {
        {point(0, 0, -1), 0.0, 0.0},          {point(0, 0.5, -1), 0.0, 0.5},           {point(0, 1, -1), 0.0, 0.0},
        {point(0.70711, 0.5, -0.70711), 0.125, 0.5}, {point(1, 0.5, 0), 0.25, 0.5},    {point(0.70711, 0.5, 0.70711), 0.375, 0.5},
        {point(0, -0.25, 1), 0.5, 0.75},      {point(-0.70711, 0.5, 0.70711), 0.625, 0.5}, {point(-1, 1.25, 0), 0.75, 0.25},
        {point(-0.70711, 0.5, -0.70711), 0.875, 0.5}};
    int k = 0;
    for (const auto& row : rows) mapping("texture_map.zig:398-407", "cylindrical_" + std::to_string(k++), TEX_CYLINDRICAL, row.p, row.u, row.v);
  }
  const Color red{1, 0, 0}, yellow{1, 1, 0}, brown{1, 0.5, 0}, green{0, 1, 0}, cyan{0, 1, 1}, blue{0, 0, 1}, purple{1, 0, 1}, white{1, 1, 1};
  const Pattern s_red = solidPattern(red), s_yellow = solidPattern(yellow), s_brown = solidPattern(brown), s_green = solidPattern(green),
                s_cyan = solidPattern(cyan), s_blue = solidPattern(blue), s_purple = solidPattern(purple), s_white = solidPattern(white);
  auto align = [](const Pattern* c, const Pattern* ul, const Pattern* ur, const Pattern* bl, const Pattern* br) {
    UvPattern uv;
    uv.kind = UV_ALIGN_CHECK;
    uv.sub[0] = c; uv.sub[1] = ul; uv.sub[2] = ur; uv.sub[3] = bl; uv.sub[4] = br;
    return uv;
  };
  {  // :423-429 layout of the align check pattern
    const UvPattern uv = align(&s_white, &s_red, &s_yellow, &s_green, &s_cyan);
    expectColor("texture_map.zig:424", "align_central", uv.uvPatternAt(0.5, 0.5, nowhere), white);
    expectColor("texture_map.zig:425", "align_ul", uv.uvPatternAt(0.1, 0.9, nowhere), red);
    expectColor("texture_map.zig:426", "align_ur", uv.uvPatternAt(0.9, 0.9, nowhere), yellow);
    expectColor("texture_map.zig:427", "align_bl", uv.uvPatternAt(0.1, 0.1, nowhere), green);
    expectColor("texture_map.zig:428", "align_br", uv.uvPatternAt(0.9, 0.1, nowhere), cyan);
  }
  {  // :431-440 faces, :450-478 per-face uv, :480-546 colours on a mapped cube (face and uv through the test pattern, then colours)
    TextureMap cube;
    cube.mapping = TEX_CUBIC;
    cube.faces[0] = align(&s_cyan, &s_red, &s_yellow, &s_brown, &s_green);      // front
    cube.faces[1] = align(&s_green, &s_purple, &s_cyan, &s_white, &s_blue);     // back
    cube.faces[2] = align(&s_yellow, &s_cyan, &s_red, &s_blue, &s_brown);       // left
    cube.faces[3] = align(&s_red, &s_yellow, &s_purple, &s_green, &s_white);    // right
    cube.faces[4] = align(&s_brown, &s_cyan, &s_purple, &s_red, &s_yellow);     // up
    cube.faces[5] = align(&s_purple, &s_brown, &s_green, &s_blue, &s_white);    // down
    const struct { Tuple p; Color want; } rows[] = {
        {point(-1, 0, 0), yellow},      {point(-1, 0.9, -0.9), cyan},   {point(-1, 0.9, 0.9), red},     {point(-1, -0.9, -0.9), blue},
        {point(-1, -0.9, 0.9), brown},  {point(0, 0, 1), cyan},         {point(-0.9, 0.9, 1), red},     {point(0.9, 0.9, 1), yellow},
        {point(-0.9, -0.9, 1), brown},  {point(0.9, -0.9, 1), green},   {point(1, 0, 0), red},          {point(1, 0.9, 0.9), yellow},
        {point(1, 0.9, -0.9), purple},  {point(1, -0.9, 0.9), green},   {point(1, -0.9, -0.9), white},  {point(0, 0, -1), green},
        {point(0.9, 0.9, -1), purple},  {point(-0.9, 0.9, -1), cyan},   {point(0.9, -0.9, -1), white},  {point(-0.9, -0.9, -1), blue},
        {point(0, 1, 0), brown},        {point(-0.9, 1, -0.9), cyan},   {point(0.9, 1, -0.9), purple},  {point(-0.9, 1, 0.9), red},
        {point(0.9, 1, 0.9), yellow},   {point(0, -1, 0), purple},      {point(-0.9, -1, 0.9), brown},  {point(0.9, -1, 0.9), green},
        {point(-0.9, -1, -0.9), blue},  {point(0.9, -1, -0.9), white}};
    int k = 0;
    for (const auto& row : rows) expectColor("texture_map.zig:514-545", "mapped_cube_" + std::to_string(k++), cube.patternAt(row.p, nowhere), row.want, 0.0);
    TextureMap probe;
    probe.mapping = TEX_CUBIC;  // all faces = test pattern: (u, v)
    const struct { const char* where; Tuple p; double u, v; } uvs[] = {
        {"texture_map.zig:451", point(-0.5, 0.5, 1), 0.25, 0.75},  {"texture_map.zig:452", point(0.5, -0.5, 1), 0.75, 0.25},
        {"texture_map.zig:456", point(0.5, 0.5, -1), 0.25, 0.75},  {"texture_map.zig:457", point(-0.5, -0.5, -1), 0.75, 0.25},
        {"texture_map.zig:461", point(-1, 0.5, -0.5), 0.25, 0.75}, {"texture_map.zig:462", point(-1, -0.5, 0.5), 0.75, 0.25},
        {"texture_map.zig:466", point(1, 0.5, 0.5), 0.25, 0.75},   {"texture_map.zig:467", point(1, -0.5, -0.5), 0.75, 0.25},
        {"texture_map.zig:471", point(-0.5, 1, -0.5), 0.25, 0.75}, {"texture_map.zig:472", point(0.5, 1, 0.5), 0.75, 0.25},
        {"texture_map.zig:476", point(-0.5, -1, 0.5), 0.25, 0.75}, {"texture_map.zig:477", point(0.5, -1, -0.5), 0.75, 0.25}};
    k = 0;
    for (const auto& row : uvs) {
      const Color c = probe.patternAt(row.p, nowhere);
      expectNear(row.where, "cube_uv_" + std::to_string(k) + "_u", c.r, row.u);
      expectNear(row.where, "cube_uv_" + std::to_string(k++) + "_v", c.g, row.v);
    }
  }
  {  // :548-568 canvas-based pattern (P3, max 10: value / 10; rows cycle 0..9 shifted by the row index)
    UvImage im;
    im.width = im.height = 10;
    im.rgb.resize(300);
    for (size_t y = 0; y < 10; ++y)
      for (size_t x = 0; x < 10; ++x)
        for (int c = 0; c < 3; ++c) im.rgb[3 * (y * 10 + x) + c] = static_cast<double>((x + y) % 10) / 10.0;
    UvPattern uv;
    uv.kind = UV_IMAGE;
    uv.image = &im;
    expectColor("texture_map.zig:564", "uv_image_0_0", uv.uvPatternAt(0.0, 0.0, nowhere), {0.9, 0.9, 0.9});
    expectColor("texture_map.zig:565", "uv_image_0.3_0", uv.uvPatternAt(0.3, 0.0, nowhere), {0.2, 0.2, 0.2});
    expectColor("texture_map.zig:566", "uv_image_0.6_0.3", uv.uvPatternAt(0.6, 0.3, nowhere), {0.1, 0.1, 0.1});
    expectColor("texture_map.zig:567", "uv_image_1_1", uv.uvPatternAt(1.0, 1.0, nowhere), {0.9, 0.9, 0.9});
  }
}

static Shape makeCsg(Shape left, Shape right, CsgOp op) {  // shape.zig:253-283 for leaf children at identity
  Shape c = Shape::make(CSG);
  c.csg_op = op;
  c.bmin = point(-1, -1, -1);
  c.bmax = point(1, 1, 1);
  c.children.push_back(std::move(left));
  c.children.push_back(std::move(right));
  return c;
}

static void csgKats() {  // csg.zig:143-258
  struct Row { CsgOp op; bool lhit, inl, inr, result; };
  const Row rows[] = {
      {CSG_UNION, true, true, true, false},         {CSG_UNION, true, true, false, true},
      {CSG_UNION, true, false, true, false},        {CSG_UNION, true, false, false, true},
      {CSG_UNION, false, true, true, false},        {CSG_UNION, false, true, false, false},
      {CSG_UNION, false, false, true, true},        {CSG_UNION, false, false, false, true},
      {CSG_INTERSECTION, true, true, true, true},   {CSG_INTERSECTION, true, true, false, false},
      {CSG_INTERSECTION, true, false, true, true},  {CSG_INTERSECTION, true, false, false, false},
      {CSG_INTERSECTION, false, true, true, true},  {CSG_INTERSECTION, false, true, false, true},
      {CSG_INTERSECTION, false, false, true, false}, {CSG_INTERSECTION, false, false, false, false},
      {CSG_DIFFERENCE, true, true, true, false},    {CSG_DIFFERENCE, true, true, false, true},
      {CSG_DIFFERENCE, true, false, true, false},   {CSG_DIFFERENCE, true, false, false, true},
      {CSG_DIFFERENCE, false, true, true, true},    {CSG_DIFFERENCE, false, true, false, true},
      {CSG_DIFFERENCE, false, false, true, false},  {CSG_DIFFERENCE, false, false, false, false}};
  int k = 0;
  for (const Row& r : rows)
    expectTrue("csg.zig:162-191", "intersection_allowed_" + std::to_string(k++),
               intersectionAllowed(r.op, r.lhit, r.inl, r.inr) == r.result);
  // csg.zig:193-225: filtering {1:s1, 2:s2, 3:s1, 4:s2}
  const struct { CsgOp op; size_t x0, x1; } filt[] = {{CSG_UNION, 0, 3}, {CSG_INTERSECTION, 1, 2}, {CSG_DIFFERENCE, 0, 1}};
  for (const auto& f : filt) {
    Shape c = makeCsg(Shape::make(SPHERE), Shape::make(CUBE), f.op);
    const Shape* s1 = &c.children[0];
    const Shape* s2 = &c.children[1];
    const Intersections xs{{1.0, s1}, {2.0, s2}, {3.0, s1}, {4.0, s2}};
    const Intersections r = csgFilter(c, xs);
    expectTrue("csg.zig:214", "filter_op" + std::to_string(f.op),
               r.size() == 2 && r[0].t == xs[f.x0].t && r[0].object == xs[f.x0].object && r[1].t == xs[f.x1].t &&
                   r[1].object == xs[f.x1].object);
  }
  {  // csg.zig:228-240
    Shape c = makeCsg(Shape::make(SPHERE), Shape::make(CUBE), CSG_UNION);
    expectTrue("csg.zig:223", "ray_misses_csg", c.intersect({point(0, 2, -5), vec3(0, 0, 1)}).empty());
  }
  {  // csg.zig:242-258
    Shape s2 = Shape::make(SPHERE);
    s2.setTransform(Matrix::identity().translate(0, 0, 0.5));
    Shape c = makeCsg(Shape::make(SPHERE), std::move(s2), CSG_UNION);
    c.bmax = point(1, 1, 1.5);  // union of the children's parent-space boxes
    const Intersections xs = c.intersect({point(0, 0, -5), vec3(0, 0, 1)});
    expectTrue("csg.zig:254-258", "ray_hits_csg",
               xs.size() == 2 && xs[0].t == 4.0 && xs[0].object == &c.children[0] && xs[1].t == 6.5 &&
                   xs[1].object == &c.children[1]);
  }
}

static void noiseKats() {  // noise.zig:106-109 (exact comparisons in the reference)
  expectNear("noise.zig:107", "noise_3.14_42_7", perlinNoise(3.14, 42, 7), 0.13691995878400012, 0.0);
  expectNear("noise.zig:108", "noise_-4.20_10_6", perlinNoise(-4.20, 10, 6), 0.14208000000000043, 0.0);
}

static void powKats() {  // Zig std.math.pow as restated by zig_pow (not part of the reference tree: identities only)
  expectNear("std/math/pow.zig", "pow_2_10", zig_pow(2.0, 10.0), 1024.0, 0.0);
  expectNear("std/math/pow.zig", "pow_half_sq", zig_pow(0.5, 2.0), 0.25, 0.0);
  expectNear("std/math/pow.zig", "pow_neg_exp", zig_pow(2.0, -2.0), 0.25, 0.0);
  expectNear("std/math/pow.zig", "pow_neg_base_odd", zig_pow(-2.0, 3.0), -8.0, 0.0);
  expectNear("std/math/pow.zig", "pow_sqrt_case", zig_pow(4.0, 0.5), 2.0, 0.0);
  expectNear("std/math/pow.zig", "pow_y0", zig_pow(123.4, 0.0), 1.0, 0.0);
  expectNear("std/math/pow.zig", "pow_x0", zig_pow(0.0, 3.3), 0.0, 0.0);
  expectNear("std/math/pow.zig", "pow_frac_0.8923", zig_pow(0.8923, 3.3), 0.686572, 1e-6);
  expectNear("std/math/pow.zig", "pow_frac_1.5", zig_pow(1.5, 3.3), 3.811546, 1e-6);
  expectNear("std/math/pow.zig", "pow_frac_37.45", zig_pow(37.45, 3.3), 155736.7160616, 1e-6);
  expectTrue("std/math/pow.zig", "pow_neg_base_frac_is_nan", std::isnan(zig_pow(-8.0, 1.0 / 3.0)));
  // the integer path agrees with libm to the accumulated rounding of its ~2*log2(y) multiplications
  expectNear("std/math/pow.zig", "pow_shininess_200", zig_pow(0.97, 200.0) / std::pow(0.97, 200.0), 1.0, 1e-13);
  expectNear("std/math/pow.zig", "pow_schlick_5", zig_pow(0.3, 5.0) / std::pow(0.3, 5.0), 1.0, 1e-15);
}

static void worldKats() {  // world.zig:293-892
  const World w = World::defaultWorld();
  expectTs("world.zig:293", "world_intersect", w.intersect({point(0, 0, -5), vec3(0, 0, 1)}), {4.0, 4.5, 5.5, 6.0});
  {
    const Shape shape = Shape::make(SPHERE);
    const Intersection i{4.0, &shape};
    const PreComputations c = PreComputations::make(i, {point(0, 0, -5), vec3(0, 0, 1)}, {i});
    expectTuple("world.zig:324", "comps_point", c.point, point(0, 0, -1));
    expectTuple("world.zig:325", "comps_eyev", c.eyev, vec3(0, 0, -1));
    expectTuple("world.zig:326", "comps_normal", c.normal, vec3(0, 0, -1));
    expectTrue("world.zig:327", "comps_outside", !c.inside);
    const Intersection i2{1.0, &shape};
    const PreComputations c2 = PreComputations::make(i2, {point(0, 0, 0), vec3(0, 0, 1)}, {i2});
    expectTuple("world.zig:341", "comps_inside_point", c2.point, point(0, 0, 1));
    expectTuple("world.zig:343", "comps_inside_normal", c2.normal, vec3(0, 0, -1));
    expectTrue("world.zig:344", "comps_inside", c2.inside);
    Shape sh = Shape::make(SPHERE);
    sh.setTransform(Matrix::identity().translate(0, 0, 1));
    const Intersection i3{5.0, &sh};
    const PreComputations c3 = PreComputations::make(i3, {point(0, 0, -5), vec3(0, 0, 1)}, {i3});
    expectTrue("world.zig:358", "over_point", c3.over_point.z < -1e-5 / 2.0 && c3.point.z > c3.over_point.z);
    Shape gs = Shape::glassSphere();
    gs.setTransform(Matrix::identity().translate(0, 0, 1));
    const Intersection i4{5.0, &gs};
    const PreComputations c4 = PreComputations::make(i4, {point(0, 0, -5), vec3(0, 0, 1)}, {i4});
    expectTrue("world.zig:374", "under_point", c4.under_point.z > 1e-5 / 2.0 && c4.point.z < c4.under_point.z);
  }
  {  // Shading, world.zig:379-443
    const Intersection i{4.0, &w.objects[0]};
    const PreComputations c = PreComputations::make(i, {point(0, 0, -5), vec3(0, 0, 1)}, {i});
    expectColor("world.zig:394", "shade_outside", w.shadeHit(c, 3), {0.38066, 0.47583, 0.2855});
    World w2 = World::defaultWorld();
    w2.lights[0] = {point(0, 0.25, 0), {1, 1, 1}};
    const Intersection i2{0.5, &w2.objects[1]};
    const PreComputations c2 = PreComputations::make(i2, {point(0, 0, 0), vec3(0, 0, 1)}, {i2});
    expectColor("world.zig:414", "shade_inside", w2.shadeHit(c2, 3), {0.90498, 0.90498, 0.90498});
    World w3;
    w3.lights.push_back({point(0, 0, -10), {1, 1, 1}});
    w3.objects.push_back(Shape::make(SPHERE));
    Shape s2 = Shape::make(SPHERE);
    s2.setTransform(Matrix::identity().translate(0, 0, 10));
    w3.objects.push_back(s2);
    const Intersection i3{4.0, &w3.objects[1]};
    const PreComputations c3 = PreComputations::make(i3, {point(0, 0, 5), vec3(0, 0, 1)}, {i3});
    expectColor("world.zig:440", "shade_in_shadow", w3.shadeHit(c3, 3), {0.1, 0.1, 0.1});
  }
  {  // Coloring, world.zig:445-491
    expectColor("world.zig:456", "color_miss", w.colorAt({point(0, 0, -5), vec3(0, 1, 0)}, 3), {0, 0, 0}, 0.0);
    expectColor("world.zig:467", "color_hit", w.colorAt({point(0, 0, -5), vec3(0, 0, 1)}, 3), {0.38066, 0.47583, 0.2855});
    World w2 = World::defaultWorld();
    w2.objects[0].material.ambient = 1.0;
    w2.objects[1].material.ambient = 1.0;
    expectColor("world.zig:487", "color_behind_ray", w2.colorAt({point(0, 0, 0.75), vec3(0, 0, -1)}, 3),
                w2.objects[1].material.pattern.rgb);
  }
  {  // isShadowed, world.zig:493-525
    World w2 = World::defaultWorld();
    const Light& l = w2.lights[0];
    expectTrue("world.zig:500", "shadow_collinear_none", !w2.isShadowed(point(0, 10, 0), l));
    expectTrue("world.zig:503", "shadow_object_between", w2.isShadowed(point(10, -10, 10), l));
    expectTrue("world.zig:506", "shadow_behind_light", !w2.isShadowed(point(-20, 20, -20), l));
    expectTrue("world.zig:509", "shadow_behind_point", !w2.isShadowed(point(-2, 2, -2), l));
    const Tuple p = point(0, 0, 0);
    w2.objects[0].casts_shadow = false;
    w2.objects[1].casts_shadow = true;
    expectTrue("world.zig:516", "shadow_optout_outer", w2.isShadowed(p, l));
    w2.objects[0].casts_shadow = true;
    w2.objects[1].casts_shadow = false;
    expectTrue("world.zig:520", "shadow_optout_inner", w2.isShadowed(p, l));
    w2.objects[0].casts_shadow = false;
    w2.objects[1].casts_shadow = false;
    expectTrue("world.zig:524", "shadow_optout_both", !w2.isShadowed(p, l));
  }
  const double S2 = std::sqrt(2.0);
  {  // Reflections, world.zig:527-679
    const Shape plane = Shape::make(PLANE);
    const Intersection i{S2, &plane};
    const PreComputations c = PreComputations::make(i, {point(0, 1, -1), vec3(0, -RS2, RS2)}, {i});
    expectTuple("world.zig:542", "reflectv", c.reflectv, vec3(0, RS2, RS2));
    World w1 = World::defaultWorld();
    w1.objects[1].material.ambient = 1.0;  // (the reference omits this; result is black either way: reflective == 0)
    const Intersection i1{1.0, &w1.objects[1]};
    const PreComputations c1 = PreComputations::make(i1, {point(0, 1, 0), vec3(0, 0, 1)}, {i1});
    expectColor("world.zig:560", "reflected_nonreflective", w1.reflectedColor(c1, 3), {0, 0, 0}, 0.0);

    World w2 = World::defaultWorld();
    Shape shape = Shape::make(PLANE);
    shape.material.reflective = 0.5;
    shape.setTransform(Matrix::identity().translate(0, -1, 0));
    w2.objects.push_back(shape);
    const Ray r{point(0, 0, -3), vec3(0, -RS2, RS2)};
    const Intersection i2{S2, &w2.objects[2]};
    const PreComputations c2 = PreComputations::make(i2, r, {i2});
    expectColor("world.zig:584", "reflected_color", w2.reflectedColor(c2, 3), {0.19033, 0.23791, 0.14275});
    expectColor("world.zig:607", "shade_with_reflection", w2.shadeHit(c2, 3), {0.87676, 0.92434, 0.82917});
    expectColor("world.zig:676", "reflected_at_depth_0", w2.reflectedColor(c2, 0), {0, 0, 0}, 0.0);

    World w3;  // mutually reflective planes terminate, world.zig:633-654
    w3.lights.push_back({point(0, 0, 0), {1, 1, 1}});
    Shape lower = Shape::make(PLANE);
    lower.material.reflective = 1.0;
    lower.setTransform(Matrix::identity().translate(0, -1, 0));
    Shape upper = Shape::make(PLANE);
    upper.material.reflective = 1.0;
    upper.setTransform(Matrix::identity().translate(0, 1, 0));
    w3.objects.push_back(lower);
    w3.objects.push_back(upper);
    const Color cc = w3.colorAt({point(0, 0, 0), vec3(0, 1, 0)}, 3);
    expectTrue("world.zig:653", "parallel_mirrors_terminate", std::isfinite(cc.r));
  }
  {  // Refraction base cases / TIR, world.zig:681-749
    World w1 = World::defaultWorld();
    const Shape& shape = w1.objects[0];
    const Intersections xs{{4.0, &shape}, {6.0, &shape}};
    const PreComputations c = PreComputations::make(xs[0], {point(0, 0, -5), vec3(0, 0, 1)}, xs);
    expectColor("world.zig:699", "refracted_opaque", w1.refractedColor(c, 3), {0, 0, 0}, 0.0);
    World w2 = World::defaultWorld();
    w2.objects[0].material.transparency = 1.0;
    w2.objects[0].material.refractive_index = 1.5;
    const Intersections xs2{{4.0, &w2.objects[0]}, {6.0, &w2.objects[0]}};
    const PreComputations c2 = PreComputations::make(xs2[0], {point(0, 0, -5), vec3(0, 0, 1)}, xs2);
    expectColor("world.zig:721", "refracted_depth_0", w2.refractedColor(c2, 0), {0, 0, 0}, 0.0);
    const Intersections xs3{{-RS2, &w2.objects[0]}, {RS2, &w2.objects[0]}};
    const PreComputations c3 = PreComputations::make(xs3[1], {point(0, 0, RS2), vec3(0, 1, 0)}, xs3);
    expectColor("world.zig:725", "total_internal_reflection", w2.refractedColor(c3, 5), {0, 0, 0}, 0.0);
  }
  {  // Recursive refraction, world.zig:751-807
    World w1 = World::defaultWorld();
    w1.objects[0].material.ambient = 1.0;
    w1.objects[0].material.pattern.kind = PAT_TEST;
    w1.objects[1].material.transparency = 1.0;
    w1.objects[1].material.refractive_index = 1.5;
    const Shape* a = &w1.objects[0];
    const Shape* b = &w1.objects[1];
    const Intersections xs{{-0.9899, a}, {-0.4899, b}, {0.4899, b}, {0.9899, a}};
    const PreComputations c = PreComputations::make(xs[2], {point(0, 0, 0.1), vec3(0, 1, 0)}, xs);
    expectColor("world.zig:776", "refracted_color", w1.refractedColor(c, 5), {0, 0.99887, 0.04721});

    World w2 = World::defaultWorld();
    Shape floor = Shape::make(PLANE);
    floor.setTransform(Matrix::identity().translate(0, -1, 0));
    floor.material.transparency = 0.5;
    floor.material.refractive_index = 1.5;
    Shape ball = Shape::make(SPHERE);
    ball.setTransform(Matrix::identity().translate(0, -3.5, -0.5));
    ball.material.pattern.rgb = {1, 0, 0};
    ball.material.ambient = 0.5;
    w2.objects.push_back(floor);
    w2.objects.push_back(ball);
    const Intersections xs2{{S2, &w2.objects[2]}};
    const PreComputations c2 = PreComputations::make(xs2[0], {point(0, 0, -3), vec3(0, -RS2, RS2)}, xs2);
    expectColor("world.zig:805", "shade_transparent", w2.shadeHit(c2, 5), {0.93642, 0.68642, 0.68642});
  }
  {  // Schlick, world.zig:809-892
    const Shape gs = Shape::glassSphere();
    const Intersections xs{{-RS2, &gs}, {RS2, &gs}};
    expectNear("world.zig:825", "schlick_tir", PreComputations::make(xs[1], {point(0, 0, RS2), vec3(0, 1, 0)}, xs).schlick(), 1.0, 0.0);
    const Intersections xs2{{-1.0, &gs}, {1.0, &gs}};
    expectNear("world.zig:841", "schlick_perpendicular", PreComputations::make(xs2[1], {point(0, 0, 0), vec3(0, 1, 0)}, xs2).schlick(), 0.04);
    const Intersections xs3{{1.8589, &gs}};
    expectNear("world.zig:856", "schlick_grazing", PreComputations::make(xs3[0], {point(0, 0.99, -2), vec3(0, 0, 1)}, xs3).schlick(), 0.48873);
    World w2 = World::defaultWorld();
    Shape floor = Shape::make(PLANE);
    floor.setTransform(Matrix::identity().translate(0, -1, 0));
    floor.material.reflective = 0.5;
    floor.material.transparency = 0.5;
    floor.material.refractive_index = 1.5;
    Shape ball = Shape::make(SPHERE);
    ball.setTransform(Matrix::identity().translate(0, -3.5, -0.5));
    ball.material.pattern.rgb = {1, 0, 0};
    ball.material.ambient = 0.5;
    w2.objects.push_back(floor);
    w2.objects.push_back(ball);
    const Intersections xs4{{S2, &w2.objects[2]}};
    const PreComputations c4 = PreComputations::make(xs4[0], {point(0, 0, -3), vec3(0, -RS2, RS2)}, xs4);
    expectColor("world.zig:890", "shade_schlick", w2.shadeHit(c4, 5), {0.93391, 0.69643, 0.69243});
  }
}

static void cameraKats() {  // camera.zig:129-187
  expectNear("camera.zig:133", "pixel_size_landscape", Camera::make(200, 125, PI / 2).pixel_size, 0.01);
  expectNear("camera.zig:139", "pixel_size_portrait", Camera::make(125, 200, PI / 2).pixel_size, 0.01);
  Camera c = Camera::make(201, 101, PI / 2);
  Ray r = c.rayForPixel(100, 50);
  expectTuple("camera.zig:148", "ray_center_origin", r.origin, point(0, 0, 0));
  expectTuple("camera.zig:149", "ray_center_dir", r.direction, vec3(0, 0, -1));
  r = c.rayForPixel(0, 0);
  expectTuple("camera.zig:156", "ray_corner_dir", r.direction, vec3(0.66519, 0.33259, -0.66851));
  c.setTransform(Matrix::identity().translate(0, -2, 5).rotateY(PI / 4));
  r = c.rayForPixel(100, 50);
  expectTuple("camera.zig:165", "ray_transformed_origin", r.origin, point(0, 2, -5));
  expectTuple("camera.zig:166", "ray_transformed_dir", r.direction, vec3(RS2, 0, -RS2));
  // Rendering, camera.zig:171-187
  const World w = World::defaultWorld();
  Camera rc = Camera::make(11, 11, PI / 2);
  rc.setTransform(Matrix::viewTransform(point(0, 0, -5), point(0, 0, 0), vec3(0, 1, 0)));
  expectColor("camera.zig:186", "render_center_pixel", w.colorAt(rc.rayForPixel(5, 5), 5), {0.38066, 0.47583, 0.2855});
}


// ------------------------------------------------------------------------------------------
// The oracle's OWN scene build (rtc_oracle_scene.hpp: the build-time halves of bounding_box.zig, group.zig, shape.zig,
// parsing/obj.zig, parsing/scene.zig) against the reference's tests of those functions.  The product loader passes the
// same vectors in tests/cpp/host_kat_main.cpp; tests/test_oracle_scene_cpu.py holds the two builds bit-equal on whole
// scenes - these cases are what makes that comparison more than two restatements by one author agreeing.
static bool bitEq(Tuple a, Tuple b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }
static bool bitEq(const Matrix& a, const Matrix& b) {
  for (int i = 0; i < 16; ++i)
    if (a.d[i / 4][i % 4] != b.d[i / 4][i % 4]) return false;
  return true;
}

static void sceneBoxKats() {  // bounding_box.zig:183-252, 362-423
  namespace sc = orc::scene;
  {
    sc::Box box = sc::newBox();  // bounding_box.zig:183-190
    sc::boxAdd(box, point(-5, 2, 0));
    sc::boxAdd(box, point(7, 0, -3));
    expectTrue("bounding_box.zig:183", "scene_add_points", bitEq(box.min, point(-5, 0, -3)) && bitEq(box.max, point(7, 2, 0)));
  }
  {
    sc::Box b = sc::newBox();  // bounding_box.zig:192-236
    b.min = point(5, -2, 0);
    b.max = point(11, 4, 7);
    const Tuple in[] = {point(5, -2, 0), point(11, 4, 7), point(8, 1, 3)};
    const Tuple out[] = {point(3, 0, 3), point(8, -4, 3), point(8, 1, -1), point(13, 1, 3), point(8, 5, 3), point(8, 1, 8)};
    bool ok = true;
    for (const Tuple& p : in) ok = ok && sc::boxContainsPoint(b, p);
    for (const Tuple& p : out) ok = ok && !sc::boxContainsPoint(b, p);
    expectTrue("bounding_box.zig:193", "scene_contains_point", ok);
    auto cb = [&](Tuple mn, Tuple mx) {  // bounding_box.zig:238-252
      sc::Box o = sc::newBox();
      o.min = mn;
      o.max = mx;
      return sc::boxContainsBox(b, o);
    };
    expectTrue("bounding_box.zig:242", "scene_contains_box",
               cb(point(5, -2, 0), point(11, 4, 7)) && cb(point(6, -1, 1), point(10, 3, 6)) &&
                   !cb(point(4, -3, -1), point(10, 3, 6)) && !cb(point(6, -1, 1), point(12, 5, 8)));
  }
  {
    sc::Box unit = sc::newBox();  // bounding_box.zig:254-266
    unit.min = point(-1, -1, -1);
    unit.max = point(1, 1, 1);
    const sc::Box tb = sc::boxTransform(unit, Matrix::identity().rotateY(PI / 4).rotateX(PI / 4));
    expectTuple("bounding_box.zig:262", "scene_transform_min", tb.min, point(-1.41421, -1.7071, -1.7071), 1e-4);
    expectTuple("bounding_box.zig:265", "scene_transform_max", tb.max, point(1.41421, 1.7071, 1.7071), 1e-4);
  }
  struct S { Tuple mn, mx, lmax, rmin; int line; const char* name; };
  const S splits[] = {{point(-1, -4, -5), point(9, 6, 5), point(4, 6, 5), point(4, -4, -5), 365, "scene_split_cube"},
                      {point(-1, -2, -3), point(9, 5.5, 3), point(4, 5.5, 3), point(4, -2, -3), 380, "scene_split_x_wide"},
                      {point(-1, -2, -3), point(5, 8, 3), point(5, 3, 3), point(-1, 3, -3), 395, "scene_split_y_wide"},
                      {point(-1, -2, -3), point(5, 3, 7), point(5, 3, 2), point(-1, -2, 2), 410, "scene_split_z_wide"}};
  for (const S& c : splits) {
    sc::Box b = sc::newBox(), l, r;
    b.min = c.mn;
    b.max = c.mx;
    sc::boxSplit(b, l, r);
    char where[48];
    std::snprintf(where, sizeof where, "bounding_box.zig:%d", c.line);
    expectTrue(where, c.name, bitEq(l.min, c.mn) && approxEqual(l.max, c.lmax) && approxEqual(r.min, c.rmin) && bitEq(r.max, c.mx));
  }
  {  // per-kind bounds: cylinder.zig:334-344, cone.zig:241-260, plane.zig:109-116, triangle.zig:198-208, shape.zig:638-654
    Shape cyl = Shape::make(CYLINDER);
    cyl.ymin = -5;
    cyl.ymax = 3;
    expectTrue("cylinder.zig:342", "scene_cylinder_bounds",
               bitEq(sc::shapeBounds(cyl).min, point(-1, -5, -1)) && bitEq(sc::shapeBounds(cyl).max, point(1, 3, 1)));
    Shape cone = Shape::make(CONE);
    cone.ymin = -5;
    cone.ymax = 3;
    expectTrue("cone.zig:249", "scene_cone_bounds",
               bitEq(sc::shapeBounds(cone).min, point(-5, -5, -5)) && bitEq(sc::shapeBounds(cone).max, point(5, 3, 5)));
    const Shape pl = Shape::make(PLANE);
    expectTrue("plane.zig:109", "scene_plane_bounds",
               sc::shapeBounds(pl).min.x == -INF && sc::shapeBounds(pl).min.y == 0 && sc::shapeBounds(pl).max.z == INF);
    const Shape tri = Shape::triangle(point(-3, 7, 2), point(6, 2, -4), point(2, -1, -1));
    expectTrue("triangle.zig:198", "scene_triangle_bounds",
               bitEq(sc::shapeBounds(tri).min, point(-3, -1, -4)) && bitEq(sc::shapeBounds(tri).max, point(6, 7, 2)));
    Shape s = Shape::make(SPHERE);
    sc::setTransform(s, Matrix::identity().scale(0.5, 2, 4).translate(1, -3, 5));
    expectTuple("shape.zig:652", "scene_parent_space_min", sc::parentSpaceBounds(s).min, point(0.5, -5, 1));
    expectTuple("shape.zig:653", "scene_parent_space_max", sc::parentSpaceBounds(s).max, point(1.5, -1, 9));
  }
}

static void sceneGroupKats() {  // group.zig:219-381
  namespace sc = orc::scene;
  {  // group.zig:219-241: a group's box contains its children
    Shape s = Shape::make(SPHERE);
    sc::setTransform(s, Matrix::identity().scale(2, 2, 2).translate(2, 5, -3));
    Shape c = Shape::make(CYLINDER);
    c.ymin = -2;
    c.ymax = 2;
    sc::setTransform(c, Matrix::identity().scale(0.5, 1, 0.5).translate(-4, -1, 4));
    Shape g = sc::newGroup();
    sc::addChild(g, s);
    sc::addChild(g, c);
    expectTuple("group.zig:240", "scene_group_bounds_min", sc::shapeBounds(g).min, point(-4.5, -3, -5));
    expectTuple("group.zig:241", "scene_group_bounds_max", sc::shapeBounds(g).max, point(4, 7, 4.5));
  }
  {  // group.zig:246-273 partitionChildren
    Shape s1 = Shape::make(SPHERE);
    sc::setTransform(s1, Matrix::identity().translate(-2, 0, 0));
    Shape s2 = Shape::make(SPHERE);
    sc::setTransform(s2, Matrix::identity().translate(2, 0, 0));
    Shape s3 = Shape::make(SPHERE);
    const size_t i1 = s1.id, i2 = s2.id, i3 = s3.id;
    Shape g = sc::newGroup();
    sc::addChild(g, s1);
    sc::addChild(g, s2);
    sc::addChild(g, s3);
    std::vector<Shape> left, right;
    sc::partitionChildren(g, left, right);
    expectTrue("group.zig:270", "scene_partition",
               g.children.size() == 1 && g.children[0].id == i3 && left.size() == 1 && left[0].id == i1 && right.size() == 1 &&
                   right[0].id == i2);
  }
  {  // group.zig:275-292 makeSubgroup
    Shape g = sc::newGroup();
    std::vector<Shape> kids{Shape::make(SPHERE), Shape::make(SPHERE)};
    sc::makeSubgroup(g, kids);
    expectTrue("group.zig:291", "scene_make_subgroup",
               g.children.size() == 1 && g.children[0].kind == GROUP && g.children[0].children.size() == 2);
  }
  {  // group.zig:294-330 divide(1)
    Shape s1 = Shape::make(SPHERE);
    sc::setTransform(s1, Matrix::identity().translate(-2, -2, 0));
    Shape s2 = Shape::make(SPHERE);
    sc::setTransform(s2, Matrix::identity().translate(-2, 2, 0));
    Shape s3 = Shape::make(SPHERE);
    sc::setTransform(s3, Matrix::identity().scale(4, 4, 4));
    const size_t i1 = s1.id, i2 = s2.id, i3 = s3.id;
    Shape g = sc::newGroup();
    sc::addChild(g, s1);
    sc::addChild(g, s2);
    sc::addChild(g, s3);
    sc::divide(g, 1);
    bool ok = g.children.size() == 2 && g.children[0].id == i3 && g.children[1].kind == GROUP;
    if (ok) {
      const Shape& sub = g.children[1];
      ok = sub.children.size() == 2 && sub.children[0].kind == GROUP && sub.children[0].children.size() == 1 &&
           sub.children[0].children[0].id == i1 && sub.children[1].children.size() == 1 && sub.children[1].children[0].id == i2;
    }
    expectTrue("group.zig:313", "scene_divide_1", ok);
  }
  {  // group.zig:332-381 divide(3) with too few children at the top
    Shape s1 = Shape::make(SPHERE);
    sc::setTransform(s1, Matrix::identity().translate(-2, 0, 0));
    Shape s2 = Shape::make(SPHERE);
    sc::setTransform(s2, Matrix::identity().translate(2, 1, 0));
    Shape s3 = Shape::make(SPHERE);
    sc::setTransform(s3, Matrix::identity().translate(2, -1, 0));
    Shape s4 = Shape::make(SPHERE);
    const size_t i1 = s1.id, i2 = s2.id, i3 = s3.id, i4 = s4.id;
    Shape sub = sc::newGroup();
    sc::addChild(sub, s1);
    sc::addChild(sub, s2);
    sc::addChild(sub, s3);
    Shape g = sc::newGroup();
    sc::addChild(g, sub);
    sc::addChild(g, s4);
    sc::divide(g, 3);
    bool ok = g.children.size() == 2 && g.children[0].kind == GROUP && g.children[0].children.size() == 2 && g.children[1].id == i4;
    if (ok) {
      const Shape& a = g.children[0].children[0];
      const Shape& b = g.children[0].children[1];
      ok = a.children.size() == 1 && a.children[0].id == i1 && b.children.size() == 2 && b.children[0].id == i2 &&
           b.children[1].id == i3;
    }
    expectTrue("group.zig:363", "scene_divide_3", ok);
  }
  {  // group.zig:201-217, shape.zig:286-296: a group's transform goes to its leaves and its box is re-boxed
    Shape s = Shape::make(SPHERE);
    sc::setTransform(s, Matrix::identity().translate(5, 0, 0));
    Shape g = sc::newGroup();
    sc::addChild(g, s);
    sc::setTransform(g, Matrix::identity().scale(2, 2, 2));
    expectTrue("shape.zig:288", "scene_group_pushes_transform",
               bitEq(g.transform, Matrix::identity()) &&
                   bitEq(g.children[0].transform, Matrix::identity().scale(2, 2, 2).mul(Matrix::identity().translate(5, 0, 0))));
    expectTuple("shape.zig:294", "scene_group_rebox_min", g.bmin, point(8, -2, -2));
    expectTuple("shape.zig:294", "scene_group_rebox_max", g.bmax, point(12, 2, 2));
  }
}

static void sceneObjKats() {  // parsing/obj.zig:288-544
  namespace sc = orc::scene;
  {
    sc::ObjParser p;  // obj.zig:288-307
    p.loadObj("There was a young lady named Bright\nwho traveled much faster than light.\nShe set out one day\nin a relative way,\nand came back the previous night.", {}, false);
    expectTrue("obj.zig:306", "scene_ignored_lines", p.lines_ignored == 5);
  }
  {
    sc::ObjParser p;  // obj.zig:309-340
    p.loadObj("v -1 1 0\nv -1.0000 0.5000 0.0000\nv 1 0 0\nv 1 1 0", {}, false);
    expectTrue("obj.zig:326", "scene_vertices",
               p.lines_ignored == 0 && p.vertices().size() == 4 && bitEq(p.vertices()[0], point(-1, 1, 0)) &&
                   bitEq(p.vertices()[1], point(-1, 0.5, 0)) && bitEq(p.vertices()[2], point(1, 0, 0)) &&
                   bitEq(p.vertices()[3], point(1, 1, 0)));
  }
  {
    sc::ObjParser p;  // obj.zig:342-374
    p.loadObj("v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\nf 1 2 3\nf 1 3 4", {}, false);
    const auto& c = p.defaultGroup().children;
    expectTrue("obj.zig:361", "scene_faces",
               p.lines_ignored == 0 && c.size() == 2 && bitEq(c[0].p1, p.vertices()[0]) && bitEq(c[0].p2, p.vertices()[1]) &&
                   bitEq(c[0].p3, p.vertices()[2]) && bitEq(c[1].p1, p.vertices()[0]) && bitEq(c[1].p2, p.vertices()[2]) &&
                   bitEq(c[1].p3, p.vertices()[3]));
  }
  {
    sc::ObjParser p;  // obj.zig:376-411
    p.loadObj("v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\nv 0 2 0\nf 1 2 3 4 5", {}, false);
    const auto& c = p.defaultGroup().children;
    expectTrue("obj.zig:395", "scene_fan_triangulation",
               p.lines_ignored == 0 && c.size() == 3 && bitEq(c[0].p1, p.vertices()[0]) && bitEq(c[0].p2, p.vertices()[1]) &&
                   bitEq(c[0].p3, p.vertices()[2]) && bitEq(c[1].p1, p.vertices()[0]) && bitEq(c[1].p2, p.vertices()[2]) &&
                   bitEq(c[1].p3, p.vertices()[3]) && bitEq(c[2].p1, p.vertices()[0]) && bitEq(c[2].p2, p.vertices()[3]) &&
                   bitEq(c[2].p3, p.vertices()[4]));
  }
  {
    sc::ObjParser p;  // obj.zig:413-448 (triangles in groups), :450-480 (converting an OBJ file to a group)
    p.loadObj("v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\ng FirstGroup\nf 1 2 3\ng SecondGroup\nf 1 3 4", {}, false);
    const auto& c = p.defaultGroup().children;  // named groups are children of the default group, in file order
    bool ok = p.lines_ignored == 0 && c.size() == 2 && c[0].kind == GROUP && c[1].kind == GROUP && c[0].children.size() == 1 &&
              c[1].children.size() == 1;
    if (ok)
      ok = bitEq(c[0].children[0].p1, p.vertices()[0]) && bitEq(c[0].children[0].p2, p.vertices()[1]) &&
           bitEq(c[0].children[0].p3, p.vertices()[2]) && bitEq(c[1].children[0].p1, p.vertices()[0]) &&
           bitEq(c[1].children[0].p2, p.vertices()[2]) && bitEq(c[1].children[0].p3, p.vertices()[3]);
    expectTrue("obj.zig:434", "scene_named_groups", ok);
    const size_t g1 = ok ? c[0].id : 0, g2 = ok ? c[1].id : 0;
    const Shape g = p.toGroup();
    expectTrue("obj.zig:471", "scene_obj_to_group",
               ok && g.kind == GROUP && g.children.size() == 2 && g.children[0].id == g1 && g.children[1].id == g2);
  }
  {
    sc::ObjParser p;  // obj.zig:482-509
    p.loadObj("vn 0 0 1\nvn 0.707 0 -0.707\nvn 1 2 3", {}, false);
    expectTrue("obj.zig:498", "scene_normals",
               p.lines_ignored == 0 && p.normals().size() == 3 && bitEq(p.normals()[0], vec3(0, 0, 1)) &&
                   bitEq(p.normals()[1], vec3(0.707, 0, -0.707)) && bitEq(p.normals()[2], vec3(1, 2, 3)));
  }
  {
    sc::ObjParser p;  // obj.zig:511-544
    p.loadObj("v 0 1 0\nv -1 0 0\nv 1 0 0\nvn -1 0 0\nvn 1 0 0\nvn 0 1 0\nf 1//3 2//1 3//2\nf 1/0/3 2/102/1 3/14/2", {}, false);
    const auto& c = p.defaultGroup().children;
    bool ok = p.lines_ignored == 0 && c.size() == 2 && c[0].kind == SMOOTH_TRIANGLE && c[1].kind == SMOOTH_TRIANGLE;
    if (ok)
      ok = bitEq(c[0].p1, p.vertices()[0]) && bitEq(c[0].p2, p.vertices()[1]) && bitEq(c[0].p3, p.vertices()[2]) &&
           bitEq(c[0].n1, p.normals()[2]) && bitEq(c[0].n2, p.normals()[0]) && bitEq(c[0].n3, p.normals()[1]) &&
           bitEq(c[1].p1, c[0].p1) && bitEq(c[1].p2, c[0].p2) && bitEq(c[1].p3, c[0].p3) && bitEq(c[1].n1, c[0].n1) &&
           bitEq(c[1].n2, c[0].n2) && bitEq(c[1].n3, c[0].n3);
    expectTrue("obj.zig:532", "scene_faces_with_normals", ok);
  }
  {  // obj.zig:197-260: normalisation - offset = the box's centre, scale = half its longest extent; vertices (p - offset) / scale
    sc::ObjParser p;
    p.loadObj("v 0 0 0\nv 4 2 1\nf 1 2 2", {}, true);
    expectTrue("obj.zig:257", "scene_normalize",
               p.scale() == 2.0 && p.offset().x == 2.0 && p.offset().y == 1.0 && p.offset().z == 0.5 && p.vertices()[0].x == -1.0 &&
                   p.vertices()[1].x == 1.0 && p.vertices()[0].w == 0.5);
  }
}

static void sceneParseKats() {  // parsing/scene.zig:664-774
  namespace sc = orc::scene;
  const char* scene_json = R"({
     "camera": { "width": 1280, "height": 1000, "field-of-view": 0.785,
                 "from": [ -6, 6, -10 ], "to": [ 6, 0, 6 ], "up": [ -0.45, 1, 0 ] },
     "objects": [ { "type": { "sphere": {} },
                    "transform": [ { "translate": [1.0, 2.0, 3.0] }, { "scale": [0.5, 0.5, 0.5] } ],
                    "material": { "pattern": { "type": { "stripes": [ { "type": { "solid": [1.0, 1.0, 1.0] } },
                                                                       { "type": { "solid": [0.0, 0.0, 0.0] } } ] },
                                               "transform": [ { "scale": [0.1, 0.1, 0.1] } ] },
                                  "reflective": 0.5 } } ],
     "lights": [ { "point-light": { "position": [-10.0, 10.0, -10.0], "intensity": [1.0, 1.0, 1.0] } } ] })";
  sc::Files files;
  files.load = [](const std::string&) -> std::string { throw std::runtime_error("no files"); };
  files.image = [](const std::string&) -> UvImage { throw std::runtime_error("no images"); };
  const auto built = sc::buildScene(scene_json, files);
  Camera expected = Camera::make(1280, 1000, 0.785);
  expected.setTransform(Matrix::viewTransform(point(-6, 6, -10), point(6, 0, 6), vec3(-0.45, 1, 0)));
  const Camera& cam = built->camera;
  expectTrue("scene.zig:724", "scene_camera",
             cam.hsize == 1280 && cam.vsize == 1000 && cam.pixel_size == expected.pixel_size && cam.half_width == expected.half_width &&
                 cam.half_height == expected.half_height && bitEq(cam.transform, expected.transform) && bitEq(cam.inverse, expected.inverse));
  expectTrue("scene.zig:737", "scene_one_object", built->world.objects.size() == 1);
  const Shape& o = built->world.objects.at(0);
  expectTrue("scene.zig:726", "scene_object_transform",
             o.kind == SPHERE && bitEq(o.transform, Matrix::identity().translate(1, 2, 3).scale(0.5, 0.5, 0.5)));
  Pattern ep;
  ep.setTransform(Matrix::identity().scale(0.1, 0.1, 0.1));
  const Pattern& op = o.material.pattern;
  expectTrue("scene.zig:740", "scene_pattern_transform", op.kind == PAT_STRIPES && bitEq(op.transform, ep.transform));
  expectTrue("scene.zig:744", "scene_pattern_inverse", bitEq(op.inverse, ep.inverse));
  expectTrue("scene.zig:748", "scene_material_ambient", o.material.ambient == 0.1);
  expectTrue("scene.zig:752", "scene_material_reflective", o.material.reflective == 0.5);
  expectTrue("scene.zig:756", "scene_stripes_a", op.a && op.a->kind == PAT_SOLID && op.a->rgb.r == 1.0 && op.a->rgb.g == 1.0 && op.a->rgb.b == 1.0);
  expectTrue("scene.zig:760", "scene_stripes_b", op.b && op.b->kind == PAT_SOLID && op.b->rgb.r == 0.0 && op.b->rgb.g == 0.0 && op.b->rgb.b == 0.0);
  expectTrue("scene.zig:770", "scene_light",
             built->world.lights.size() == 1 && bitEq(built->world.lights[0].position, point(-10, 10, -10)) &&
                 built->world.lights[0].intensity.r == 1.0 && built->world.lights[0].intensity.g == 1.0 &&
                 built->world.lights[0].intensity.b == 1.0);
  // scene.zig:440-546: from-definition inside a group; the group's transform and material reach the leaf, the
  // definition's own and the use site's are applied inside out (scene.zig:455-546, shape.zig:286-296)
  const char* defs = R"({"camera":{"width":2,"height":2,"field-of-view":1,"from":[0,0,-5],"to":[0,0,0],"up":[0,1,0]},"lights":[],
    "shape-definitions":[{"name":"leg","value":{"type":{"sphere":{}},"transform":[{"scale":[2,2,2]}],"material":{"ambient":0.3}}}],
    "objects":[{"type":{"group":[{"type":{"from-definition":"leg"},"transform":[{"translate":[1,0,0]}],"material":{"diffuse":0.4}}]},
                "transform":[{"translate":[0,5,0]}],"material":{"specular":0.25}}]})";
  const auto di = sc::buildScene(defs, files);
  bool ok = di->world.objects.size() == 1 && di->world.objects[0].kind == GROUP && di->world.objects[0].children.size() == 1;
  if (ok) {
    const Shape& leaf = di->world.objects[0].children[0];
    const Matrix want = Matrix::identity().translate(0, 5, 0).mul(
        Matrix::identity().translate(1, 0, 0).mul(Matrix::identity().scale(2, 2, 2).mul(Matrix::identity())));
    ok = bitEq(leaf.transform, want) && leaf.material.ambient == 0.3 && leaf.material.diffuse == 0.4 && leaf.material.specular == 0.25;
  }
  expectTrue("scene.zig:455", "scene_from_definition_in_group", ok);
  auto fails = [&](const std::string& js) {
    try {
      (void)sc::buildScene(js, files);
    } catch (const std::exception&) {
      return true;
    }
    return false;
  };
  const std::string cam_json = R"("camera":{"width":2,"height":2,"field-of-view":1,"from":[0,0,-5],"to":[0,0,0],"up":[0,1,0]},"lights":[])";
  expectTrue("scene.zig:493", "scene_unknown_definition", fails("{" + cam_json + R"(,"objects":[{"type":{"from-definition":"nope"}}]})"));
  expectTrue("scene.zig:203", "scene_missing_field", fails(R"({"lights":[],"objects":[]})"));
}

// The reference tests that had no vector of their own until round 5's audit (tests/test_reference_test_index.py holds every
// `test "..."` block of the reference's files to at least one KAT inside its line range).  Most are one-liners - a constructor,
// a default, a bounds() - which is why they were passed over; they are the reference's all the same.
static void auditKats() {
  namespace sc = orc::scene;
  auto nearM = [](const Matrix& a, const Matrix& b) {  // Matrix.approxEqual (matrix.zig: every element within the tolerance)
    for (int i = 0; i < 16; ++i)
      if (!(std::fabs(a.d[i / 4][i % 4] - b.d[i / 4][i % 4]) < 1e-5)) return false;
    return true;
  };
  {  // csg.zig:143-154 CSG is created with an operation and two shapes (the scene builder's constructor)
    const Shape s1 = Shape::make(SPHERE), s2 = Shape::make(CUBE);
    const Shape c = sc::newCsg(s1, s2, CSG_UNION);
    expectTrue("csg.zig:143", "scene_csg_creation", c.kind == CSG && c.csg_op == CSG_UNION && c.children.size() == 2 &&
                                                        c.children[0].id == s1.id && c.children[0].kind == SPHERE &&
                                                        c.children[1].id == s2.id && c.children[1].kind == CUBE);
  }
  {  // ray.zig:37-41 Ray creation
    const Ray r{point(1, 2, 3), vec3(4, 5, 6)};
    expectTrue("ray.zig:37", "ray_creation", bitEq(r.origin, point(1, 2, 3)) && bitEq(r.direction, vec3(4, 5, 6)));
  }
  {  // shape.zig:441-448 Id uniqueness; :450-462 Creation
    const Shape s1 = Shape::make(SPHERE), s2 = Shape::make(SPHERE), s3 = Shape::make(TEST_SHAPE);
    expectTrue("shape.zig:441", "id_uniqueness", s1.id != s2.id && s2.id != s3.id && s3.id != s1.id);
    Shape s = Shape::make(TEST_SHAPE);
    expectTrue("shape.zig:452", "default_transform", bitEq(s.transform, Matrix::identity()));
    s.setTransform(Matrix::identity().translate(2, 3, 4));
    expectTrue("shape.zig:456", "set_transform", nearM(s.transform, Matrix::identity().translate(2, 3, 4)));
    expectTrue("shape.zig:459", "set_transform_inverse", nearM(s.inverse, Matrix::identity().translate(-2, -3, -4)));
  }
  {  // bounds() of the kinds whose tests are three lines: sphere.zig:186, cube.zig:211, shape.zig:631, cylinder.zig:334
    auto unit = [](ShapeKind k) {
      const sc::Box b = sc::shapeBounds(Shape::make(k));
      return bitEq(b.min, point(-1, -1, -1)) && bitEq(b.max, point(1, 1, 1));
    };
    expectTrue("sphere.zig:186", "scene_sphere_bounds", unit(SPHERE));
    expectTrue("cube.zig:211", "scene_cube_bounds", unit(CUBE));
    expectTrue("shape.zig:631", "scene_test_shape_bounds", unit(TEST_SHAPE));
    const sc::Box c = sc::shapeBounds(Shape::make(CYLINDER));
    expectTrue("cylinder.zig:334", "scene_unbounded_cylinder_bounds", bitEq(c.min, point(-1, -INF, -1)) && bitEq(c.max, point(1, INF, 1)));
  }
  {  // triangle.zig:289-299 Constructing a smooth triangle; :344-357 its bounding box
    const Shape t = Shape::smoothTriangle(point(0, 1, 0), point(-1, 0, 0), point(1, 0, 0), vec3(0, 1, 0), vec3(-1, 0, 0), vec3(1, 0, 0));
    expectTrue("triangle.zig:289", "smooth_construction",
               bitEq(t.p1, point(0, 1, 0)) && bitEq(t.p2, point(-1, 0, 0)) && bitEq(t.p3, point(1, 0, 0)) &&
                   bitEq(t.n1, vec3(0, 1, 0)) && bitEq(t.n2, vec3(-1, 0, 0)) && bitEq(t.n3, vec3(1, 0, 0)));
    const Shape b = Shape::smoothTriangle(point(-3, 7, 2), point(6, 2, -4), point(2, -1, -1), vec3(0, 0, 0), vec3(0, 0, 0), vec3(0, 0, 0));
    expectTrue("triangle.zig:344", "scene_smooth_triangle_bounds",
               bitEq(sc::shapeBounds(b).min, point(-3, -1, -4)) && bitEq(sc::shapeBounds(b).max, point(6, 7, 2)));
  }
  {  // group.zig:139-147 Creating a new group; :149-161 Adding a child to a group
    Shape g = sc::newGroup();
    expectTrue("group.zig:139", "scene_new_group", bitEq(g.transform, Matrix::identity()) && g.children.empty());
    const Shape s = Shape::make(TEST_SHAPE);
    sc::addChild(g, s);
    expectTrue("group.zig:149", "scene_add_child", g.children.size() == 1 && g.children[0].id == s.id && g.children[0].kind == TEST_SHAPE &&
                                                        bitEq(g.children[0].transform, s.transform));
  }
  {  // gradient.zig:59-77 Gradient (the two ends), :79-103 RadialGradient
    const Color white{1, 1, 1}, black{0, 0, 0};
    const Pattern sw = solidPattern(white), sb = solidPattern(black);
    Pattern gr;
    gr.kind = PAT_GRADIENT;
    gr.a = &sw;
    gr.b = &sb;
    expectColor("gradient.zig:69", "gradient_0", gr.patternAt(point(0, 0, 0)), white, 0.0);
    expectColor("gradient.zig:75", "gradient_0.75", gr.patternAt(point(0.75, 0, 0)), {0.25, 0.25, 0.25}, 0.0);
    Pattern rad;
    rad.kind = PAT_RADIAL_GRADIENT;
    rad.a = &sw;
    rad.b = &sb;
    const struct { Tuple p; double want; double tol; } rows[] = {
        {point(0, 0, 0), 1.0, 0.0},    {point(0.25, 0, 0), 0.75, 0.0}, {point(0.5, 0, 0), 0.5, 0.0}, {point(0.75, 0, 0), 0.25, 0.0},
        {point(0, 0, 0.25), 0.75, 0.0}, {point(0, 0, 0.5), 0.5, 0.0},   {point(0.353553, 0, 0.353553), 0.5, 1e-5}};
    int i = 0;
    for (const auto& r : rows)
      expectColor(("gradient.zig:" + std::to_string(89 + 2 * i)).c_str(), "radial_gradient_" + std::to_string(i), rad.patternAt(r.p),
                  {r.want, r.want, r.want}, r.tol), ++i;
  }
  {  // texture_map.zig:432-442 Identifying the face of a cube from a point: six faces of one colour each
    const Color cols[6] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {0, 1, 1}, {1, 0, 1}};  // front, back, left, right, up, down
    Pattern solid[6];
    TextureMap cube;
    cube.mapping = TEX_CUBIC;
    for (int f = 0; f < 6; ++f) {
      solid[f] = solidPattern(cols[f]);
      cube.faces[f].kind = UV_ALIGN_CHECK;
      for (int k = 0; k < 5; ++k) cube.faces[f].sub[k] = &solid[f];
    }
    const struct { Tuple p; int face; const char* name; } rows[] = {
        {point(-1, 0.5, -0.25), 2, "face_left"}, {point(1.1, -0.75, 0.8), 3, "face_right"}, {point(0.1, 0.6, 0.9), 0, "face_front"},
        {point(-0.7, 0, -2), 1, "face_back"},    {point(0.5, 1, 0.9), 4, "face_up"},        {point(-0.2, -1.3, 1.1), 5, "face_down"}};
    int i = 0;
    for (const auto& r : rows)
      expectColor(("texture_map.zig:" + std::to_string(436 + i)).c_str(), r.name, cube.patternAt(r.p, point(0, 0, 0)), cols[r.face], 0.0), ++i;
  }
}

int main() {
  tupleKats();
  matrixKats();
  rayKats();
  sphereKats();
  planeKats();
  cubeKats();
  cylinderKats();
  coneKats();
  triangleKats();
  groupAndBoxKats();
  nestedGroupKats();
  refractionIndexKats();
  materialKats();
  patternKats();
  textureMapKats();
  csgKats();
  noiseKats();
  powKats();
  worldKats();
  cameraKats();
  sceneBoxKats();
  sceneGroupKats();
  sceneObjKats();
  sceneParseKats();
  auditKats();
  std::printf("KAT-SUMMARY total=%d failed=%d\n", g_total, g_failed);
  return g_failed ? 1 : 0;
}
