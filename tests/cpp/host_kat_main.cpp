// host_kat_main.cpp — known-answer tests for the PRODUCT host library's build-time,
// result-affecting helpers and the loaders either side of the hot path,
// restated from the reference's own inline tests:
//   matrix.zig:330-683, tuple.zig:146-216        -> rtc_math.hpp
//   bounding_box.zig:183-252, 362-423            -> BoundingBox add/contains/transform/split
//   group.zig:219-381, shape.zig:619-654         -> addChild / partitionChildren / makeSubgroup / divide
//   camera.zig:129-142                           -> Camera::create
//   parsing/scene.zig:664-774                    -> parseScene
//   parsing/obj.zig:288-544                      -> ObjParser
//   canvas.zig:258-303                           -> Canvas::ppm
// Output format is the same as oracle/kat_main.cpp: "KAT <where> <name> PASS|FAIL".
#include <cstdio>
#include <cstring>
#include <string>

#include "../../ray-tracer-challenge_amd/host/rtc_api.hpp"
#include "../../ray-tracer-challenge_amd/host/rtc_loader.hpp"

using namespace rtc;

static int g_failed = 0, g_total = 0;
static void report(const char* where, const std::string& name, bool ok, const std::string& detail = "") {
  ++g_total;
  if (!ok) ++g_failed;
  std::printf("KAT %s %s %s%s%s\n", where, name.c_str(), ok ? "PASS" : "FAIL", detail.empty() ? "" : " ", detail.c_str());
}
static bool near(double a, double b, double tol = 1e-5) { return std::fabs(a - b) <= tol; }
static bool nearT(const Tuple& a, const Tuple& b, double tol = 1e-5) {
  return near(a.x, b.x, tol) && near(a.y, b.y, tol) && near(a.z, b.z, tol) && near(a.w, b.w, tol);
}
static std::string fmt(const Tuple& t) {
  char b[160];
  std::snprintf(b, sizeof b, "(%.8g,%.8g,%.8g,%.8g)", t.x, t.y, t.z, t.w);
  return b;
}
static void expectT(const char* where, const std::string& name, const Tuple& got, const Tuple& want, double tol = 1e-5) {
  report(where, name, nearT(got, want, tol), "got " + fmt(got) + " want " + fmt(want));
}
static const double PI = 3.14159265358979323846;
static const double RS2 = 1.0 / std::sqrt(2.0);

static void mathKats() {
  const double a_[4][4] = {{1, 2, 3, 4}, {5, 6, 7, 8}, {9, 8, 7, 6}, {5, 4, 3, 2}};
  const double b_[4][4] = {{-2, 1, 2, 3}, {3, 2, 1, -1}, {4, 3, 6, 5}, {1, 2, 7, 8}};
  const double ab_[4][4] = {{20, 22, 50, 48}, {44, 54, 114, 108}, {40, 58, 110, 102}, {16, 26, 46, 42}};
  report("matrix.zig:352", "mul", Matrix4::rows(a_).mul(Matrix4::rows(b_)).approxEqual(Matrix4::rows(ab_)));
  const double d_[4][4] = {{-2, -8, 3, 5}, {-3, 1, 7, 3}, {1, 2, -9, 6}, {-6, 7, 7, -9}};
  report("matrix.zig:450", "det4", near(Matrix4::rows(d_).det(), -4071.0));
  report("matrix.zig:446", "cofactor00", near(Matrix4::rows(d_).cofactor(0, 0), 690.0));
  const double ia_[4][4] = {{8, -5, 9, 2}, {7, 5, 6, 1}, {-6, 0, 9, 6}, {-3, 0, -9, -4}};
  const double iai_[4][4] = {{-0.15385, -0.15385, -0.28205, -0.53846}, {-0.07692, 0.12308, 0.02564, 0.03077},
                             {0.35897, 0.35897, 0.43590, 0.92308},     {-0.69231, -0.69231, -0.76923, -1.92308}};
  report("matrix.zig:468", "inverse_a", Matrix4::rows(ia_).inverse().approxEqual(Matrix4::rows(iai_)));
  const double ib_[4][4] = {{9, 3, 0, 9}, {-5, -2, -6, -3}, {-4, 9, 6, 4}, {-7, 6, 6, 2}};
  const double ibi_[4][4] = {{-0.04074, -0.07778, 0.14444, -0.22222}, {-0.07778, 0.03333, 0.36667, -0.33333},
                             {-0.02901, -0.14630, -0.10926, 0.12963},  {0.17778, 0.06667, -0.26667, 0.33333}};
  report("matrix.zig:484", "inverse_b", Matrix4::rows(ib_).inverse().approxEqual(Matrix4::rows(ibi_)));
  bool threw = false;
  try {
    const double s_[4][4] = {{-4, 2, -2, -3}, {9, 6, 2, 6}, {0, -5, 1, -5}, {0, 0, 0, 0}};
    Matrix4::rows(s_).inverse();
  } catch (const Error& e) {
    threw = e.name == "NotInvertible";
  }
  report("matrix.zig:204", "NotInvertible", threw);
  const Matrix4 I = Matrix4::identity();
  expectT("matrix.zig:581", "chained", I.rotateX(PI / 2).scale(5, 5, 5).translate(10, 5, 7).tupleMul(Tuple::point(1, 0, 1)),
          Tuple::point(15, 0, 7));
  ShearArgs sh;
  sh.xz = 1.0;
  expectT("matrix.zig:563", "shear_xz", I.shear(sh).tupleMul(Tuple::point(2, 3, 4)), Tuple::point(6, 3, 4));
  const double vw_[4][4] = {{-0.50709, 0.50709, 0.67612, -2.36643},
                            {0.76772, 0.60609, 0.12122, -2.82843},
                            {-0.35857, 0.59761, -0.71714, 0.00000},
                            {0.00000, 0.00000, 0.00000, 1.00000}};
  report("matrix.zig:633", "view_arbitrary",
         Matrix4::viewTransform(Tuple::point(1, 3, 2), Tuple::point(4, -2, 8), Tuple::vec3(1, 1, 0)).approxEqual(Matrix4::rows(vw_)));
  report("matrix.zig:676", "rodrigues_x", I.rotateX(1.0).approxEqual(I.rotate(Tuple::vec3(1, 0, 0), 1.0)));
  report("matrix.zig:679", "rodrigues_y", I.rotateY(1.0).approxEqual(I.rotate(Tuple::vec3(0, 1, 0), 1.0)));
  report("matrix.zig:682", "rodrigues_z", I.rotateZ(1.0).approxEqual(I.rotate(Tuple::vec3(0, 0, 1), 1.0)));
  expectT("tuple.zig:193", "cross", Tuple::vec3(1, 2, 3).cross(Tuple::vec3(2, 3, 4)), Tuple::vec3(-1, 2, -1));
  expectT("tuple.zig:209", "reflect", Tuple::vec3(0, -1, 0).reflect(Tuple::vec3(RS2, RS2, 0)), Tuple::vec3(1, 0, 0));
  report("color.zig:74", "clamp", clampChannel(1.5) == 255 && clampChannel(0.5) == 128 && clampChannel(-0.5) == 0);
}

static void boxKats() {
  BoundingBox box;
  box.add(Tuple::point(-5, 2, 0));
  box.add(Tuple::point(7, 0, -3));
  report("bounding_box.zig:183", "add_points", box.min.bitEqual(Tuple::point(-5, 0, -3)) && box.max.bitEqual(Tuple::point(7, 2, 0)));
  BoundingBox b2;
  b2.min = Tuple::point(5, -2, 0);
  b2.max = Tuple::point(11, 4, 7);
  const Tuple in[] = {Tuple::point(5, -2, 0), Tuple::point(11, 4, 7), Tuple::point(8, 1, 3)};
  const Tuple outp[] = {Tuple::point(3, 0, 3), Tuple::point(8, -4, 3), Tuple::point(8, 1, -1),
                        Tuple::point(13, 1, 3), Tuple::point(8, 5, 3), Tuple::point(8, 1, 8)};
  bool ok = true;
  for (const Tuple& p : in) ok = ok && b2.containsPoint(p);
  for (const Tuple& p : outp) ok = ok && !b2.containsPoint(p);
  report("bounding_box.zig:193", "contains_point", ok);
  auto cb = [&](Tuple mn, Tuple mx) {
    BoundingBox o;
    o.min = mn;
    o.max = mx;
    return b2.containsBox(o);
  };
  report("bounding_box.zig:242", "contains_box",
         cb(Tuple::point(5, -2, 0), Tuple::point(11, 4, 7)) && cb(Tuple::point(6, -1, 1), Tuple::point(10, 3, 6)) &&
             !cb(Tuple::point(4, -3, -1), Tuple::point(10, 3, 6)) && !cb(Tuple::point(6, -1, 1), Tuple::point(12, 5, 8)));
  BoundingBox unit;
  unit.min = Tuple::point(-1, -1, -1);
  unit.max = Tuple::point(1, 1, 1);
  const BoundingBox tb = unit.transform(Matrix4::identity().rotateY(PI / 4).rotateX(PI / 4));
  expectT("bounding_box.zig:262", "transform_min", tb.min, Tuple::point(-1.41421, -1.7071, -1.7071));
  expectT("bounding_box.zig:265", "transform_max", tb.max, Tuple::point(1.41421, 1.7071, 1.7071));
  struct S { Tuple mn, mx, lmax, rmin; };
  const S splits[] = {{Tuple::point(-1, -4, -5), Tuple::point(9, 6, 5), Tuple::point(4, 6, 5), Tuple::point(4, -4, -5)},
                      {Tuple::point(-1, -2, -3), Tuple::point(9, 5.5, 3), Tuple::point(4, 5.5, 3), Tuple::point(4, -2, -3)},
                      {Tuple::point(-1, -2, -3), Tuple::point(5, 8, 3), Tuple::point(5, 3, 3), Tuple::point(-1, 3, -3)},
                      {Tuple::point(-1, -2, -3), Tuple::point(5, 3, 7), Tuple::point(5, 3, 2), Tuple::point(-1, -2, 2)}};
  int i = 0;
  for (const S& s : splits) {
    BoundingBox b;
    b.min = s.mn;
    b.max = s.mx;
    const auto h = b.split();
    report(("bounding_box.zig:" + std::to_string(365 + 15 * i)).c_str(), "split_" + std::to_string(i),   // the four tests at :365, :380, :395, :410
           h.first.min.bitEqual(s.mn) && nearT(h.first.max, s.lmax) && nearT(h.second.min, s.rmin) && h.second.max.bitEqual(s.mx));
    ++i;
  }
  // per-kind bounds
  Shape cyl = Shape::cylinder();
  cyl.ymin = -5;
  cyl.ymax = 3;
  report("cylinder.zig:342", "cylinder_bounds",
         cyl.bounds().min.bitEqual(Tuple::point(-1, -5, -1)) && cyl.bounds().max.bitEqual(Tuple::point(1, 3, 1)));
  {  // cone.zig:241-260 (the reference's second test is named "A bounded cylinder ..." but builds a cone)
    const Shape unbounded = Shape::cone();
    report("cone.zig:241", "cone_unbounded_bounds",
           unbounded.bounds().min.x == -kInf && unbounded.bounds().min.y == -kInf && unbounded.bounds().min.z == -kInf &&
               unbounded.bounds().max.x == kInf && unbounded.bounds().max.y == kInf && unbounded.bounds().max.z == kInf);
    Shape cone = Shape::cone();
    cone.ymin = -5;
    cone.ymax = 3;
    report("cone.zig:249", "cone_bounds",
           cone.bounds().min.bitEqual(Tuple::point(-5, -5, -5)) && cone.bounds().max.bitEqual(Tuple::point(5, 3, 5)));
  }
  {  // the three-line bounds() tests: sphere.zig:186, cube.zig:211, shape.zig:631, cylinder.zig:334, triangle.zig:344
    auto unit = [](const Shape& sh) {
      return sh.bounds().min.bitEqual(Tuple::point(-1, -1, -1)) && sh.bounds().max.bitEqual(Tuple::point(1, 1, 1));
    };
    report("sphere.zig:186", "sphere_bounds", unit(Shape::sphere()));
    report("cube.zig:211", "cube_bounds", unit(Shape::cube()));
    report("shape.zig:631", "test_shape_bounds", unit(Shape::testShape()));
    const Shape open_cyl = Shape::cylinder();
    report("cylinder.zig:334", "unbounded_cylinder_bounds",
           open_cyl.bounds().min.bitEqual(Tuple::point(-1, -kInf, -1)) && open_cyl.bounds().max.bitEqual(Tuple::point(1, kInf, 1)));
    const Shape st = Shape::smoothTriangle(Tuple::point(-3, 7, 2), Tuple::point(6, 2, -4), Tuple::point(2, -1, -1), Tuple::vec3(0, 0, 0),
                                           Tuple::vec3(0, 0, 0), Tuple::vec3(0, 0, 0));
    report("triangle.zig:344", "smooth_triangle_bounds",
           st.bounds().min.bitEqual(Tuple::point(-3, -1, -4)) && st.bounds().max.bitEqual(Tuple::point(6, 7, 2)));
  }
  const Shape pl = Shape::plane();
  report("plane.zig:109", "plane_bounds", pl.bounds().min.x == -kInf && pl.bounds().min.y == 0 && pl.bounds().max.z == kInf);
  const Shape tri = Shape::triangle(Tuple::point(-3, 7, 2), Tuple::point(6, 2, -4), Tuple::point(2, -1, -1));
  report("triangle.zig:198", "triangle_bounds",
         tri.bounds().min.bitEqual(Tuple::point(-3, -1, -4)) && tri.bounds().max.bitEqual(Tuple::point(6, 7, 2)));
  // parent space bounds (shape.zig:638-654)
  Shape s = Shape::sphere();
  s.setTransform(Matrix4::identity().scale(0.5, 2, 4).translate(1, -3, 5));
  expectT("shape.zig:652", "parent_space_min", s.parentSpaceBounds().min, Tuple::point(0.5, -5, 1));
  expectT("shape.zig:653", "parent_space_max", s.parentSpaceBounds().max, Tuple::point(1.5, -1, 9));
}

static void groupKats() {
  {  // group.zig:139-147 Creating a new group; :149-161 Adding a child; shape.zig:441-462 ids and setTransform
    Shape g = Shape::group();
    report("group.zig:139", "new_group", [&] { const Matrix4 id = Matrix4::identity(); return std::memcmp(&g.transform, &id, sizeof id) == 0; }() && g.children.empty());
    const Shape t = Shape::testShape();
    g.addChild(t);
    report("group.zig:149", "add_child", g.children.size() == 1 && g.children[0].id == t.id && g.children[0].kind == ShapeKind::TestShape);
    const Shape s1 = Shape::sphere(), s2 = Shape::sphere(), s3 = Shape::testShape();
    report("shape.zig:441", "id_uniqueness", s1.id != s2.id && s2.id != s3.id && s3.id != s1.id);
    Shape moved = Shape::testShape();
    moved.setTransform(Matrix4::identity().translate(2, 3, 4));
    const Matrix4 want = Matrix4::identity().translate(-2, -3, -4);
    bool ok = true;
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) ok = ok && std::fabs(moved.inverse.d[r][c] - want.d[r][c]) < 1e-5;
    report("shape.zig:450", "creation_set_transform", ok);
  }
  {  // group.zig:219-241
    Shape s = Shape::sphere();
    s.setTransform(Matrix4::identity().scale(2, 2, 2).translate(2, 5, -3));
    Shape c = Shape::cylinder();
    c.ymin = -2;
    c.ymax = 2;
    c.setTransform(Matrix4::identity().scale(0.5, 1, 0.5).translate(-4, -1, 4));
    Shape g = Shape::group();
    g.addChild(s);
    g.addChild(c);
    expectT("group.zig:240", "group_bounds_min", g.bounds().min, Tuple::point(-4.5, -3, -5));
    expectT("group.zig:240", "group_bounds_max", g.bounds().max, Tuple::point(4, 7, 4.5));
  }
  {  // group.zig:243-273 partition
    Shape s1 = Shape::sphere();
    s1.setTransform(Matrix4::identity().translate(-2, 0, 0));
    Shape s2 = Shape::sphere();
    s2.setTransform(Matrix4::identity().translate(2, 0, 0));
    Shape s3 = Shape::sphere();
    const size_t i1 = s1.id, i2 = s2.id, i3 = s3.id;
    Shape g = Shape::group();
    g.addChild(s1);
    g.addChild(s2);
    g.addChild(s3);
    auto parts = g.partitionChildren();
    report("group.zig:270", "partition",
           g.children.size() == 1 && g.children[0].id == i3 && parts.first.size() == 1 && parts.first[0].id == i1 &&
               parts.second.size() == 1 && parts.second[0].id == i2);
  }
  {  // group.zig:275-292 makeSubgroup
    Shape g = Shape::group();
    std::vector<Shape> kids{Shape::sphere(), Shape::sphere()};
    g.makeSubgroup(std::move(kids));
    report("group.zig:291", "make_subgroup", g.children.size() == 1 && g.children[0].isGroup() && g.children[0].children.size() == 2);
  }
  {  // group.zig:294-330 divide(1)
    Shape s1 = Shape::sphere();
    s1.setTransform(Matrix4::identity().translate(-2, -2, 0));
    Shape s2 = Shape::sphere();
    s2.setTransform(Matrix4::identity().translate(-2, 2, 0));
    Shape s3 = Shape::sphere();
    s3.setTransform(Matrix4::identity().scale(4, 4, 4));
    const size_t i1 = s1.id, i2 = s2.id, i3 = s3.id;
    Shape g = Shape::group();
    g.addChild(s1);
    g.addChild(s2);
    g.addChild(s3);
    g.divide(1);
    bool ok = g.children.size() == 2 && g.children[0].id == i3 && g.children[1].isGroup();
    if (ok) {
      const Shape& sub = g.children[1];
      ok = sub.children.size() == 2 && sub.children[0].isGroup() && sub.children[0].children.size() == 1 &&
           sub.children[0].children[0].id == i1 && sub.children[1].children.size() == 1 && sub.children[1].children[0].id == i2;
    }
    report("group.zig:313", "divide_1", ok);
  }
  {  // group.zig:332-381 divide(3) with too few children at the top
    Shape s1 = Shape::sphere();
    s1.setTransform(Matrix4::identity().translate(-2, 0, 0));
    Shape s2 = Shape::sphere();
    s2.setTransform(Matrix4::identity().translate(2, 1, 0));
    Shape s3 = Shape::sphere();
    s3.setTransform(Matrix4::identity().translate(2, -1, 0));
    Shape s4 = Shape::sphere();
    const size_t i1 = s1.id, i2 = s2.id, i3 = s3.id, i4 = s4.id;
    Shape sub = Shape::group();
    sub.addChild(s1);
    sub.addChild(s2);
    sub.addChild(s3);
    Shape g = Shape::group();
    g.addChild(sub);
    g.addChild(s4);
    g.divide(3);
    bool ok = g.children.size() == 2 && g.children[0].isGroup() && g.children[0].children.size() == 2 && g.children[1].id == i4;
    if (ok) {
      const Shape& a = g.children[0].children[0];
      const Shape& b = g.children[0].children[1];
      ok = a.children.size() == 1 && a.children[0].id == i1 && b.children.size() == 2 && b.children[0].id == i2 &&
           b.children[1].id == i3;
    }
    report("group.zig:363", "divide_3", ok);
  }
  {  // group.zig:201-217: group transform pushed to the leaf
    Shape s = Shape::sphere();
    s.setTransform(Matrix4::identity().translate(5, 0, 0));
    Shape g = Shape::group();
    g.addChild(s);
    g.setTransform(Matrix4::identity().scale(2, 2, 2));
    report("shape.zig:288", "group_pushes_transform",
           g.transform.bitEqual(Matrix4::identity()) &&
               g.children[0].transform.bitEqual(Matrix4::identity().scale(2, 2, 2).mul(Matrix4::identity().translate(5, 0, 0))));
    expectT("shape.zig:294", "group_rebox_min", g.bbox.min, Tuple::point(8, -2, -2));
    expectT("shape.zig:294", "group_rebox_max", g.bbox.max, Tuple::point(12, 2, 2));
  }
  report("camera.zig:133", "pixel_size_landscape", near(Camera::create(200, 125, PI / 2).pixel_size, 0.01));
  report("camera.zig:139", "pixel_size_portrait", near(Camera::create(125, 200, PI / 2).pixel_size, 0.01));
}

static void sceneKats() {  // scene.zig:664-774
  const char* scene = R"({
     "camera": { "width": 1280, "height": 1000, "field-of-view": 0.785,
                 "from": [ -6, 6, -10 ], "to": [ 6, 0, 6 ], "up": [ -0.45, 1, 0 ] },
     "objects": [ { "type": { "sphere": {} },
                    "transform": [ { "translate": [1.0, 2.0, 3.0] }, { "scale": [0.5, 0.5, 0.5] } ],
                    "material": { "pattern": { "type": { "stripes": [ { "type": { "solid": [1.0, 1.0, 1.0] } },
                                                                       { "type": { "solid": [0.0, 0.0, 0.0] } } ] },
                                               "transform": [ { "scale": [0.1, 0.1, 0.1] } ] },
                                  "reflective": 0.5 } } ],
     "lights": [ { "point-light": { "position": [-10.0, 10.0, -10.0], "intensity": [1.0, 1.0, 1.0] } } ] })";
  const SceneInfo info = parseScene(scene, directoryLoader(""));
  Camera expected = Camera::create(1280, 1000, 0.785);
  expected.setTransform(Matrix4::viewTransform(Tuple::point(-6, 6, -10), Tuple::point(6, 0, 6), Tuple::vec3(-0.45, 1, 0)));
  report("scene.zig:724", "camera",
         info.camera.hsize == 1280 && info.camera.vsize == 1000 && info.camera.pixel_size == expected.pixel_size &&
             info.camera.half_width == expected.half_width && info.camera.half_height == expected.half_height &&
             info.camera.transform.bitEqual(expected.transform) && info.camera.inverse.bitEqual(expected.inverse));
  report("scene.zig:737", "one_object", info.world.objects.size() == 1);
  const Shape& o = info.world.objects[0];
  report("scene.zig:726", "object_transform",
         o.kind == ShapeKind::Sphere && o.transform.bitEqual(Matrix4::identity().translate(1, 2, 3).scale(0.5, 0.5, 0.5)));
  Pattern ep = Pattern::binary(PatternKind::Stripes, Pattern::solid({1, 1, 1}), Pattern::solid({0, 0, 0}));
  ep.setTransform(Matrix4::identity().scale(0.1, 0.1, 0.1));
  report("scene.zig:740", "pattern",
         o.material.pattern.kind == PatternKind::Stripes && o.material.pattern.transform.bitEqual(ep.transform) &&
             o.material.pattern.inverse.bitEqual(ep.inverse) && o.material.pattern.a->rgb.r == 1.0 &&
             o.material.pattern.b->rgb.r == 0.0);
  report("scene.zig:748", "material", o.material.ambient == 0.1 && o.material.reflective == 0.5);
  report("scene.zig:771", "light",
         info.world.lights.size() == 1 && info.world.lights[0].position.bitEqual(Tuple::point(-10, 10, -10)) &&
             info.world.lights[0].intensity.r == 1.0);
  // error names
  auto errName = [](const char* js) -> std::string {
    try {
      parseScene(js, directoryLoader(""));
    } catch (const Error& e) {
      return e.name;
    }
    return "";
  };
  const char* cam = R"("camera":{"width":2,"height":2,"field-of-view":1,"from":[0,0,-5],"to":[0,0,0],"up":[0,1,0]},"lights":[])";
  report("scene.zig:493", "UnknownDefinition",
         errName((std::string("{") + cam + R"(,"objects":[{"type":{"from-definition":"nope"}}]})").c_str()) == "UnknownDefinition");
  report("scene.zig:578", "NotInvertible",
         errName((std::string("{") + cam + R"(,"objects":[{"type":{"sphere":{}},"transform":[{"scale":[0,1,1]}]}]})").c_str()) ==
             "NotInvertible");
  report("scene.zig:203", "MissingField", errName(R"({"lights":[],"objects":[]})") == "MissingField");
  report("scene.zig:203", "UnknownField", errName((std::string("{") + cam + R"(,"objects":[],"extra":1})").c_str()) == "UnknownField");
  // from-definition + group transform push-down + inherited material (scene.zig:455-546)
  const std::string defs = std::string("{") + cam + R"(,
    "shape-definitions":[{"name":"leg","value":{"type":{"sphere":{}},"transform":[{"scale":[2,2,2]}],"material":{"ambient":0.3}}}],
    "objects":[{"type":{"group":[{"type":{"from-definition":"leg"},"transform":[{"translate":[1,0,0]}],"material":{"diffuse":0.4}}]},
                "transform":[{"translate":[0,5,0]}],"material":{"specular":0.25}}]})";
  const SceneInfo di = parseScene(defs, directoryLoader(""));
  bool ok = di.world.objects.size() == 1 && di.world.objects[0].isGroup() && di.world.objects[0].children.size() == 1;
  if (ok) {
    const Shape& leaf = di.world.objects[0].children[0];
    const Matrix4 want = Matrix4::identity().translate(0, 5, 0).mul(
        Matrix4::identity().translate(1, 0, 0).mul(Matrix4::identity().scale(2, 2, 2).mul(Matrix4::identity())));
    ok = leaf.transform.bitEqual(want) && leaf.material.ambient == 0.3 && leaf.material.diffuse == 0.4 &&
         leaf.material.specular == 0.25;
  }
  report("scene.zig:455", "from_definition_in_group", ok);
}

static void objKats() {  // obj.zig:288-544
  {
    ObjParser p;
    p.loadObj("There was a young lady named Bright\nwho traveled much faster than light.\nShe set out one day\nin a relative way,\nand came back the previous night.", {}, false);
    report("obj.zig:288", "ignored_lines", p.lines_ignored == 5);
  }
  {
    ObjParser p;
    p.loadObj("v -1 1 0\nv -1.0000 0.5000 0.0000\nv 1 0 0\nv 1 1 0", {}, false);
    report("obj.zig:309", "vertices",
           p.lines_ignored == 0 && p.vertices.size() == 4 && p.vertices[0].bitEqual(Tuple::point(-1, 1, 0)) &&
               p.vertices[1].bitEqual(Tuple::point(-1, 0.5, 0)) && p.vertices[3].bitEqual(Tuple::point(1, 1, 0)));
  }
  {
    ObjParser p;
    p.loadObj("v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\nf 1 2 3\nf 1 3 4", {}, false);
    const auto& c = p.default_group.children;
    report("obj.zig:343", "faces",
           p.lines_ignored == 0 && c.size() == 2 && c[0].p1.bitEqual(p.vertices[0]) && c[0].p2.bitEqual(p.vertices[1]) &&
               c[0].p3.bitEqual(p.vertices[2]) && c[1].p2.bitEqual(p.vertices[2]) && c[1].p3.bitEqual(p.vertices[3]));
  }
  {
    ObjParser p;
    p.loadObj("v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\nv 0 2 0\nf 1 2 3 4 5", {}, false);
    const auto& c = p.default_group.children;
    report("obj.zig:377", "fan_triangulation",
           c.size() == 3 && c[1].p2.bitEqual(p.vertices[2]) && c[1].p3.bitEqual(p.vertices[3]) && c[2].p2.bitEqual(p.vertices[3]) &&
               c[2].p3.bitEqual(p.vertices[4]));
  }
  {
    ObjParser p;
    p.loadObj("v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\ng FirstGroup\nf 1 2 3\ng SecondGroup\nf 1 3 4", {}, false);
    bool ok = p.lines_ignored == 0 && p.named_groups.count("FirstGroup") && p.named_groups.count("SecondGroup");
    if (ok) {
      const Shape& g1 = p.default_group.children[p.named_groups["FirstGroup"]];
      const Shape& g2 = p.default_group.children[p.named_groups["SecondGroup"]];
      ok = g1.children.size() == 1 && g2.children.size() == 1 && g1.children[0].p3.bitEqual(p.vertices[2]) &&
           g2.children[0].p3.bitEqual(p.vertices[3]);
      const Shape g = p.toGroup();
      ok = ok && g.children.size() == 2 && g.children[0].isGroup();
    }
    report("obj.zig:416", "named_groups", ok);
  }
  {
    ObjParser p;
    p.loadObj("vn 0 0 1\nvn 0.707 0 -0.707\nvn 1 2 3", {}, false);
    report("obj.zig:482", "normals", p.normals.size() == 3 && p.normals[1].bitEqual(Tuple::vec3(0.707, 0, -0.707)));
  }
  {
    ObjParser p;
    p.loadObj("v 0 1 0\nv -1 0 0\nv 1 0 0\nvn -1 0 0\nvn 1 0 0\nvn 0 1 0\nf 1//3 2//1 3//2\nf 1/0/3 2/102/1 3/14/2", {}, false);
    const auto& c = p.default_group.children;
    bool ok = p.lines_ignored == 0 && c.size() == 2 && c[0].kind == ShapeKind::SmoothTriangle;
    if (ok)
      ok = c[0].n1.bitEqual(p.normals[2]) && c[0].n2.bitEqual(p.normals[0]) && c[0].n3.bitEqual(p.normals[1]) &&
           c[1].n1.bitEqual(c[0].n1) && c[1].n2.bitEqual(c[0].n2) && c[1].p3.bitEqual(c[0].p3);
    report("obj.zig:511", "faces_with_normals", ok);
  }
  {  // normalisation: offset = box centre, scale = half the longest extent; w becomes 1/scale (obj.zig:66)
    ObjParser p;
    p.loadObj("v 0 0 0\nv 4 2 1\nf 1 2 2", {}, true);
    report("obj.zig:257", "normalize",
           p.scale == 2.0 && p.offset.x == 2.0 && p.offset.y == 1.0 && p.offset.z == 0.5 && p.vertices[0].x == -1.0 &&
               p.vertices[1].x == 1.0 && p.vertices[0].w == 0.5);
  }
}

static void canvasKats() {  // canvas.zig:258-303
  Canvas c = Canvas::create(5, 3);
  c.getPixelPointerMut(0, 0)->r = 1.5;
  c.getPixelPointerMut(2, 1)->g = 0.5;
  *c.getPixelPointerMut(4, 2) = Color{-0.5, 0.0, 1.0};
  report("canvas.zig:268", "ppm_small",
         c.ppm() == "P3\n5 3\n255\n255 0 0 0 0 0 0 0 0 0 0 0 0 0 0\n0 0 0 0 0 0 0 128 0 0 0 0 0 0 0\n0 0 0 0 0 0 0 0 0 0 0 0 0 0 255\n");
  Canvas c2 = Canvas::create(10, 2);
  for (Color& p : c2.pixels) p = Color{1, 0.8, 0.6};
  report("canvas.zig:290", "ppm_wrap",
         c2.ppm() == "P3\n10 2\n255\n"
                     "255 204 153 255 204 153 255 204 153 255 204 153 255 204 153 255 204\n"
                     "153 255 204 153 255 204 153 255 204 153 255 204 153\n"
                     "255 204 153 255 204 153 255 204 153 255 204 153 255 204 153 255 204\n"
                     "153 255 204 153 255 204 153 255 204 153 255 204 153\n");
  report("canvas.zig:132", "pixel_out_of_range", c.getPixelPointer(5, 0) == nullptr && c.getPixelPointer(0, 3) == nullptr);

  // canvas.zig:305-422 - the P3 reader
  auto errorOf = [](const std::string& text) -> std::string {
    try {
      Canvas::fromPpm(text);
    } catch (const Error& e) {
      return e.name;
    }
    return "";
  };
  auto px = [](const Canvas& cv, size_t x, size_t y, double r, double g, double b) {
    const Color* p = cv.getPixelPointer(x, y);
    return p && std::fabs(p->r - r) < 1e-5 && std::fabs(p->g - g) < 1e-5 && std::fabs(p->b - b) < 1e-5;
  };
  report("canvas.zig:305", "ppm_wrong_magic", errorOf("P32\n1 1\n255\n0 0 0") == "InvalidMagicNumber");
  {
    std::string text = "P3\n10 2\n255\n";
    for (int row = 0; row < 4; ++row) text += "0 0 0  0 0 0  0 0 0  0 0 0  0 0 0\n";
    const Canvas r = Canvas::fromPpm(text);
    report("canvas.zig:317", "ppm_size", r.width == 10 && r.height == 2 && r.pixels.size() == 20);
  }
  {
    const Canvas r = Canvas::fromPpm("P3\n4 3\n255\n255 127 0  0 127 255  127 255 0  255 255 255\n0 0 0  255 0 0  0 255 0  0 0 255\n"
                                     "255 255 0  0 255 255  255 0 255  127 127 127");
    report("canvas.zig:337", "ppm_pixels",
           px(r, 0, 0, 1, 0.49804, 0) && px(r, 1, 0, 0, 0.49804, 1) && px(r, 2, 0, 0.49804, 1, 0) && px(r, 3, 0, 1, 1, 1) &&
               px(r, 0, 1, 0, 0, 0) && px(r, 1, 1, 1, 0, 0) && px(r, 2, 1, 0, 1, 0) && px(r, 3, 1, 0, 0, 1) && px(r, 0, 2, 1, 1, 0) &&
               px(r, 1, 2, 0, 1, 1) && px(r, 2, 2, 1, 0, 1) && px(r, 3, 2, 0.49804, 0.49804, 0.49804));
  }
  {
    const Canvas r = Canvas::fromPpm("P3\n# this is a comment\n2 1\n# this, too\n255\n# another comment\n255 255 255\n"
                                     "# oh, no, comments in the pixel data!\n255 0 255");
    report("canvas.zig:366", "ppm_comments", px(r, 0, 0, 1, 1, 1) && px(r, 1, 0, 1, 0, 1));
  }
  {
    const Canvas r = Canvas::fromPpm("P3\n1 1\n255\n51\n153\n\n204");
    report("canvas.zig:388", "ppm_triple_spans_lines", px(r, 0, 0, 0.2, 0.6, 0.8));
  }
  {
    const Canvas r = Canvas::fromPpm("P3\n2 2\n100\n100 100 100  50 50 50\n75 50 25  0 0 0");
    report("canvas.zig:407", "ppm_scale", px(r, 0, 1, 0.75, 0.5, 0.25));
  }
  // (canvas.zig:48-121 beyond its tests: what the other error names are for, and that ppm() reads back)
  report("canvas.zig:58", "ppm_no_dimensions", errorOf("P3\n# only a comment") == "InvalidDimensions" && errorOf("P3\n3\n255\n") == "InvalidDimensions" &&
                                                     errorOf("P3\n1 1 1\n255\n0 0 0") == "InvalidDimensions");
  report("canvas.zig:78", "ppm_no_scale", errorOf("P3\n1 1\n") == "InvalidScale" && errorOf("P3\n1 1\n255 255\n0 0 0") == "InvalidScale");
  report("canvas.zig:116", "ppm_sample_count", errorOf("P3\n2 1\n255\n0 0 0") == "InvalidDimensions" && errorOf("P3\n1 1\n255\n0 0 x") == "InvalidCharacter");
  {
    const Canvas back = Canvas::fromPpm(c2.ppm());
    report("canvas.zig:181", "ppm_round_trip", back.width == 10 && back.height == 2 && px(back, 9, 1, 1.0, 0.8, 0.6));
  }
}

static void flattenKats() {
  World w = World::defaultWorld();
  const FlatScene f = flattenWorld(w);
  report("flatten", "default_world", f.leafCount() == 2 && f.nodeCount() == 0 && f.roots.size() == 2 && f.light_pos.size() == 3);
  Shape g = Shape::group();
  for (int i = 0; i < 9; ++i) {
    Shape s = Shape::sphere();
    s.setTransform(Matrix4::identity().translate(3.0 * i, 0, 0));
    g.addChild(s);
  }
  g.divide(8);
  World w2;
  w2.objects.push_back(g);
  const FlatScene f2 = flattenWorld(w2);
  report("flatten", "divided_group", f2.leafCount() == 9 && f2.nodeCount() >= 3 && f2.roots.size() == 1 &&
                                         (f2.roots[0] & RTC_CHILD_NODE_BIT) && f2.xf_inv.size() == 9 * 16);
}

static void csgKats() {  // csg.zig:143-154, shape.zig:253-302,393-396, scene.zig:547-575
  {
    Shape s1 = Shape::sphere(), s2 = Shape::cube();
    const size_t id1 = s1.id, id2 = s2.id;
    Shape c = Shape::csg(s1, s2, CsgOp::Union);
    report("csg.zig:152", "csg_left_right", c.isCsg() && c.children.size() == 2 && c.children[0].id == id1 && c.children[1].id == id2 &&
                                                c.csg_op == CsgOp::Union);
  }
  {  // the box is the union of the children's parent-space boxes at construction (shape.zig:257-261) ...
    Shape s1 = Shape::sphere();
    Shape s2 = Shape::sphere();
    s2.setTransform(Matrix4::identity().translate(0, 0, 0.5));
    Shape c = Shape::csg(s1, s2, CsgOp::Union);
    expectT("shape.zig:259", "csg_bbox_min", c.bounds().min, Tuple::point(-1, -1, -1));
    expectT("shape.zig:261", "csg_bbox_max", c.bounds().max, Tuple::point(1, 1, 1.5));
    // ... and a transform set on the csg goes to the children only: the csg keeps its box (shape.zig:298-302)
    c.setTransform(Matrix4::identity().translate(10, 0, 0));
    expectT("shape.zig:300", "csg_pushes_transform_left", c.children[0].transform.tupleMul(Tuple::point(0, 0, 0)), Tuple::point(10, 0, 0));
    expectT("shape.zig:301", "csg_pushes_transform_right", c.children[1].transform.tupleMul(Tuple::point(0, 0, 0)), Tuple::point(10, 0, 0.5));
    expectT("shape.zig:298", "csg_box_not_reboxed", c.bounds().max, Tuple::point(1, 1, 1.5));
  }
  {  // divide() recurses into both sides (shape.zig:393-396)
    Shape g = Shape::group();
    for (int i = 0; i < 9; ++i) {
      Shape s = Shape::sphere();
      s.setTransform(Matrix4::identity().translate(3.0 * i, 0, 0));
      g.addChild(s);
    }
    Shape c = Shape::csg(g, Shape::cube(), CsgOp::Difference);
    c.divide(8);
    bool subdivided = false;
    for (const Shape& k : c.children[0].children) subdivided |= k.isGroup();
    report("shape.zig:394", "csg_divide_recurses", subdivided && c.leafCount() == 10);
    World w;
    w.objects.push_back(c);
    const FlatScene f = flattenWorld(w);
    report("flatten", "csg_node", f.roots.size() == 1 && (f.roots[0] & RTC_CHILD_NODE_BIT) && f.node_op.size() == f.nodeCount() &&
                                      f.node_op[f.roots[0] & ~RTC_CHILD_NODE_BIT] == RTC_CSG_DIFFERENCE &&
                                      f.node_count[f.roots[0] & ~RTC_CHILD_NODE_BIT] == 2 && f.leafCount() == 10);
  }
  {  // scene grammar (scene.zig:147-151, 547-575)
    const char* js = R"({ "camera": { "width": 10, "height": 10, "field-of-view": 1, "from": [0,0,-5], "to": [0,0,0], "up": [0,1,0] },
      "lights": [], "objects": [ { "type": { "csg": { "operation": "intersection",
         "left": { "type": { "sphere": {} } }, "right": { "type": { "cube": {} }, "transform": [ { "translate": [0.5, 0, 0] } ] } } },
         "material": { "ambient": 0.7 } } ] })";
    const SceneInfo info = parseScene(js, directoryLoader(""));
    const Shape& c = info.world.objects[0];
    report("scene.zig:573", "parse_csg", c.isCsg() && c.csg_op == CsgOp::Intersection && c.children[0].kind == ShapeKind::Sphere &&
                                           c.children[1].kind == ShapeKind::Cube && c.children[1].material.ambient == 0.7);
    bool threw = false;
    try {
      parseScene(R"({ "camera": { "width": 1, "height": 1, "field-of-view": 1, "from": [0,0,-5], "to": [0,0,0], "up": [0,1,0] }, "lights": [],
        "objects": [ { "type": { "csg": { "operation": "xor", "left": { "type": { "sphere": {} } }, "right": { "type": { "cube": {} } } } } } ] })",
                 directoryLoader(""));
    } catch (const Error& e) {
      threw = std::string(e.name) == "InvalidEnumTag";
    }
    report("scene.zig:150", "csg_bad_operation", threw);
  }
}

static void cameraMotionKats() {  // lib.zig:166-190 (no tests in the reference: geometric identities of the two functions)
  Camera c = Camera::create(100, 50, 1.0);
  c.saved_from = Tuple::point(0, 1, -5);
  c.saved_to = Tuple::point(0, 1, 0);
  c.saved_up = Tuple::vec3(0, 1, 0);
  c.setTransform(Matrix4::viewTransform(c.saved_from, c.saved_to, c.saved_up));
  moveCamera(c, 0.2);  // a fifth of the way to the target
  expectT("lib.zig:185", "move_from", c.saved_from, Tuple::point(0, 1, -4));
  expectT("lib.zig:188", "move_view", c.transform.tupleMul(Tuple::point(0, 1, 0)), Tuple::point(0, 0, -4));
  rotateCamera(c, PI / 2);  // quarter orbit about `up` through the target
  const double r = c.saved_from.sub(c.saved_to).magnitude();
  report("lib.zig:173", "rotate_keeps_distance", std::fabs(r - 4.0) < 1e-12 && std::fabs(c.saved_from.y - 1.0) < 1e-12);
  report("lib.zig:172", "rotate_quarter_turn", std::fabs(std::fabs(c.saved_from.x) - 4.0) < 1e-9 && std::fabs(c.saved_from.z) < 1e-9);
  expectT("lib.zig:176", "rotate_view_looks_at_target", c.transform.tupleMul(Tuple::point(0, 1, 0)), Tuple::point(0, 0, -4));
  rotateCamera(c, -PI / 2);
  expectT("lib.zig:172", "rotate_back", c.saved_from, Tuple::point(0, 1, -4));
}

int main() {
  mathKats();
  boxKats();
  groupKats();
  sceneKats();
  objKats();
  canvasKats();
  flattenKats();
  csgKats();
  cameraMotionKats();
  std::printf("KAT-SUMMARY total=%d failed=%d\n", g_total, g_failed);
  return g_failed ? 1 : 0;
}
