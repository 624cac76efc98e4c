// rtc_scene.hpp — host-side scene model: Shape tree, Group/BVH build, Material,
// Pattern, Light, World, Camera.  This is the data the GPU path consumes; the
// per-pixel work itself (intersect / shade) is NOT here — it lives in
// csrc/rtc_kernels.hip.
//
// Build-time, result-affecting helpers restated from the reference:
//   Shape.setTransform            src/raytracer/shapes/shape.zig:286-310
//   Shape.parentSpaceBounds       src/raytracer/shapes/shape.zig:364-370
//   Shape.divide                  src/raytracer/shapes/shape.zig:372-399
//   Group.addChild / partitionChildren / makeSubgroup
//                                 src/raytracer/shapes/group.zig:75-135
//   BoundingBox.*                 src/raytracer/shapes/bounding_box.zig:24-110
//   Shape.triangle / smoothTriangle ctor   shape.zig:186-227
//   Camera.new / setTransform     src/raytracer/camera.zig:33-61
#pragma once
#include <memory>
#include <utility>
#include <vector>

#include "rtc_math.hpp"

namespace rtc {

// ---------------------------------------------------------------- patterns
enum class PatternKind : uint8_t {  // numeric values == RTC_PAT_* in include/rtc.h
  Solid = 0, Stripes = 1, Rings = 2, Gradient = 3, RadialGradient = 4,
  Checkers = 5, Blend = 6, Perturb = 7, TextureMap = 8, Test = 9
};

// ---------------------------------------------------------------- texture maps (patterns/texture_map.zig)
struct Pattern;
struct UvImageData {  // Canvas(T) made by Canvas.fromImage (canvas.zig:34-46): zigimg's f32 colour per pixel
  size_t width = 0, height = 0;
  std::vector<float> rgb;  // [height][width][3]
};
enum class UvKind : uint8_t { AlignCheck = 0, Checkers = 1, Image = 2, Test = 3 };        // == RTC_UV_*
enum class TexMapping : uint8_t { Spherical = 0, Planar = 1, Cylindrical = 2, Cubic = 3 };  // == RTC_TEX_*
struct UvPattern {  // texture_map.zig:107-171
  UvKind kind = UvKind::Test;
  double width = 0.0, height = 0.0;                 // UvCheckers
  std::vector<std::shared_ptr<const Pattern>> sub;  // align check: central, ul, ur, bl, br; checkers: a, b
  std::shared_ptr<const UvImageData> image;         // UvImage
  bool bilinear = false;
};
struct TextureMap {  // texture_map.zig:173-330
  TexMapping mapping = TexMapping::Spherical;
  std::vector<UvPattern> faces;  // one, or six in Cubic.Face order: front, back, left, right, up, down
};

struct Pattern {  // patterns/pattern.zig:21-49
  Matrix4 transform = Matrix4::identity();
  Matrix4 inverse = Matrix4::identity();
  PatternKind kind = PatternKind::Solid;
  Color rgb{1, 1, 1};                 // solid.zig:14
  std::shared_ptr<const Pattern> a, b;  // higher-order patterns keep pointers (stripes.zig:19-20)
  // Perturb.PerturbInfo defaults (perturb.zig:21-25); the scene grammar cannot change them (scene.zig:356)
  double perturb_scale = 0.3, perturb_persistence = 0.8;
  unsigned perturb_octaves = 3;
  std::shared_ptr<const TextureMap> texture_map;  // PatternKind::TextureMap

  static Pattern solid(Color c) {
    Pattern p;
    p.kind = PatternKind::Solid;
    p.rgb = c;
    return p;
  }
  static Pattern testPattern() {
    Pattern p;
    p.kind = PatternKind::Test;
    return p;
  }
  static Pattern binary(PatternKind k, Pattern a_, Pattern b_) {
    Pattern p;
    p.kind = k;
    p.a = std::make_shared<const Pattern>(std::move(a_));
    p.b = std::make_shared<const Pattern>(std::move(b_));
    return p;
  }
  void setTransform(const Matrix4& m) {  // pattern.zig:103-106
    transform = m;
    inverse = m.inverse();
  }
};

struct Material {  // material.zig:18-25
  Pattern pattern = Pattern::solid({1.0, 1.0, 1.0});
  double ambient = 0.1, diffuse = 0.9, specular = 0.9, shininess = 200.0;
  double reflective = 0.0, transparency = 0.0, refractive_index = 1.0;
};

struct Light {  // light.zig:14-15
  Tuple position;
  Color intensity;
};

// ---------------------------------------------------------------- bounding boxes
struct BoundingBox {  // bounding_box.zig:21-22
  Tuple min = Tuple::point(kInf, kInf, kInf);
  Tuple max = Tuple::point(-kInf, -kInf, -kInf);

  // Zig @min/@max return the non-NaN operand == C fmin/fmax.
  void add(const Tuple& p) {  // bounding_box.zig:24-32
    min.x = std::fmin(min.x, p.x);
    min.y = std::fmin(min.y, p.y);
    min.z = std::fmin(min.z, p.z);
    max.x = std::fmax(max.x, p.x);
    max.y = std::fmax(max.y, p.y);
    max.z = std::fmax(max.z, p.z);
  }
  bool containsPoint(const Tuple& p) const {  // bounding_box.zig:34-38
    return min.x <= p.x && p.x <= max.x && min.y <= p.y && p.y <= max.y && min.z <= p.z && p.z <= max.z;
  }
  bool containsBox(const BoundingBox& o) const { return containsPoint(o.min) && containsPoint(o.max); }
  void merge(const BoundingBox& o) {  // bounding_box.zig:44-47
    add(o.min);
    add(o.max);
  }
  BoundingBox transform(const Matrix4& m) const {  // bounding_box.zig:49-70 — AABB of the 8 corners
    const Tuple p1 = min;
    const Tuple p2 = Tuple::point(min.x, min.y, max.z);
    const Tuple p3 = Tuple::point(min.x, max.y, min.z);
    const Tuple p4 = Tuple::point(min.x, max.y, max.z);
    const Tuple p5 = Tuple::point(max.x, min.y, min.z);
    const Tuple p6 = Tuple::point(max.x, min.y, max.z);
    const Tuple p7 = Tuple::point(max.x, max.y, min.z);
    const Tuple p8 = max;
    BoundingBox n;
    n.add(m.tupleMul(p1));
    n.add(m.tupleMul(p2));
    n.add(m.tupleMul(p3));
    n.add(m.tupleMul(p4));
    n.add(m.tupleMul(p5));
    n.add(m.tupleMul(p6));
    n.add(m.tupleMul(p7));
    n.add(m.tupleMul(p8));
    return n;
  }
  std::pair<BoundingBox, BoundingBox> split() const {  // bounding_box.zig:72-110
    const double dx = max.x - min.x, dy = max.y - min.y, dz = max.z - min.z;
    const double greatest = std::fmax(dx, std::fmax(dy, dz));
    double x0 = min.x, y0 = min.y, z0 = min.z;
    double x1 = max.x, y1 = max.y, z1 = max.z;
    if (greatest == dx) {
      x0 = x0 + dx / 2.0;
      x1 = x0;
    } else if (greatest == dy) {
      y0 = y0 + dy / 2.0;
      y1 = y0;
    } else {
      z0 = z0 + dz / 2.0;
      z1 = z0;
    }
    BoundingBox left, right;
    left.min = min;
    left.max = Tuple::point(x1, y1, z1);
    right.min = Tuple::point(x0, y0, z0);
    right.max = max;
    return {left, right};
  }
};

// ---------------------------------------------------------------- shapes
enum class ShapeKind : uint8_t {
  Sphere, Plane, Cube, Cylinder, Cone, Triangle, SmoothTriangle, Group, TestShape, Csg
};
enum class CsgOp : uint8_t { Union = 1, Intersection = 2, Difference = 3 };  // csg.zig:16-20 (== RTC_CSG_*)

size_t nextShapeId();  // shape.zig:123-130 — process-wide counter (also bumped by bounding boxes)
// The loader builds the objects of a scene on several threads (rtc_loader.cpp): inside a ShapeIdScope the ids drawn on
// this thread count from 0; when the object is done it gets the block of process-wide ids the reference's one counter
// would have handed it at that point (reserveShapeIds, in the objects' order) and offsetShapeIds() moves its tree there.
struct ShapeIdScope {
  ShapeIdScope();
  ~ShapeIdScope();
  ShapeIdScope(const ShapeIdScope&) = delete;
  ShapeIdScope& operator=(const ShapeIdScope&) = delete;
  size_t drawn = 0;

 private:
  size_t* outer_;
};
size_t reserveShapeIds(size_t count);  // the first id of a block of `count`
struct Shape;
void offsetShapeIds(Shape& tree, size_t base);

struct Shape {
  size_t id = 0;                                   // shape.zig:113
  Matrix4 transform = Matrix4::identity();         // _transform
  Matrix4 inverse = Matrix4::identity();           // _inverse_transform
  Matrix4 inverse_transpose = Matrix4::identity(); // _inverse_transform_transpose
  Material material;
  bool casts_shadow = true;
  ShapeKind kind = ShapeKind::Sphere;

  // cylinder / cone (cylinder.zig:26-28)
  double ymin = -kInf, ymax = kInf;
  bool closed = false;
  // triangle / smooth triangle (triangle.zig:21-26, 214-221)
  Tuple p1, p2, p3, e1, e2, normal, n1, n2, n3;
  // group (group.zig:21-23); a csg keeps {left, right} here (csg.zig:28-31)
  std::vector<Shape> children;
  BoundingBox bbox;
  CsgOp csg_op = CsgOp::Union;

  static Shape make(ShapeKind k) {
    Shape s;
    s.id = nextShapeId();
    s.kind = k;
    return s;
  }
  static Shape sphere() { return make(ShapeKind::Sphere); }
  static Shape glassSphere() {  // shape.zig:158-163
    Shape s = sphere();
    s.material.transparency = 1.0;
    s.material.refractive_index = 1.5;
    return s;
  }
  static Shape plane() { return make(ShapeKind::Plane); }
  static Shape cube() { return make(ShapeKind::Cube); }
  static Shape cylinder() { return make(ShapeKind::Cylinder); }
  static Shape cone() { return make(ShapeKind::Cone); }
  static Shape testShape() { return make(ShapeKind::TestShape); }
  static Shape triangle(const Tuple& p1, const Tuple& p2, const Tuple& p3) {  // shape.zig:186-204
    Shape s = make(ShapeKind::Triangle);
    s.p1 = p1; s.p2 = p2; s.p3 = p3;
    s.e1 = p2.sub(p1);
    s.e2 = p3.sub(p1);
    s.normal = s.e2.cross(s.e1).normalized();
    return s;
  }
  static Shape smoothTriangle(const Tuple& p1, const Tuple& p2, const Tuple& p3,
                              const Tuple& n1, const Tuple& n2, const Tuple& n3) {  // shape.zig:207-227
    Shape s = make(ShapeKind::SmoothTriangle);
    s.p1 = p1; s.p2 = p2; s.p3 = p3;
    s.e1 = p2.sub(p1);
    s.e2 = p3.sub(p1);
    s.n1 = n1; s.n2 = n2; s.n3 = n3;
    return s;
  }
  static Shape group() {  // shape.zig:235-250 — the bbox is itself a Shape and consumes an id
    Shape s;
    (void)nextShapeId();  // bbox.* = Shape(T).boundingBox() is created first
    s.id = nextShapeId();
    s.kind = ShapeKind::Group;
    return s;
  }

  // shape.zig:253-283.  The box is the union of the children's parent-space boxes AT CONSTRUCTION; unlike a
  // group's it is never re-boxed when a transform is pushed through the csg later (shape.zig:298-302).
  static Shape csg(Shape left, Shape right, CsgOp op) {
    Shape s = make(ShapeKind::Csg);
    s.csg_op = op;
    s.bbox = left.parentSpaceBounds();
    s.bbox.merge(right.parentSpaceBounds());
    s.children.push_back(std::move(left));
    s.children.push_back(std::move(right));
    return s;
  }

  bool isGroup() const { return kind == ShapeKind::Group; }
  bool isCsg() const { return kind == ShapeKind::Csg; }

  // shape.zig:353-362 + per-kind bounds(): object-space box.
  BoundingBox bounds() const;
  // shape.zig:364-370
  BoundingBox parentSpaceBounds() const { return bounds().transform(transform); }
  // shape.zig:286-310
  void setTransform(const Matrix4& m);
  // group.zig:75-78
  void addChild(Shape&& child);  // (a Shape is a kilobyte: taken by reference, moved once)
  void addChild(const Shape& child) { addChild(Shape(child)); }
  // group.zig:85-115: returns {left,right}; children keeps the straddlers.
  std::pair<std::vector<Shape>, std::vector<Shape>> partitionChildren();
  // group.zig:117-135
  void makeSubgroup(std::vector<Shape> list);
  // shape.zig:372-399
  void divide(size_t threshold);

  size_t leafCount() const;
};

struct World {  // world.zig:24-25
  std::vector<Shape> objects;
  std::vector<Light> lights;
  static World defaultWorld();  // world.zig:40-62
};

struct Camera {  // camera.zig:18-61
  size_t hsize = 0, vsize = 0;
  double fov = 0, half_width = 0, half_height = 0, pixel_size = 0;
  Tuple saved_from, saved_to, saved_up;
  Matrix4 transform = Matrix4::identity();
  Matrix4 inverse = Matrix4::identity();

  static Camera create(size_t hsize, size_t vsize, double fov);  // Camera.new
  void setTransform(const Matrix4& m) {
    transform = m;
    inverse = m.inverse();
  }
};

}  // namespace rtc
