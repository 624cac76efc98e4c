// rtc_scene.cpp — Shape tree construction (transform push-down, AABB re-boxing,
// divide(8) BVH build).  Restates, in order, shapes/shape.zig:286-399,
// shapes/group.zig:75-135 and the per-kind bounds() functions.
#include "rtc_scene.hpp"
#include <cstring>

#include <atomic>
#include <memory>

namespace rtc {

namespace {
// shape.zig:123-130 keeps a non-atomic static; the loader is single-threaded there.
std::atomic<size_t> g_next_id{0};
thread_local size_t* tl_scope_counter = nullptr;
}  // namespace

size_t nextShapeId() {
  if (tl_scope_counter) return (*tl_scope_counter)++;
  return g_next_id.fetch_add(1);
}
ShapeIdScope::ShapeIdScope() : outer_(tl_scope_counter) { tl_scope_counter = &drawn; }
ShapeIdScope::~ShapeIdScope() { tl_scope_counter = outer_; }
size_t reserveShapeIds(size_t count) { return g_next_id.fetch_add(count); }
void offsetShapeIds(Shape& tree, size_t base) {
  tree.id += base;
  for (Shape& child : tree.children) offsetShapeIds(child, base);
}

BoundingBox Shape::bounds() const {
  BoundingBox box;
  switch (kind) {
    case ShapeKind::Sphere:     // sphere.zig:55-63
    case ShapeKind::Cube:       // cube.zig:99-107
    case ShapeKind::TestShape:  // shape.zig:429-437
      box.min = Tuple::point(-1.0, -1.0, -1.0);
      box.max = Tuple::point(1.0, 1.0, 1.0);
      break;
    case ShapeKind::Plane:  // plane.zig:45-53
      box.min = Tuple::point(-kInf, 0.0, -kInf);
      box.max = Tuple::point(kInf, 0.0, kInf);
      break;
    case ShapeKind::Cylinder:  // cylinder.zig:114-120
      box.min = Tuple::point(-1.0, ymin, -1.0);
      box.max = Tuple::point(1.0, ymax, 1.0);
      break;
    case ShapeKind::Cone: {  // cone.zig:134-142
      const double limit = std::fmax(std::fabs(ymin), std::fabs(ymax));
      box.min = Tuple::point(-limit, ymin, -limit);
      box.max = Tuple::point(limit, ymax, limit);
      break;
    }
    case ShapeKind::Triangle:        // triangle.zig:72-79
    case ShapeKind::SmoothTriangle:  // triangle.zig:267-274
      box.add(p1);
      box.add(p2);
      box.add(p3);
      break;
    case ShapeKind::Group:  // group.zig:81-83
    case ShapeKind::Csg:    // csg.zig:106-108
      box = bbox;
      break;
  }
  return box;
}

void Shape::setTransform(const Matrix4& m) {
  if (kind == ShapeKind::Group) {
    // Groups pass the transformation on to their children and re-box themselves
    // with the AABB of their transformed AABB (shape.zig:288-296).
    for (Shape& child : children) child.setTransform(m.mul(child.transform));
    bbox = bbox.transform(m);
    (void)nextShapeId();  // the replacement bbox Shape consumes an id (bounding_box.zig:59)
  } else if (kind == ShapeKind::Csg) {
    // CSG pass the transformation on to their children too, but keep the box they were built with
    // (shape.zig:298-302 has no counterpart of the group's re-boxing): restated as is.
    children[0].setTransform(m.mul(children[0].transform));
    children[1].setTransform(m.mul(children[1].transform));
  } else {
    transform = m;
    // (a group pushes ONE matrix down to all its leaves - 23 490 triangles per dragon, six dragons: the inverse of the
    // matrix inverted last is reused; same bits, 540 000 cofactor inverses less for dragons.json)
    thread_local Matrix4 last_m, last_inverse, last_inverse_transpose;
    thread_local bool have_last = false;
    if (!have_last || std::memcmp(&last_m, &m, sizeof m) != 0) {
      last_inverse = m.inverse();  // throws Error("NotInvertible")
      last_inverse_transpose = last_inverse.transpose();
      last_m = m;
      have_last = true;
    }
    inverse = last_inverse;
    inverse_transpose = last_inverse_transpose;
  }
}

void Shape::addChild(Shape&& child) {
  bbox.merge(child.parentSpaceBounds());
  children.push_back(std::move(child));
}

std::pair<std::vector<Shape>, std::vector<Shape>> Shape::partitionChildren() {
  std::vector<Shape> left, right, keep;
  left.reserve(children.size());   // (a Shape is a kilobyte: no re-allocation moves)
  right.reserve(children.size());
  keep.reserve(children.size());
  const auto halves = bbox.split();
  for (Shape& child : children) {
    const BoundingBox cb = child.parentSpaceBounds();
    if (halves.first.containsBox(cb)) {
      left.push_back(std::move(child));
    } else if (halves.second.containsBox(cb)) {
      right.push_back(std::move(child));
    } else {
      keep.push_back(std::move(child));
    }
  }
  children = std::move(keep);
  return {std::move(left), std::move(right)};
}

void Shape::makeSubgroup(std::vector<Shape> list) {
  Shape sub = Shape::group();
  sub.children.reserve(list.size());
  for (Shape& c : list) sub.addChild(std::move(c));
  addChild(std::move(sub));
}

// divide() of a group, planned before anything moves.  The reference's partitionChildren / makeSubgroup hand every
// child down one level at a time (group.zig:85-135), and at every level partitionChildren asks each child for its
// parentSpaceBounds again; here a Shape is a kilobyte, and dragons.json's 141 000 triangles sink through ~20 levels: 5 GB
// of moves and 3 M box transforms, most of the 2 s its load took.  The plan runs the SAME steps in the SAME order on
// (pointer, box) pairs - a child's parent-space box cannot change while its ancestors are partitioned: the subgroups in
// between have identity transforms and a child is divided only after every partition above it - and the tree is then
// built with one move per child.  Ids are drawn where the reference draws them (two per subgroup: its box, itself).
namespace {

struct DividePlan;
struct DivideItem {
  Shape* child = nullptr;   // a child the group already had ...
  std::unique_ptr<DividePlan> sub;  // ... or a subgroup the division makes
  BoundingBox box;          // parentSpaceBounds()
};
struct DividePlan {
  size_t id = 0;
  BoundingBox bbox;
  std::vector<DivideItem> items;
};

void planDivide(DividePlan& g, size_t threshold) {
  if (g.items.size() >= threshold) {
    std::vector<DivideItem> left, right, keep;  // partitionChildren (group.zig:85-115)
    const auto halves = g.bbox.split();
    for (DivideItem& it : g.items) {
      if (halves.first.containsBox(it.box)) {
        left.push_back(std::move(it));
      } else if (halves.second.containsBox(it.box)) {
        right.push_back(std::move(it));
      } else {
        keep.push_back(std::move(it));
      }
    }
    g.items = std::move(keep);
    auto make_subgroup = [&](std::vector<DivideItem>& list) {  // makeSubgroup (group.zig:117-135)
      if (list.empty()) return;
      auto sub = std::make_unique<DividePlan>();
      (void)nextShapeId();  // Shape::group(): the box's id, then the group's
      sub->id = nextShapeId();
      for (const DivideItem& it : list) sub->bbox.merge(it.box);  // addChild
      sub->items = std::move(list);
      DivideItem item;
      item.box = sub->bbox.transform(Matrix4::identity());  // the subgroup's parentSpaceBounds()
      item.sub = std::move(sub);
      g.bbox.merge(item.box);  // addChild
      g.items.push_back(std::move(item));
    };
    make_subgroup(left);
    make_subgroup(right);
  }
  for (DivideItem& it : g.items) {
    if (it.sub) {
      planDivide(*it.sub, threshold);
    } else {
      it.child->divide(threshold);
    }
  }
}

void buildDivided(Shape& g, DividePlan& plan) {
  g.bbox = plan.bbox;
  g.children.clear();
  g.children.reserve(plan.items.size());
  for (DivideItem& it : plan.items) {
    if (it.sub) {
      Shape sub;  // Shape::group() with the ids planDivide drew
      sub.kind = ShapeKind::Group;
      sub.id = it.sub->id;
      buildDivided(sub, *it.sub);
      g.children.push_back(std::move(sub));
    } else {
      g.children.push_back(std::move(*it.child));
    }
  }
}

}  // namespace

void Shape::divide(size_t threshold) {
  if (kind == ShapeKind::Csg) {  // shape.zig:393-396
    children[0].divide(threshold);
    children[1].divide(threshold);
    return;
  }
  if (kind != ShapeKind::Group) return;
  if (children.size() < threshold) {  // nothing to partition (shape.zig:376-390 goes straight to the children)
    for (Shape& child : children) child.divide(threshold);
    return;
  }
  DividePlan plan;
  plan.bbox = bbox;
  plan.items.resize(children.size());
  for (size_t i = 0; i < children.size(); ++i) {
    plan.items[i].child = &children[i];
    plan.items[i].box = children[i].parentSpaceBounds();
  }
  planDivide(plan, threshold);
  std::vector<Shape> old = std::move(children);  // (the plan points into it)
  children = std::vector<Shape>();
  buildDivided(*this, plan);
}

size_t Shape::leafCount() const {
  if (kind != ShapeKind::Group && kind != ShapeKind::Csg) return 1;
  size_t n = 0;
  for (const Shape& c : children) n += c.leafCount();
  return n;
}

World World::defaultWorld() {
  World w;
  Shape s1 = Shape::sphere();
  s1.material.pattern = Pattern::solid({0.8, 1.0, 0.6});
  s1.material.diffuse = 0.7;
  s1.material.specular = 0.2;
  Shape s2 = Shape::sphere();
  s2.setTransform(Matrix4::identity().scale(0.5, 0.5, 0.5));
  w.objects.push_back(std::move(s1));
  w.objects.push_back(std::move(s2));
  w.lights.push_back({Tuple::point(-10.0, 10.0, -10.0), {1.0, 1.0, 1.0}});
  return w;
}

Camera Camera::create(size_t hsize, size_t vsize, double fov) {  // camera.zig:33-52
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

}  // namespace rtc
