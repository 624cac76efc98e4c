// rtc_flatten.cpp — see rtc_flatten.hpp.
#include "rtc_flatten.hpp"

#include <cstring>
#include <map>

namespace rtc {

namespace {

void appendBits(std::string& key, const double* v, size_t n) {
  key.append(reinterpret_cast<const char*>(v), n * sizeof(double));
}

struct Flattener {
  FlatScene out;
  std::unordered_map<std::string, uint32_t> xform_ids, material_ids, pattern_ids;

  uint32_t internXform(const Shape& s) {
    std::string key;
    appendBits(key, &s.inverse.d[0][0], 16);
    appendBits(key, &s.inverse_transpose.d[0][0], 16);
    auto it = xform_ids.find(key);
    if (it != xform_ids.end()) return it->second;
    const uint32_t id = static_cast<uint32_t>(out.xf_inv.size() / 16);
    out.xf_inv.insert(out.xf_inv.end(), &s.inverse.d[0][0], &s.inverse.d[0][0] + 16);
    out.xf_inv_t.insert(out.xf_inv_t.end(), &s.inverse_transpose.d[0][0], &s.inverse_transpose.d[0][0] + 16);
    xform_ids.emplace(std::move(key), id);
    return id;
  }

  std::map<const UvImageData*, uint32_t> image_ids;

  uint32_t internImage(const std::shared_ptr<const UvImageData>& im) {
    auto it = image_ids.find(im.get());
    if (it != image_ids.end()) return it->second;
    const uint32_t id = static_cast<uint32_t>(out.img_width.size());
    out.img_width.push_back(static_cast<uint32_t>(im->width));
    out.img_height.push_back(static_cast<uint32_t>(im->height));
    out.img_offset.push_back(out.img_rgb.size() / 3);
    out.img_rgb.insert(out.img_rgb.end(), im->rgb.begin(), im->rgb.end());
    image_ids.emplace(im.get(), id);
    return id;
  }

  uint32_t internUv(const UvPattern& uv) {
    uint32_t sub[5] = {0, 0, 0, 0, 0};
    for (size_t i = 0; i < uv.sub.size() && i < 5; ++i) sub[i] = internPattern(*uv.sub[i]);
    const uint32_t image = uv.kind == UvKind::Image ? internImage(uv.image) : 0u;
    const uint32_t id = static_cast<uint32_t>(out.uv_kind.size());
    out.uv_kind.push_back(static_cast<uint8_t>(uv.kind));
    out.uv_size.push_back(uv.width);
    out.uv_size.push_back(uv.height);
    out.uv_sub.insert(out.uv_sub.end(), sub, sub + 5);
    out.uv_image.push_back(image);
    out.uv_interp.push_back(uv.bilinear ? 1 : 0);
    return id;
  }

  uint32_t internTextureMap(const TextureMap& tm) {
    uint32_t uv[6];
    for (size_t f = 0; f < 6; ++f) uv[f] = f < tm.faces.size() ? internUv(tm.faces[f]) : uv[0];
    const uint32_t id = static_cast<uint32_t>(out.tex_mapping.size());
    out.tex_mapping.push_back(static_cast<uint8_t>(tm.mapping));
    out.tex_uv.insert(out.tex_uv.end(), uv, uv + 6);
    return id;
  }

  uint32_t internPattern(const Pattern& p) {
    uint32_t a = 0, b = 0;
    if (p.a) a = internPattern(*p.a);
    if (p.b) b = internPattern(*p.b);
    if (p.kind == PatternKind::TextureMap) {  // pat_a names the texture map; every one is its own table entry
      a = internTextureMap(*p.texture_map);
      b = 0;
    }
    std::string key;
    key.push_back(static_cast<char>(p.kind));
    appendBits(key, &p.inverse.d[0][0], 16);
    // a perturb pattern has no colour of its own: the slot carries PerturbInfo (rtc.h, pat_rgb)
    const bool perturb = p.kind == PatternKind::Perturb;
    const double rgb[3] = {perturb ? p.perturb_scale : p.rgb.r, perturb ? static_cast<double>(p.perturb_octaves) : p.rgb.g,
                           perturb ? p.perturb_persistence : p.rgb.b};
    if (perturb) b = a;
    appendBits(key, rgb, 3);
    key.append(reinterpret_cast<const char*>(&a), 4);
    key.append(reinterpret_cast<const char*>(&b), 4);
    auto it = pattern_ids.find(key);
    if (it != pattern_ids.end()) return it->second;
    const uint32_t id = static_cast<uint32_t>(out.pat_kind.size());
    out.pat_kind.push_back(static_cast<uint8_t>(p.kind));
    out.pat_inv.insert(out.pat_inv.end(), &p.inverse.d[0][0], &p.inverse.d[0][0] + 16);
    out.pat_rgb.insert(out.pat_rgb.end(), rgb, rgb + 3);
    out.pat_a.push_back(a);
    out.pat_b.push_back(b);
    pattern_ids.emplace(std::move(key), id);
    return id;
  }

  uint32_t internMaterial(const Material& m) {
    const uint32_t pat = internPattern(m.pattern);
    const double params[RTC_MAT_STRIDE] = {m.ambient,    m.diffuse,      m.specular,        m.shininess,
                                           m.reflective, m.transparency, m.refractive_index};
    std::string key;
    appendBits(key, params, RTC_MAT_STRIDE);
    key.append(reinterpret_cast<const char*>(&pat), 4);
    auto it = material_ids.find(key);
    if (it != material_ids.end()) return it->second;
    const uint32_t id = static_cast<uint32_t>(out.mat_pattern.size());
    out.mat_params.insert(out.mat_params.end(), params, params + RTC_MAT_STRIDE);
    out.mat_pattern.push_back(pat);
    material_ids.emplace(std::move(key), id);
    return id;
  }

  static void push3(std::vector<double>& v, const Tuple& t) {
    v.push_back(t.x);
    v.push_back(t.y);
    v.push_back(t.z);
  }

  // Returns the encoded child reference, or UINT32_MAX for shapes that can never be hit.
  uint32_t visit(const Shape& s) {
    if (s.kind == ShapeKind::Group || s.kind == ShapeKind::Csg) return RTC_CHILD_NODE_BIT | visitGroup(s);
    if (s.kind == ShapeKind::TestShape) return UINT32_MAX;  // shape.zig:411-420: no intersections
    const uint32_t leaf = static_cast<uint32_t>(out.leaf_kind.size());
    uint8_t kind = RTC_SPHERE;
    uint32_t geom = 0;
    switch (s.kind) {
      case ShapeKind::Sphere: kind = RTC_SPHERE; break;
      case ShapeKind::Plane: kind = RTC_PLANE; break;
      case ShapeKind::Cube: kind = RTC_CUBE; break;
      case ShapeKind::Cylinder:
      case ShapeKind::Cone:
        kind = s.kind == ShapeKind::Cylinder ? RTC_CYLINDER : RTC_CONE;
        geom = static_cast<uint32_t>(out.cyl_min.size());
        out.cyl_min.push_back(s.ymin);
        out.cyl_max.push_back(s.ymax);
        out.cyl_closed.push_back(s.closed ? 1 : 0);
        break;
      case ShapeKind::Triangle:
      case ShapeKind::SmoothTriangle: {
        const bool smooth = s.kind == ShapeKind::SmoothTriangle;
        kind = smooth ? RTC_SMOOTH_TRIANGLE : RTC_TRIANGLE;
        geom = static_cast<uint32_t>(out.tri_p1.size() / 3);
        push3(out.tri_p1, s.p1);
        push3(out.tri_e1, s.e1);
        push3(out.tri_e2, s.e2);
        push3(out.tri_n1, smooth ? s.n1 : s.normal);
        push3(out.tri_n2, smooth ? s.n2 : Tuple{});
        push3(out.tri_n3, smooth ? s.n3 : Tuple{});
        break;
      }
      default: break;
    }
    out.leaf_kind.push_back(kind);
    out.leaf_xform.push_back(internXform(s));
    out.leaf_material.push_back(internMaterial(s.material));
    out.leaf_shadow.push_back(s.casts_shadow ? 1 : 0);
    out.leaf_id.push_back(static_cast<uint32_t>(s.id));
    out.leaf_geom.push_back(geom);
    return leaf;
  }

  uint32_t visitGroup(const Shape& g) {
    const uint32_t node = static_cast<uint32_t>(out.node_first.size());
    push3(out.node_min, g.bbox.min);
    push3(out.node_max, g.bbox.max);
    std::vector<uint32_t> refs;
    out.node_first.push_back(0);
    out.node_count.push_back(0);
    out.node_op.push_back(g.isCsg() ? static_cast<uint8_t>(g.csg_op) : RTC_CSG_NONE);
    for (const Shape& c : g.children) {
      uint32_t r = visit(c);
      if (r == UINT32_MAX && g.isCsg()) {  // a csg always has a left and a right: an empty group stands in
        r = RTC_CHILD_NODE_BIT | static_cast<uint32_t>(out.node_first.size());
        push3(out.node_min, BoundingBox{}.min);
        push3(out.node_max, BoundingBox{}.max);
        out.node_first.push_back(static_cast<uint32_t>(out.children.size()));
        out.node_count.push_back(0);
        out.node_op.push_back(RTC_CSG_NONE);
      }
      if (r != UINT32_MAX) refs.push_back(r);
    }
    // children of one group are contiguous in children[]; sub-groups were appended first.
    out.node_first[node] = static_cast<uint32_t>(out.children.size());
    out.node_count[node] = static_cast<uint32_t>(refs.size());
    out.children.insert(out.children.end(), refs.begin(), refs.end());
    return node;
  }
};

}  // namespace

FlatScene flattenWorld(const World& world) {
  Flattener f;
  for (const Shape& s : world.objects) {
    const uint32_t r = f.visit(s);
    if (r != UINT32_MAX) f.out.roots.push_back(r);
  }
  for (const Light& l : world.lights) {
    Flattener::push3(f.out.light_pos, l.position);
    f.out.light_rgb.push_back(l.intensity.r);
    f.out.light_rgb.push_back(l.intensity.g);
    f.out.light_rgb.push_back(l.intensity.b);
  }
  return std::move(f.out);
}

rtc_scene_desc FlatScene::desc() const {
  rtc_scene_desc d;
  std::memset(&d, 0, sizeof(d));
  d.abi_version = RTC_ABI_VERSION;
  d.n_xforms = static_cast<uint32_t>(xf_inv.size() / 16);
  d.xf_inv = xf_inv.data();
  d.xf_inv_t = xf_inv_t.data();
  d.n_leaves = static_cast<uint32_t>(leaf_kind.size());
  d.leaf_kind = leaf_kind.data();
  d.leaf_xform = leaf_xform.data();
  d.leaf_material = leaf_material.data();
  d.leaf_shadow = leaf_shadow.data();
  d.leaf_id = leaf_id.data();
  d.leaf_geom = leaf_geom.data();
  d.n_cyls = static_cast<uint32_t>(cyl_min.size());
  d.cyl_min = cyl_min.data();
  d.cyl_max = cyl_max.data();
  d.cyl_closed = cyl_closed.data();
  d.n_tris = static_cast<uint32_t>(tri_p1.size() / 3);
  d.tri_p1 = tri_p1.data();
  d.tri_e1 = tri_e1.data();
  d.tri_e2 = tri_e2.data();
  d.tri_n1 = tri_n1.data();
  d.tri_n2 = tri_n2.data();
  d.tri_n3 = tri_n3.data();
  d.n_materials = static_cast<uint32_t>(mat_pattern.size());
  d.mat_params = mat_params.data();
  d.mat_pattern = mat_pattern.data();
  d.n_patterns = static_cast<uint32_t>(pat_kind.size());
  d.pat_kind = pat_kind.data();
  d.pat_inv = pat_inv.data();
  d.pat_rgb = pat_rgb.data();
  d.pat_a = pat_a.data();
  d.pat_b = pat_b.data();
  d.n_nodes = static_cast<uint32_t>(node_first.size());
  d.node_min = node_min.data();
  d.node_max = node_max.data();
  d.node_first = node_first.data();
  d.node_count = node_count.data();
  d.node_op = node_op.data();
  d.n_children = static_cast<uint32_t>(children.size());
  d.children = children.data();
  d.n_roots = static_cast<uint32_t>(roots.size());
  d.roots = roots.data();
  d.n_lights = static_cast<uint32_t>(light_pos.size() / 3);
  d.light_pos = light_pos.data();
  d.light_rgb = light_rgb.data();
  d.n_texmaps = static_cast<uint32_t>(tex_mapping.size());
  d.tex_mapping = tex_mapping.data();
  d.tex_uv = tex_uv.data();
  d.n_uvs = static_cast<uint32_t>(uv_kind.size());
  d.uv_kind = uv_kind.data();
  d.uv_size = uv_size.data();
  d.uv_sub = uv_sub.data();
  d.uv_image = uv_image.data();
  d.uv_interp = uv_interp.data();
  d.n_images = static_cast<uint32_t>(img_width.size());
  d.img_width = img_width.data();
  d.img_height = img_height.data();
  d.img_offset = img_offset.data();
  d.img_rgb = img_rgb.data();
  return d;
}

rtc_camera flattenCamera(const Camera& c) {
  rtc_camera out;
  out.hsize = static_cast<uint32_t>(c.hsize);
  out.vsize = static_cast<uint32_t>(c.vsize);
  out.half_width = c.half_width;
  out.half_height = c.half_height;
  out.pixel_size = c.pixel_size;
  std::memcpy(out.inv_view, &c.inverse.d[0][0], sizeof(out.inv_view));
  return out;
}

}  // namespace rtc
