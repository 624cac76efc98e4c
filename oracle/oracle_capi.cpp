// oracle_capi.cpp — CPU ORACLE entry points (liboracle.so).  TEST INFRASTRUCTURE.
//
// orc_render has the same inputs as rtc_render (include/rtc.h): it rebuilds the
// reference's Shape tree from the flat scene description — one Group per node,
// one leaf Shape per leaf, each with the inverse matrices and boxes exactly as
// given — and then runs the literal restatement in rtc_oracle.hpp with the
// reference's execution model: one job per image row on a thread pool
// (camera.zig:88-97), colorAt(rayForPixel(x, y), depth) per pixel.
// orc_built_* (further down) build the World from the scene JSON with the oracle's
// OWN restatement of scene.zig / obj.zig (rtc_oracle_scene.hpp) instead, so that
// the product loader can be checked against it table by table.
#include <atomic>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <thread>

#include "../include/rtc.h"
#include "rtc_oracle.hpp"
#include "rtc_oracle_scene.hpp"

namespace {

thread_local std::string g_error;

struct OracleScene {
  std::vector<orc::Pattern> patterns;  // stable addresses: sized once
  std::vector<orc::TextureMap> texmaps;
  std::vector<orc::UvImage> images;
  orc::World world;
};

orc::Matrix matrixFrom(const double* p) {
  orc::Matrix m;
  std::memcpy(m.d, p, sizeof(m.d));
  return m;
}

orc::Shape buildRef(const rtc_scene_desc& d, const OracleScene& os, uint32_t ref);

orc::Shape buildLeaf(const rtc_scene_desc& d, const OracleScene& os, uint32_t leaf) {
  if (leaf >= d.n_leaves) throw std::runtime_error("BadIndex: leaf");
  orc::Shape s = orc::Shape::make(static_cast<orc::ShapeKind>(d.leaf_kind[leaf]));
  s.id = d.leaf_id[leaf];
  const uint32_t xf = d.leaf_xform[leaf];
  if (xf >= d.n_xforms) throw std::runtime_error("BadIndex: xform");
  s.inverse = matrixFrom(d.xf_inv + 16 * xf);
  s.inverse_transpose = matrixFrom(d.xf_inv_t + 16 * xf);
  s.casts_shadow = d.leaf_shadow[leaf] != 0;
  const uint32_t mi = d.leaf_material[leaf];
  if (mi >= d.n_materials) throw std::runtime_error("BadIndex: material");
  const double* mp = d.mat_params + RTC_MAT_STRIDE * mi;
  s.material.ambient = mp[0];
  s.material.diffuse = mp[1];
  s.material.specular = mp[2];
  s.material.shininess = mp[3];
  s.material.reflective = mp[4];
  s.material.transparency = mp[5];
  s.material.refractive_index = mp[6];
  if (d.mat_pattern[mi] >= d.n_patterns) throw std::runtime_error("BadIndex: pattern");
  s.material.pattern = os.patterns[d.mat_pattern[mi]];
  const uint32_t g = d.leaf_geom[leaf];
  switch (s.kind) {
    case orc::CYLINDER:
    case orc::CONE:
      if (g >= d.n_cyls) throw std::runtime_error("BadIndex: cyl");
      s.ymin = d.cyl_min[g];
      s.ymax = d.cyl_max[g];
      s.closed = d.cyl_closed[g] != 0;
      break;
    case orc::TRIANGLE:
    case orc::SMOOTH_TRIANGLE: {
      if (g >= d.n_tris) throw std::runtime_error("BadIndex: tri");
      auto v3 = [&](const double* base) { return orc::vec3(base[3 * g], base[3 * g + 1], base[3 * g + 2]); };
      const double* p = d.tri_p1 + 3 * g;
      s.p1 = orc::point(p[0], p[1], p[2]);
      s.e1 = v3(d.tri_e1);
      s.e2 = v3(d.tri_e2);
      if (s.kind == orc::TRIANGLE) {
        s.normal = v3(d.tri_n1);
      } else {
        s.n1 = v3(d.tri_n1);
        s.n2 = v3(d.tri_n2);
        s.n3 = v3(d.tri_n3);
      }
      break;
    }
    case orc::SPHERE:
    case orc::PLANE:
    case orc::CUBE: break;
    default: throw std::runtime_error("Unsupported: leaf kind");
  }
  return s;
}

orc::Shape buildNode(const rtc_scene_desc& d, const OracleScene& os, uint32_t node) {
  if (node >= d.n_nodes) throw std::runtime_error("BadIndex: node");
  orc::Shape g = orc::Shape::make(orc::GROUP);
  g.bmin = orc::point(d.node_min[3 * node], d.node_min[3 * node + 1], d.node_min[3 * node + 2]);
  g.bmax = orc::point(d.node_max[3 * node], d.node_max[3 * node + 1], d.node_max[3 * node + 2]);
  const uint32_t first = d.node_first[node], count = d.node_count[node];
  if (static_cast<uint64_t>(first) + count > d.n_children) throw std::runtime_error("BadIndex: children");
  g.children.reserve(count);
  for (uint32_t i = 0; i < count; ++i) g.children.push_back(buildRef(d, os, d.children[first + i]));
  if (d.node_op && d.node_op[node] != RTC_CSG_NONE) {
    if (count != 2 || d.node_op[node] > RTC_CSG_DIFFERENCE) throw std::runtime_error("InvalidArgument: csg node");
    g.kind = orc::CSG;
    g.csg_op = static_cast<orc::CsgOp>(d.node_op[node]);
  }
  return g;
}

orc::Shape buildRef(const rtc_scene_desc& d, const OracleScene& os, uint32_t ref) {
  if (ref & RTC_CHILD_NODE_BIT) return buildNode(d, os, ref & ~RTC_CHILD_NODE_BIT);
  return buildLeaf(d, os, ref);
}

OracleScene* buildScene(const rtc_scene_desc& d) {
  auto os = std::make_unique<OracleScene>();
  os->patterns.resize(d.n_patterns);
  for (uint32_t i = 0; i < d.n_patterns; ++i) {
    orc::Pattern& p = os->patterns[i];
    p.kind = static_cast<orc::PatternKind>(d.pat_kind[i]);
    p.inverse = matrixFrom(d.pat_inv + 16 * i);
    p.rgb = {d.pat_rgb[3 * i], d.pat_rgb[3 * i + 1], d.pat_rgb[3 * i + 2]};
    if (d.pat_a[i] >= d.n_patterns || d.pat_b[i] >= d.n_patterns) throw std::runtime_error("BadIndex: sub-pattern");
    p.a = &os->patterns[d.pat_a[i]];
    p.b = &os->patterns[d.pat_b[i]];
  }
  // texture maps (rtc.h tex_*, uv_*, img_*)
  os->images.resize(d.n_images);
  for (uint32_t i = 0; i < d.n_images; ++i) {
    orc::UvImage& im = os->images[i];
    im.width = d.img_width[i];
    im.height = d.img_height[i];
    if (im.width == 0 || im.height == 0) throw std::runtime_error("InvalidArgument: empty image");
    const float* src = d.img_rgb + 3 * d.img_offset[i];
    im.rgb.assign(src, src + 3 * im.width * im.height);  // canvas.zig:41 widens zigimg's f32 colour to T
  }
  os->texmaps.resize(d.n_texmaps);
  for (uint32_t i = 0; i < d.n_texmaps; ++i) {
    orc::TextureMap& tm = os->texmaps[i];
    if (d.tex_mapping[i] > RTC_TEX_CUBIC) throw std::runtime_error("Unsupported: texture mapping");
    tm.mapping = static_cast<orc::TexMapping>(d.tex_mapping[i]);
    for (int f = 0; f < 6; ++f) {
      const uint32_t u = d.tex_uv[6 * i + f];
      if (u >= d.n_uvs) throw std::runtime_error("BadIndex: uv pattern");
      orc::UvPattern& uv = tm.faces[f];
      if (d.uv_kind[u] > RTC_UV_TEST) throw std::runtime_error("Unsupported: uv pattern kind");
      uv.kind = static_cast<orc::UvKind>(d.uv_kind[u]);
      uv.width = d.uv_size[2 * u];
      uv.height = d.uv_size[2 * u + 1];
      for (int k = 0; k < 5; ++k) {
        if (d.uv_sub[5 * u + k] >= d.n_patterns) throw std::runtime_error("BadIndex: uv sub-pattern");
        uv.sub[k] = &os->patterns[d.uv_sub[5 * u + k]];
      }
      if (uv.kind == orc::UV_IMAGE) {
        if (d.uv_image[u] >= d.n_images) throw std::runtime_error("BadIndex: image");
        uv.image = &os->images[d.uv_image[u]];
      }
      uv.bilinear = d.uv_interp[u] != 0;
    }
  }
  for (uint32_t i = 0; i < d.n_patterns; ++i) {
    if (os->patterns[i].kind != orc::PAT_TEXTURE_MAP) continue;
    if (d.pat_a[i] >= d.n_texmaps) throw std::runtime_error("BadIndex: texture map");
    os->patterns[i].texture_map = &os->texmaps[d.pat_a[i]];
  }
  for (uint32_t i = 0; i < d.n_roots; ++i) os->world.objects.push_back(buildRef(d, *os, d.roots[i]));
  for (uint32_t i = 0; i < d.n_lights; ++i) {
    os->world.lights.push_back({orc::point(d.light_pos[3 * i], d.light_pos[3 * i + 1], d.light_pos[3 * i + 2]),
                                {d.light_rgb[3 * i], d.light_rgb[3 * i + 1], d.light_rgb[3 * i + 2]}});
  }
  return os.release();
}

orc::Camera cameraFrom(const rtc_camera& c) {
  orc::Camera cam;
  cam.hsize = c.hsize;
  cam.vsize = c.vsize;
  cam.fov = 0;
  cam.half_width = c.half_width;
  cam.half_height = c.half_height;
  cam.pixel_size = c.pixel_size;
  cam.inverse = matrixFrom(c.inv_view);
  return cam;
}

// counters_out (may be null): [primary, secondary, shadow, bbox_tests, tri_tests, smooth_hits, xforms, leaf_tests]
// row_step > 1 renders only rows y0, y0+row_step, ... (bounded-sample timing); other rows are left untouched.
int renderWorld(const orc::World& world, const orc::Camera& camera, uint32_t max_depth, uint32_t x0, uint32_t y0, uint32_t w,
                uint32_t h, uint32_t row_step, uint32_t n_threads, double* rgb_out, uint64_t* counters_out) {
  try {
    if (row_step == 0) row_step = 1;
    if (n_threads == 0) n_threads = std::max(1u, std::thread::hardware_concurrency());
    std::atomic<uint32_t> next_row{0};
    std::vector<orc::Counters> per_thread(n_threads);
    std::string error;
    std::atomic<bool> failed{false};
    auto worker = [&](uint32_t tid) {
      orc::counters() = orc::Counters{};
      try {
        while (true) {
          const uint32_t r = next_row.fetch_add(row_step);  // one job per row, camera.zig:91
          if (r >= h || failed.load()) break;
          const uint32_t y = y0 + r;
          for (uint32_t i = 0; i < w; ++i) {
            const uint32_t x = x0 + i;
            orc::counters().primary++;
            const orc::Ray ray = camera.rayForPixel(x, y);
            const orc::Color c = world.colorAt(ray, max_depth);
            double* px = rgb_out + 3 * (static_cast<size_t>(r) * w + i);
            px[0] = c.r;
            px[1] = c.g;
            px[2] = c.b;
            orc::Arena::mine().reset();  // arena.reset(.retain_capacity) after every pixel, camera.zig:119
          }
        }
      } catch (const std::exception& e) {
        if (!failed.exchange(true)) error = e.what();
      }
      per_thread[tid] = orc::counters();
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < n_threads; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto& t : pool) t.join();
    if (failed.load()) {
      g_error = error;
      return 1;
    }
    if (counters_out) {
      orc::Counters total;
      for (const auto& c : per_thread) total.add(c);
      counters_out[0] = total.primary;
      counters_out[1] = total.secondary;
      counters_out[2] = total.shadow;
      counters_out[3] = total.bbox_tests;
      counters_out[4] = total.tri_tests;
      counters_out[5] = total.smooth_hits;
      counters_out[6] = total.xforms;
      counters_out[7] = total.leaf_tests;
    }
    return 0;
  } catch (const std::exception& e) {
    g_error = e.what();
    return 1;
  }
}



// ---------------------------------------------------------------------------------------------------------------
// The oracle's OWN scene build (rtc_oracle_scene.hpp: scene.zig + obj.zig restated): nothing here sees the product
// loader's description.  orc_built_* hand the built tree out in a canonical order - leaves and nodes numbered by the
// depth-first walk of World.objects - for tests/test_oracle_scene_cpu.py to set beside the product's tables.
// ---------------------------------------------------------------------------------------------------------------
struct BuiltScene {
  std::map<std::string, orc::UvImage> images;  // decoded by the caller (there is no zigimg here)
  std::unique_ptr<orc::scene::Built> built;
  std::vector<const orc::Shape*> leaves, nodes;  // canonical numbering
  std::vector<std::vector<uint32_t>> node_children;
  std::vector<uint32_t> roots;
  std::vector<std::string> blobs;  // distinct serialised materials
  std::vector<uint32_t> leaf_blob;
  std::map<std::string, uint32_t> blob_index;
};

uint32_t crc32Of(const void* data, size_t n) {  // (zlib's CRC-32: the test uses zlib.crc32 on the same bytes)
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    init = true;
  }
  uint32_t c = 0xFFFFFFFFu;
  const unsigned char* p = static_cast<const unsigned char*>(data);
  for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}
template <class T>
void put(std::string& out, const T& v) {
  out.append(reinterpret_cast<const char*>(&v), sizeof(T));
}

// The canonical byte string of a pattern tree (the test builds the same string from the product's pattern tables):
// kind u8, inverse 16 f64, then by kind - solid: rgb; perturb: PerturbInfo as 3 f64, the wrapped pattern; texture map:
// mapping u8, per face (6 for cubic, else 1) uv kind u8 + [align check: 5 patterns | checkers: width, height, 2 patterns
// | image: width u32, height u32, bilinear u8, CRC-32 of the f64 pixels]; test: nothing; every other kind: a, b.
void serialisePattern(std::string& out, const orc::Pattern& p) {
  put<uint8_t>(out, static_cast<uint8_t>(p.kind));
  out.append(reinterpret_cast<const char*>(p.inverse.d), sizeof(p.inverse.d));
  switch (p.kind) {
    case orc::PAT_SOLID:
      put(out, p.rgb.r);
      put(out, p.rgb.g);
      put(out, p.rgb.b);
      break;
    case orc::PAT_TEST: break;
    case orc::PAT_PERTURB:
      put(out, p.rgb.r);
      put(out, p.rgb.g);
      put(out, p.rgb.b);
      serialisePattern(out, *p.a);
      break;
    case orc::PAT_TEXTURE_MAP: {
      const orc::TextureMap& tm = *p.texture_map;
      put<uint8_t>(out, static_cast<uint8_t>(tm.mapping));
      const int faces = tm.mapping == orc::TEX_CUBIC ? 6 : 1;
      for (int f = 0; f < faces; ++f) {
        const orc::UvPattern& uv = tm.faces[f];
        put<uint8_t>(out, static_cast<uint8_t>(uv.kind));
        if (uv.kind == orc::UV_ALIGN_CHECK) {
          for (int k = 0; k < 5; ++k) serialisePattern(out, *uv.sub[k]);
        } else if (uv.kind == orc::UV_CHECKERS) {
          put(out, uv.width);
          put(out, uv.height);
          serialisePattern(out, *uv.sub[0]);
          serialisePattern(out, *uv.sub[1]);
        } else if (uv.kind == orc::UV_IMAGE) {
          put<uint32_t>(out, static_cast<uint32_t>(uv.image->width));
          put<uint32_t>(out, static_cast<uint32_t>(uv.image->height));
          put<uint8_t>(out, uv.bilinear ? 1 : 0);
          put<uint32_t>(out, crc32Of(uv.image->rgb.data(), uv.image->rgb.size() * sizeof(double)));
        }
      }
      break;
    }
    default:  // stripes, rings, gradient, radial gradient, checkers, blend
      serialisePattern(out, *p.a);
      serialisePattern(out, *p.b);
  }
}
std::string serialiseMaterial(const orc::Material& m) {
  std::string out;
  put(out, m.ambient);
  put(out, m.diffuse);
  put(out, m.specular);
  put(out, m.shininess);
  put(out, m.reflective);
  put(out, m.transparency);
  put(out, m.refractive_index);
  serialisePattern(out, m.pattern);
  return out;
}

uint32_t numberShape(BuiltScene& b, const orc::Shape& s) {
  if (s.kind == orc::GROUP || s.kind == orc::CSG) {
    const uint32_t n = static_cast<uint32_t>(b.nodes.size());
    b.nodes.push_back(&s);
    b.node_children.emplace_back();
    for (const orc::Shape& c : s.children) {
      const uint32_t ref = numberShape(b, c);
      b.node_children[n].push_back(ref);
    }
    return n | RTC_CHILD_NODE_BIT;
  }
  const uint32_t leaf = static_cast<uint32_t>(b.leaves.size());
  b.leaves.push_back(&s);
  const std::string blob = serialiseMaterial(s.material);
  auto it = b.blob_index.find(blob);
  if (it == b.blob_index.end()) {
    it = b.blob_index.emplace(blob, static_cast<uint32_t>(b.blobs.size())).first;
    b.blobs.push_back(blob);
  }
  b.leaf_blob.push_back(it->second);
  return leaf;
}

}  // namespace

extern "C" {

const char* orc_last_error(void) { return g_error.c_str(); }

int orc_scene_create(const rtc_scene_desc* desc, void** out) {
  try {
    *out = buildScene(*desc);
    return 0;
  } catch (const std::exception& e) {
    g_error = e.what();
    return 1;
  }
}
void orc_scene_destroy(void* scene) { delete static_cast<OracleScene*>(scene); }

int orc_render(void* scene, const rtc_camera* cam, uint32_t max_depth, uint32_t x0, uint32_t y0, uint32_t w,
               uint32_t h, uint32_t row_step, uint32_t n_threads, double* rgb_out, uint64_t* counters_out) {
  const OracleScene* os = static_cast<OracleScene*>(scene);
  return renderWorld(os->world, cameraFrom(*cam), max_depth, x0, y0, w, h, row_step, n_threads, rgb_out, counters_out);
}


// ---- the oracle's own scene build (rtc_oracle_scene.hpp) ----
int orc_built_create(void** out) {
  *out = new BuiltScene();
  return 0;
}
void orc_built_destroy(void* b) { delete static_cast<BuiltScene*>(b); }
// A decoded image the scene may name: [h][w][3] f32 as zigimg's colour iterator yields it (canvas.zig:34-46 widens to T).
int orc_built_add_image(void* b_, const char* name, uint32_t w, uint32_t h, const float* rgb) {
  BuiltScene* b = static_cast<BuiltScene*>(b_);
  orc::UvImage im;
  im.width = w;
  im.height = h;
  im.rgb.assign(rgb, rgb + 3ull * w * h);
  b->images[name] = std::move(im);
  return 0;
}
// Parses `scene_json` as scene.zig does; files it names (OBJ) are read from data_dir.  width / height != 0 override the
// camera's.
int orc_built_parse(void* b_, const char* scene_json, const char* data_dir, uint32_t width, uint32_t height) {
  BuiltScene* b = static_cast<BuiltScene*>(b_);
  try {
    orc::scene::Files files;
    const std::string dir = data_dir ? data_dir : "";
    files.load = [dir](const std::string& name) {
      std::ifstream f(dir + name, std::ios::binary);
      if (!f) throw std::runtime_error("FileNotFound: " + dir + name);
      std::ostringstream ss;
      ss << f.rdbuf();
      return ss.str();
    };
    files.image = [b](const std::string& name) {
      auto it = b->images.find(name);
      if (it == b->images.end()) throw std::runtime_error("FileNotFound: image " + name);
      return it->second;
    };
    b->built = orc::scene::buildScene(scene_json, files, width, height);
    b->leaves.clear();
    b->nodes.clear();
    b->node_children.clear();
    b->roots.clear();
    b->blobs.clear();
    b->leaf_blob.clear();
    b->blob_index.clear();
    for (const orc::Shape& s : b->built->world.objects) {
      const uint32_t ref = numberShape(*b, s);
      b->roots.push_back(ref);
    }
    return 0;
  } catch (const std::exception& e) {
    g_error = e.what();
    return 1;
  }
}
// out[8]: leaves, nodes, entries of all child lists, roots, lights, distinct materials, the id counter at the start of
// the parse, OBJ lines ignored
int orc_built_counts(void* b_, uint64_t* out) {
  const BuiltScene* b = static_cast<BuiltScene*>(b_);
  if (!b->built) return 1;
  uint64_t n_children = 0;
  for (const auto& c : b->node_children) n_children += c.size();
  out[0] = b->leaves.size();
  out[1] = b->nodes.size();
  out[2] = n_children;
  out[3] = b->roots.size();
  out[4] = b->built->world.lights.size();
  out[5] = b->blobs.size();
  out[6] = b->built->first_id;
  out[7] = b->built->lines_ignored;
  return 0;
}
// Per leaf, in depth-first order: kind, Shape.id, casts_shadow, _inverse_transform [16], its transpose [16], _transform
// [16], {min, max, closed} of a cylinder / cone, {p1, e1, e2, normal | n1, n2, n3} x 3 of a triangle, material index.
int orc_built_leaves(void* b_, uint8_t* kind, uint64_t* id, uint8_t* shadow, double* inv, double* inv_t, double* xf,
                     double* cyl, double* tri, uint32_t* material) {
  const BuiltScene* b = static_cast<BuiltScene*>(b_);
  for (size_t i = 0; i < b->leaves.size(); ++i) {
    const orc::Shape& s = *b->leaves[i];
    kind[i] = static_cast<uint8_t>(s.kind);
    id[i] = s.id;
    shadow[i] = s.casts_shadow ? 1 : 0;
    std::memcpy(inv + 16 * i, s.inverse.d, sizeof(s.inverse.d));
    std::memcpy(inv_t + 16 * i, s.inverse_transpose.d, sizeof(s.inverse.d));
    std::memcpy(xf + 16 * i, s.transform.d, sizeof(s.inverse.d));
    cyl[3 * i] = s.ymin;
    cyl[3 * i + 1] = s.ymax;
    cyl[3 * i + 2] = s.closed ? 1.0 : 0.0;
    const orc::Tuple n1 = s.kind == orc::TRIANGLE ? s.normal : s.n1;
    const orc::Tuple t[6] = {s.p1, s.e1, s.e2, n1, s.n2, s.n3};
    for (int k = 0; k < 6; ++k) {
      tri[18 * i + 3 * k] = t[k].x;
      tri[18 * i + 3 * k + 1] = t[k].y;
      tri[18 * i + 3 * k + 2] = t[k].z;
    }
    material[i] = b->leaf_blob[i];
  }
  return 0;
}
// Per node (Group or Csg), in depth-first pre-order: _bbox min, max [6]; csg operation (0: a group); number of
// children; all child lists one after the other (RTC_CHILD_NODE_BIT | node, or leaf); World.objects.
int orc_built_nodes(void* b_, double* box, uint8_t* op, uint32_t* count, uint32_t* children, uint32_t* roots) {
  const BuiltScene* b = static_cast<BuiltScene*>(b_);
  size_t at = 0;
  for (size_t i = 0; i < b->nodes.size(); ++i) {
    const orc::Shape& s = *b->nodes[i];
    box[6 * i] = s.bmin.x;
    box[6 * i + 1] = s.bmin.y;
    box[6 * i + 2] = s.bmin.z;
    box[6 * i + 3] = s.bmax.x;
    box[6 * i + 4] = s.bmax.y;
    box[6 * i + 5] = s.bmax.z;
    op[i] = s.kind == orc::CSG ? static_cast<uint8_t>(s.csg_op) : 0;
    count[i] = static_cast<uint32_t>(b->node_children[i].size());
    for (uint32_t c : b->node_children[i]) children[at++] = c;
  }
  for (size_t i = 0; i < b->roots.size(); ++i) roots[i] = b->roots[i];
  return 0;
}
// The serialised material `index` (see serialisePattern): returns its length; copies at most `cap` bytes.
uint64_t orc_built_material(void* b_, uint32_t index, uint8_t* buf, uint64_t cap) {
  const BuiltScene* b = static_cast<BuiltScene*>(b_);
  if (index >= b->blobs.size()) return 0;
  const std::string& blob = b->blobs[index];
  if (buf) std::memcpy(buf, blob.data(), std::min<uint64_t>(cap, blob.size()));
  return blob.size();
}
int orc_built_lights(void* b_, double* pos_rgb) {  // [n][6]
  const BuiltScene* b = static_cast<BuiltScene*>(b_);
  size_t i = 0;
  for (const orc::Light& l : b->built->world.lights) {
    pos_rgb[6 * i] = l.position.x;
    pos_rgb[6 * i + 1] = l.position.y;
    pos_rgb[6 * i + 2] = l.position.z;
    pos_rgb[6 * i + 3] = l.intensity.r;
    pos_rgb[6 * i + 4] = l.intensity.g;
    pos_rgb[6 * i + 5] = l.intensity.b;
    ++i;
  }
  return 0;
}
int orc_built_camera(void* b_, rtc_camera* out) {
  const BuiltScene* b = static_cast<BuiltScene*>(b_);
  const orc::Camera& c = b->built->camera;
  out->hsize = static_cast<uint32_t>(c.hsize);
  out->vsize = static_cast<uint32_t>(c.vsize);
  out->half_width = c.half_width;
  out->half_height = c.half_height;
  out->pixel_size = c.pixel_size;
  std::memcpy(out->inv_view, c.inverse.d, sizeof(c.inverse.d));
  return 0;
}
// colorAt(rayForPixel(x, y), max_depth) over the tile, on the oracle-built World with the oracle-built Camera.
int orc_built_render(void* b_, uint32_t max_depth, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t row_step,
                     uint32_t n_threads, double* rgb_out, uint64_t* counters_out) {
  const BuiltScene* b = static_cast<BuiltScene*>(b_);
  if (!b->built) return 1;
  return renderWorld(b->built->world, b->built->camera, max_depth, x0, y0, w, h, row_step, n_threads, rgb_out, counters_out);
}

}  // extern "C"
