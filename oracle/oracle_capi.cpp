// oracle_capi.cpp — CPU ORACLE entry points (liboracle.so).  TEST INFRASTRUCTURE.
//
// orc_render has the same inputs as rtc_render (include/rtc.h): it rebuilds the
// reference's Shape tree from the flat scene description — one Group per node,
// one leaf Shape per leaf, each with the inverse matrices and boxes exactly as
// given — and then runs the literal restatement in rtc_oracle.hpp with the
// reference's execution model: one job per image row on a thread pool
// (camera.zig:88-97), colorAt(rayForPixel(x, y), depth) per pixel.
#include <atomic>
#include <cstring>
#include <string>
#include <thread>

#include "../include/rtc.h"
#include "rtc_oracle.hpp"

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

// counters_out (may be null): [primary, secondary, shadow, bbox_tests, tri_tests, smooth_hits, xforms, leaf_tests]
// row_step > 1 renders only rows y0, y0+row_step, ... (bounded-sample timing); other rows are left untouched.
int orc_render(void* scene, const rtc_camera* cam, uint32_t max_depth, uint32_t x0, uint32_t y0, uint32_t w,
               uint32_t h, uint32_t row_step, uint32_t n_threads, double* rgb_out, uint64_t* counters_out) {
  try {
    const OracleScene* os = static_cast<OracleScene*>(scene);
    const orc::Camera camera = cameraFrom(*cam);
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
            const orc::Color c = os->world.colorAt(ray, max_depth);
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

}  // extern "C"
