// rtc_flatten.hpp — World(T) -> flat SoA tables (include/rtc.h: rtc_scene_desc).
//
// Walks World.objects depth-first in the order World.intersect / Group.localIntersect
// visit them (world.zig:74, group.zig:52), so that leaf index == the position the
// reference's nested stable sorts give an intersection among equal t's.
// One node per reference Group, with the Group's own _bbox: identical boxes give
// identical candidate sets.
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rtc.h"
#include "rtc_scene.hpp"

namespace rtc {

struct FlatScene {
  std::vector<double> xf_inv, xf_inv_t;
  std::vector<uint8_t> leaf_kind, leaf_shadow;
  std::vector<uint32_t> leaf_xform, leaf_material, leaf_id, leaf_geom;
  std::vector<double> cyl_min, cyl_max;
  std::vector<uint8_t> cyl_closed;
  std::vector<double> tri_p1, tri_e1, tri_e2, tri_n1, tri_n2, tri_n3;
  std::vector<double> mat_params;
  std::vector<uint32_t> mat_pattern;
  std::vector<uint8_t> pat_kind;
  std::vector<double> pat_inv, pat_rgb;
  std::vector<uint32_t> pat_a, pat_b;
  std::vector<double> node_min, node_max;
  std::vector<uint32_t> node_first, node_count, children, roots;
  std::vector<uint8_t> node_op;
  // texture maps (rtc.h tex_*, uv_*, img_*)
  std::vector<uint8_t> tex_mapping, uv_kind, uv_interp;
  std::vector<uint32_t> tex_uv, uv_sub, uv_image, img_width, img_height;
  std::vector<double> uv_size;
  std::vector<uint64_t> img_offset;
  std::vector<float> img_rgb;
  std::vector<double> light_pos, light_rgb;

  // View over the vectors above; valid while *this is alive and unmodified.
  rtc_scene_desc desc() const;

  size_t leafCount() const { return leaf_kind.size(); }
  size_t nodeCount() const { return node_first.size(); }
};

FlatScene flattenWorld(const World& world);
rtc_camera flattenCamera(const Camera& camera);

}  // namespace rtc
