// rtc_bounds.h — conservative world-space bounds of leaves and roots, and the builder of the candidate BVH
// (DESIGN.md section 3): everything here only ever REMOVES work that provably contributes no entry.
#pragma once
#include <thread>
#include "rtc_host_internal.h"

namespace {


// ---- conservative world-space bounding spheres for the root-loop rejection test ----------------
// These only ever REMOVE work whose result is provably "no entry"; they are computed in plain double
// arithmetic with an inflated radius, never feed a colour, and so need not follow reference rounding.
// Forward transform (object -> world) = inverse of the stored affine inverse; false if singular.
bool forwardOf(const double* inv16, double M[12]) {
  const double a = inv16[0], b = inv16[1], c = inv16[2], d = inv16[4], e = inv16[5], f = inv16[6], g = inv16[8],
               h = inv16[9], i = inv16[10];
  const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
  if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
  const double r[9] = {(e * i - f * h) / det, (c * h - b * i) / det, (b * f - c * e) / det,
                       (f * g - d * i) / det, (a * i - c * g) / det, (c * d - a * f) / det,
                       (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det};
  const double tx = inv16[3], ty = inv16[7], tz = inv16[11];
  for (int k = 0; k < 3; ++k) {
    M[4 * k + 0] = r[3 * k + 0];
    M[4 * k + 1] = r[3 * k + 1];
    M[4 * k + 2] = r[3 * k + 2];
    M[4 * k + 3] = -(r[3 * k + 0] * tx + r[3 * k + 1] * ty + r[3 * k + 2] * tz);
  }
  for (int k = 0; k < 12; ++k)
    if (!std::isfinite(M[k])) return false;
  return true;
}

Sphere sphereOfPoints(const double (*pts)[3], int n) {
  Sphere s;
  double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) {
      mn[k] = std::fmin(mn[k], pts[i][k]);
      mx[k] = std::fmax(mx[k], pts[i][k]);
    }
  s.cx = 0.5 * (mn[0] + mx[0]);
  s.cy = 0.5 * (mn[1] + mx[1]);
  s.cz = 0.5 * (mn[2] + mx[2]);
  double r2 = 0;
  for (int i = 0; i < n; ++i) {
    const double dx = pts[i][0] - s.cx, dy = pts[i][1] - s.cy, dz = pts[i][2] - s.cz;
    r2 = std::fmax(r2, dx * dx + dy * dy + dz * dz);
  }
  s.r = std::sqrt(r2);
  return s;
}

// Object-space box [lo,hi] pushed through M; any convex shape inside the box is inside the sphere.
Sphere sphereOfBox(const double M[12], const double lo[3], const double hi[3]) {
  for (int k = 0; k < 3; ++k)
    if (!std::isfinite(lo[k]) || !std::isfinite(hi[k])) return Sphere{};
  double pts[8][3];
  for (int c = 0; c < 8; ++c) {
    const double x = (c & 1) ? hi[0] : lo[0], y = (c & 2) ? hi[1] : lo[1], z = (c & 4) ? hi[2] : lo[2];
    for (int k = 0; k < 3; ++k) pts[c][k] = M[4 * k] * x + M[4 * k + 1] * y + M[4 * k + 2] * z + M[4 * k + 3];
  }
  return sphereOfPoints(pts, 8);
}

Sphere leafSphere(const rtc_scene_desc& d, uint32_t leaf) {
  double M[12];
  if (!forwardOf(d.xf_inv + 16ull * d.leaf_xform[leaf], M)) return Sphere{};
  const uint32_t g = d.leaf_geom[leaf];
  switch (d.leaf_kind[leaf]) {
    case RTC_SPHERE: {
      // radius = largest singular value of the 3x3 part = sqrt(largest eigenvalue of A = M3^T M3),
      // from the closed-form (trigonometric) eigenvalues of a symmetric 3x3 matrix, with a margin;
      // the Frobenius norm is an always-valid upper bound and caps it.
      double fro = 0;
      for (int k = 0; k < 3; ++k) fro += M[4 * k] * M[4 * k] + M[4 * k + 1] * M[4 * k + 1] + M[4 * k + 2] * M[4 * k + 2];
      double A[3][3];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] = M[i] * M[j] + M[4 + i] * M[4 + j] + M[8 + i] * M[8 + j];
      double r2 = fro;
      {
        const double p1 = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double q = (A[0][0] + A[1][1] + A[2][2]) / 3.0;
        const double p2 = (A[0][0] - q) * (A[0][0] - q) + (A[1][1] - q) * (A[1][1] - q) + (A[2][2] - q) * (A[2][2] - q) + 2.0 * p1;
        double lmax = q;
        if (p2 > 0.0) {
          const double pp = std::sqrt(p2 / 6.0);
          double B[3][3];
          for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) B[i][j] = (A[i][j] - (i == j ? q : 0.0)) / pp;
          const double detB = B[0][0] * (B[1][1] * B[2][2] - B[1][2] * B[2][1]) - B[0][1] * (B[1][0] * B[2][2] - B[1][2] * B[2][0]) +
                              B[0][2] * (B[1][0] * B[2][1] - B[1][1] * B[2][0]);
          const double rr = std::fmax(-1.0, std::fmin(1.0, detB / 2.0));
          lmax = q + 2.0 * pp * std::cos(std::acos(rr) / 3.0);
        }
        if (std::isfinite(lmax) && lmax > 0.0) r2 = std::fmin(r2, lmax * (1.0 + 1e-6));
      }
      Sphere s;
      s.cx = M[3];
      s.cy = M[7];
      s.cz = M[11];
      s.r = std::sqrt(r2);
      const double lo[3] = {-1, -1, -1}, hi[3] = {1, 1, 1};
      const Sphere box = sphereOfBox(M, lo, hi);  // also valid; keep the tighter one
      return (box.finite() && box.r < s.r) ? box : s;
    }
    case RTC_CUBE: {
      const double lo[3] = {-1, -1, -1}, hi[3] = {1, 1, 1};
      return sphereOfBox(M, lo, hi);
    }
    case RTC_CYLINDER: {
      const double lo[3] = {-1, d.cyl_min[g], -1}, hi[3] = {1, d.cyl_max[g], 1};
      return sphereOfBox(M, lo, hi);
    }
    case RTC_CONE:
      // NOT bounded by its truncated box: for a ray parallel to one of the cone's halves the reference
      // appends the single surface hit t = -c / 2b WITHOUT the min < y < max filter (cone.zig:79-86),
      // so a truncated cone can report an entry anywhere on the infinite double cone.
      return Sphere{};
    case RTC_TRIANGLE:
    case RTC_SMOOTH_TRIANGLE: {
      double pts[3][3];
      for (int v = 0; v < 3; ++v) {
        double p[3];
        for (int k = 0; k < 3; ++k)
          p[k] = d.tri_p1[3ull * g + k] + (v == 1 ? d.tri_e1[3ull * g + k] : 0.0) + (v == 2 ? d.tri_e2[3ull * g + k] : 0.0);
        for (int k = 0; k < 3; ++k) pts[v][k] = M[4 * k] * p[0] + M[4 * k + 1] * p[1] + M[4 * k + 2] * p[2] + M[4 * k + 3];
      }
      return sphereOfPoints(pts, 3);
    }
    default: return Sphere{};  // planes are unbounded
  }
}

Sphere inflate(Sphere s) {
  if (!s.finite()) return Sphere{};
  s.r = s.r * (1.0 + 1e-6) + 1e-9;
  return s;
}


// ---- candidate BVH (BvhNode, rtc_device.h) ------------------------------------------------------
struct Aabb {
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  void add(const double p[3]) {
    for (int k = 0; k < 3; ++k) {
      lo[k] = std::fmin(lo[k], p[k]);
      hi[k] = std::fmax(hi[k], p[k]);
    }
  }
  void merge(const Aabb& o) {
    for (int k = 0; k < 3; ++k) {
      lo[k] = std::fmin(lo[k], o.lo[k]);
      hi[k] = std::fmax(hi[k], o.hi[k]);
    }
  }
  bool finite() const {
    for (int k = 0; k < 3; ++k)
      if (!std::isfinite(lo[k]) || !std::isfinite(hi[k]) || lo[k] > hi[k]) return false;
    return true;
  }
  double area() const {
    const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return 2.0 * (dx * dy + dy * dz + dz * dx);
  }
};

const float kHuge = 3.0e38f;  // "unbounded" in an FP32 box

// World-space box of one leaf: the object-space box of the shape pushed through the forward transform.
Aabb leafWorldBox(const rtc_scene_desc& d, uint32_t leaf) {
  Aabb box;
  double M[12];
  if (!forwardOf(d.xf_inv + 16ull * d.leaf_xform[leaf], M)) return Aabb{};
  auto addObjectBox = [&](const double lo[3], const double hi[3]) {
    for (int k = 0; k < 3; ++k)
      if (!std::isfinite(lo[k]) || !std::isfinite(hi[k])) {
        box = Aabb{};
        return;
      }
    for (int c = 0; c < 8; ++c) {
      const double x = (c & 1) ? hi[0] : lo[0], y = (c & 2) ? hi[1] : lo[1], z = (c & 4) ? hi[2] : lo[2];
      double p[3];
      for (int k = 0; k < 3; ++k) p[k] = M[4 * k] * x + M[4 * k + 1] * y + M[4 * k + 2] * z + M[4 * k + 3];
      box.add(p);
    }
  };
  const uint32_t g = d.leaf_geom[leaf];
  switch (d.leaf_kind[leaf]) {
    case RTC_SPHERE:
    case RTC_CUBE: {
      const double lo[3] = {-1, -1, -1}, hi[3] = {1, 1, 1};
      addObjectBox(lo, hi);
      break;
    }
    case RTC_CYLINDER: {
      const double lo[3] = {-1, d.cyl_min[g], -1}, hi[3] = {1, d.cyl_max[g], 1};
      addObjectBox(lo, hi);
      break;
    }
    case RTC_CONE:
      break;  // unbounded, see leafSphere(): the parallel-ray entry ignores the truncation
    case RTC_TRIANGLE:
    case RTC_SMOOTH_TRIANGLE:
      for (int v = 0; v < 3; ++v) {
        double q[3], p[3];
        for (int k = 0; k < 3; ++k)
          q[k] = d.tri_p1[3ull * g + k] + (v == 1 ? d.tri_e1[3ull * g + k] : 0.0) + (v == 2 ? d.tri_e2[3ull * g + k] : 0.0);
        for (int k = 0; k < 3; ++k) p[k] = M[4 * k] * q[0] + M[4 * k + 1] * q[1] + M[4 * k + 2] * q[2] + M[4 * k + 3];
        box.add(p);
      }
      break;
    default: break;  // planes are unbounded
  }
  return box;
}

#ifndef RTC_COLLAPSE_DP
#define RTC_COLLAPSE_DP 1
#endif
#ifndef RTC_COLLAPSE_PRIM_COST
#define RTC_COLLAPSE_PRIM_COST 4.0  // what testing a leaf record costs, in node steps (measured at 0.3 / 0.5 / 1 / 2 / 4 / 10 / 1000: dragons 4K 1.868 / 1.834 / 1.808 / 1.789 / 1.788 / 1.781 / 1.779 ms, teapot 0.2518 / 0.2500 / 0.2495 / 0.2495 / 0.2493 / 0.2527 / 0.2588, nefertiti 0.496 / 0.487 / 0.481 / 0.477 / 0.476 / 0.476 / 0.475)
#endif
#ifndef RTC_SAH_BINS
#define RTC_SAH_BINS 32
#endif
#ifndef RTC_SAH_SWEEP
#define RTC_SAH_SWEEP 16  // ranges of up to this many leaves are split by the exact (sorted sweep) surface-area heuristic
#endif
struct BvhPrim {
  Aabb box;        // may be non-finite: treated as unbounded
  double c[3];     // centroid (0 for unbounded)
  uint32_t leaf;   // depth-first leaf index
};

struct BvhBuilder {
  std::vector<BvhNode>& nodes;
  std::vector<uint32_t>& leaves;
  uint32_t max_depth = 0;  // deepest path in nodes: what the kernel's traversal stack must hold
  std::vector<BvhPrim> prims;
  float mag = 0.0f;

  static float down(double v) {
    float f = static_cast<float>(v);
    if (static_cast<double>(f) > v) f = std::nextafterf(f, -INFINITY);
    return f;
  }
  static float up(double v) {
    float f = static_cast<float>(v);
    if (static_cast<double>(f) < v) f = std::nextafterf(f, INFINITY);
    return f;
  }
  static void storeBox(const Aabb& b, float lo[3], float hi[3], float& mag) {
    if (!b.finite()) {
      for (int k = 0; k < 3; ++k) {
        lo[k] = -kHuge;
        hi[k] = kHuge;
      }
      return;
    }
    for (int k = 0; k < 3; ++k) {
      // rounded outward plus a relative cushion; the kernel adds the ray-dependent part of the margin
      const double pad = 1e-6 * (std::fabs(b.lo[k]) + std::fabs(b.hi[k])) + 1e-30;
      lo[k] = down(b.lo[k] - pad);
      hi[k] = up(b.hi[k] + pad);
      if (!std::isfinite(lo[k]) || !std::isfinite(hi[k])) {
        lo[k] = -kHuge;
        hi[k] = kHuge;
      } else {
        mag = std::fmax(mag, std::fmax(std::fabs(lo[k]), std::fabs(hi[k])));
      }
    }
  }
  Aabb boundsOf(size_t first, size_t count) const {
    Aabb b;
    bool unbounded = false;
    for (size_t i = first; i < first + count; ++i) {
      if (!prims[i].box.finite()) unbounded = true;
      else b.merge(prims[i].box);
    }
    if (unbounded) {
      for (int k = 0; k < 3; ++k) {
        b.lo[k] = -INFINITY;
        b.hi[k] = INFINITY;
      }
    }
    return b;
  }
  static size_t maxLeaf() {
    // Up to two leaves per BVH leaf.  With the kernel's while-while traversal (all lanes run their exact FP64 leaf
    // tests together), the four-wide nodes and the final schedule, measured at 1 / 2 / 3 leaves per node:
    // dragons 4K 2.78 / 2.72 / 2.73 ms, nefertiti 0.728 / 0.713 / 0.719 ms, teapot 0.382 / 0.357 / 0.368 ms.
    return static_cast<size_t>(std::min(RTC_BVH8 ? 4.0 : 8.0, std::max(1.0, static_cast<double>(rtcOptions().bvh_leaf))));  // (an eight-wide node addresses at most four records per leaf child)
  }
  // Sorts prims[first, first + count) into the two sides of the cheapest of the 45 candidate planes (16-bin SAH on three
  // axes; the median of the longest axis where no plane separates anything) and returns where the second side begins.
  size_t split(size_t first, size_t count) {
#if RTC_SAH_SWEEP
    // Small ranges: the exact surface-area heuristic - every one of the count - 1 positions of the range sorted along each
    // axis (ties by leaf index: the order, and with it the tree, does not depend on where the range came from).
    if (count <= RTC_SAH_SWEEP) {
      bool all_finite = true;
      for (size_t i = first; i < first + count; ++i) all_finite = all_finite && prims[i].box.finite();
      if (all_finite) {
        double best = INFINITY;
        int best_axis = -1;
        size_t best_pos = 0;
        std::vector<uint32_t> order(count);
        std::vector<double> right_area(count);
        for (int ax = 0; ax < 3; ++ax) {
          for (size_t i = 0; i < count; ++i) order[i] = static_cast<uint32_t>(i);
          std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            const BvhPrim &pa = prims[first + a], &pb = prims[first + b];
            return pa.c[ax] < pb.c[ax] || (pa.c[ax] == pb.c[ax] && pa.leaf < pb.leaf);
          });
          Aabb acc;
          for (size_t i = count; i-- > 1;) {
            acc.merge(prims[first + order[i]].box);
            right_area[i] = acc.area();
          }
          acc = Aabb{};
          for (size_t i = 0; i + 1 < count; ++i) {
            acc.merge(prims[first + order[i]].box);
            const double cost = acc.area() * static_cast<double>(i + 1) + right_area[i + 1] * static_cast<double>(count - i - 1);
            if (cost < best) {
              best = cost;
              best_axis = ax;
              best_pos = i + 1;
            }
          }
        }
        if (best_axis >= 0) {
          std::sort(prims.begin() + first, prims.begin() + first + count, [&](const BvhPrim& a, const BvhPrim& b) {
            return a.c[best_axis] < b.c[best_axis] || (a.c[best_axis] == b.c[best_axis] && a.leaf < b.leaf);
          });
          return first + best_pos;
        }
      }
    }
#endif
    // centroid bounds
    double clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = first; i < first + count; ++i)
      for (int k = 0; k < 3; ++k) {
        clo[k] = std::fmin(clo[k], prims[i].c[k]);
        chi[k] = std::fmax(chi[k], prims[i].c[k]);
      }
    // 16-bin SAH, all three axes: the cheapest of the 45 candidate planes
    int axis = 0;
    for (int k = 1; k < 3; ++k)
      if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
    size_t mid = first + count / 2;
    bool split_done = false;
    constexpr int kBins = RTC_SAH_BINS;
    double best = INFINITY;
    int best_axis = -1, best_b = -1;
    auto binOf = [&](const BvhPrim& p, int ax) {
      const double extent = chi[ax] - clo[ax];
      int b = static_cast<int>((p.c[ax] - clo[ax]) / extent * kBins);
      return b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
    };
    const bool one_axis = rtcOptions().bvh_one_axis != 0.0;  // (tuning option: longest axis only)
    for (int ax = 0; ax < 3; ++ax) {
      if (one_axis && ax != axis) continue;
      const double extent = chi[ax] - clo[ax];
      if (!(extent > 0.0) || !std::isfinite(extent)) continue;
      Aabb bin_box[kBins];
      size_t bin_n[kBins] = {};
      for (size_t i = first; i < first + count; ++i) {
        const int b = binOf(prims[i], ax);
        bin_n[b]++;
        if (prims[i].box.finite()) bin_box[b].merge(prims[i].box);
      }
      double right_area[kBins];
      size_t right_n[kBins];
      Aabb acc;
      size_t n = 0;
      for (int b = kBins - 1; b > 0; --b) {
        acc.merge(bin_box[b]);
        n += bin_n[b];
        right_area[b] = acc.finite() ? acc.area() : 0.0;
        right_n[b] = n;
      }
      acc = Aabb{};
      n = 0;
      for (int b = 0; b < kBins - 1; ++b) {
        acc.merge(bin_box[b]);
        n += bin_n[b];
        if (n == 0 || right_n[b + 1] == 0) continue;
        const double cost = (acc.finite() ? acc.area() : 0.0) * n + right_area[b + 1] * right_n[b + 1];
        if (cost < best) {
          best = cost;
          best_axis = ax;
          best_b = b;
        }
      }
    }
    if (best_b >= 0) {
      axis = best_axis;
      auto it = std::partition(prims.begin() + first, prims.begin() + first + count,
                               [&](const BvhPrim& p) { return binOf(p, best_axis) <= best_b; });
      mid = static_cast<size_t>(it - prims.begin());
      split_done = mid > first && mid < first + count;
    }
    if (!split_done) {
      mid = first + count / 2;
      std::nth_element(prims.begin() + first, prims.begin() + mid, prims.begin() + first + count,
                       [&](const BvhPrim& a, const BvhPrim& b) { return a.c[axis] < b.c[axis]; });
    }
    return mid;
  }
  // returns the child reference for prims[first, first+count)
  uint32_t build(size_t first, size_t count, uint32_t depth = 1) {
    max_depth = std::max(max_depth, depth);
    const size_t max_leaf = maxLeaf();
    if (count <= max_leaf) {
      const uint32_t at = static_cast<uint32_t>(leaves.size());
      for (size_t i = first; i < first + count; ++i) leaves.push_back(prims[i].leaf);
      return RTC_NODE_BIT | (at << 3) | static_cast<uint32_t>(count - 1);
    }
    const size_t mid = split(first, count);
    const uint32_t me = static_cast<uint32_t>(nodes.size());
    nodes.emplace_back();
    const Aabb b0 = boundsOf(first, mid - first), b1 = boundsOf(mid, first + count - mid);
    const uint32_t c0 = build(first, mid - first, depth + 1);
    const uint32_t c1 = build(mid, first + count - mid, depth + 1);
    BvhNode& N = nodes[me];
    storeBox(b0, N.lo0, N.hi0, mag);
    storeBox(b1, N.lo1, N.hi1, mag);
    N.c0 = c0;
    N.c1 = c1;
    N.pad_[0] = N.pad_[1] = 0;
    return me;
  }
  // The same tree, built by several threads (one leaf per BVH leaf only - the default): a subtree over `count` leaves
  // then has exactly count - 1 nodes and its leaves are the next `count` entries of the leaf list, so every node and
  // leaf has its place before it exists - the node of prims[first, first + count) at `node_at`, its left subtree behind
  // it, its right subtree behind that, leaf k of the builder at leaf_base + k - and subtrees can be built side by side
  // into a table sized in advance: node for node what build() appends one after the other.  `spawn`: levels at which the
  // left subtree goes to a thread of its own.
  struct Sub {
    uint32_t ref, depth;
    float mag;
  };
  Sub buildFixed(size_t first, size_t count, uint32_t depth, uint32_t node_at, uint32_t leaf_base, int spawn) {
    if (count == 1) {
      leaves[leaf_base + first] = prims[first].leaf;
      return {RTC_NODE_BIT | (static_cast<uint32_t>(leaf_base + first) << 3), depth, 0.0f};
    }
    const size_t mid = split(first, count);
    const size_t lc = mid - first, rc = first + count - mid;
    const Aabb b0 = boundsOf(first, lc), b1 = boundsOf(mid, rc);
    Sub l, r;
    if (spawn > 0 && count >= 4096) {
      std::thread left([&]() { l = buildFixed(first, lc, depth + 1, node_at + 1u, leaf_base, spawn - 1); });
      r = buildFixed(mid, rc, depth + 1, node_at + static_cast<uint32_t>(lc), leaf_base, spawn - 1);
      left.join();
    } else {
      l = buildFixed(first, lc, depth + 1, node_at + 1u, leaf_base, 0);
      r = buildFixed(mid, rc, depth + 1, node_at + static_cast<uint32_t>(lc), leaf_base, 0);
    }
    float m = std::fmax(l.mag, r.mag);
    BvhNode& N = nodes[node_at];
    storeBox(b0, N.lo0, N.hi0, m);
    storeBox(b1, N.lo1, N.hi1, m);
    N.c0 = l.ref;
    N.c1 = r.ref;
    N.pad_[0] = N.pad_[1] = 0;
    return {node_at, std::max(l.depth, r.depth), m};
  }
  uint32_t buildTree(int spawn) {  // prims[0, size): build(), or its several-threads form where that pays
    if (spawn <= 0 || maxLeaf() != 1 || prims.size() < 8192) return build(0, prims.size());
    const uint32_t node_at = static_cast<uint32_t>(nodes.size()), leaf_base = static_cast<uint32_t>(leaves.size());
    nodes.resize(nodes.size() + prims.size() - 1);
    leaves.resize(leaves.size() + prims.size());
    const Sub t = buildFixed(0, prims.size(), 1, node_at, leaf_base, spawn);
    max_depth = std::max(max_depth, t.depth);
    mag = std::fmax(mag, t.mag);
    return t.ref;
  }
  // root node index of a BVH over `items` (always a node, so the kernel can start from a node)
  uint32_t buildRoot(std::vector<BvhPrim> items, int spawn = 0) {
    prims = std::move(items);
    const uint32_t me = static_cast<uint32_t>(nodes.size());
    if (prims.size() > 4) return buildTree(spawn);
    nodes.emplace_back();
    BvhNode N;
    std::memset(&N, 0, sizeof N);
    for (int k = 0; k < 3; ++k) {  // empty boxes: never entered
      N.lo0[k] = N.lo1[k] = kHuge;
      N.hi0[k] = N.hi1[k] = -kHuge;
    }
    N.c0 = N.c1 = RTC_NO_LEAF;  // empty child
    if (!prims.empty()) {
      storeBox(boundsOf(0, prims.size()), N.lo0, N.hi0, mag);
      N.c0 = build(0, prims.size());
    }
    nodes[me] = N;
    return me;
  }
};

// Collapses a binary tree of BvhBuilder into four-wide nodes: the children of a node are its two children, the larger
// (by surface area) inner one of which is replaced by ITS two children until there are four or none is left to open.
// Boxes are copied, never recomputed: whatever the binary tree visits, this one visits.
struct Bvh4Collapse {
  const std::vector<BvhNode>& in;
  std::vector<Bvh4Node>& out;
  struct Kid {
    const float* lo;
    const float* hi;
    uint32_t ref;
  };
  static bool inner(uint32_t ref) { return ref != RTC_NO_LEAF && !(ref & RTC_NODE_BIT); }
  static double area(const Kid& k) {
    const double x = static_cast<double>(k.hi[0]) - k.lo[0], y = static_cast<double>(k.hi[1]) - k.lo[1], z = static_cast<double>(k.hi[2]) - k.lo[2];
    return (x < 0.0 || y < 0.0 || z < 0.0) ? 0.0 : 2.0 * (x * y + y * z + z * x);
  }
  // Returns the index of the four-wide node for binary node `n`; `need` = traversal stack entries its subtree can
  // occupy after it has been popped (the kernel pushes every child it enters and pops the nearest).
  uint32_t convert(uint32_t n, uint32_t& need) {
    std::vector<Kid> kids;
    auto add = [&](const BvhNode& N, int which) {
      const uint32_t ref = which == 0 ? N.c0 : N.c1;
      if (ref != RTC_NO_LEAF) kids.push_back({which == 0 ? N.lo0 : N.lo1, which == 0 ? N.hi0 : N.hi1, ref});
    };
    add(in[n], 0);
    add(in[n], 1);
    while (kids.size() < 4) {
      int open = -1;
      for (size_t i = 0; i < kids.size(); ++i)
        if (inner(kids[i].ref) && (open < 0 || area(kids[i]) > area(kids[open]))) open = static_cast<int>(i);
      if (open < 0) break;
      const BvhNode& C = in[kids[open].ref];
      kids.erase(kids.begin() + open);
      add(C, 0);
      add(C, 1);
    }
    const uint32_t me = static_cast<uint32_t>(out.size());
    out.emplace_back();
    Bvh4Node N;
    std::memset(&N, 0, sizeof N);
    uint32_t deepest = 0;
    for (size_t i = 0; i < 4; ++i) {
      if (i < kids.size()) {
        for (int a = 0; a < 3; ++a) {
          N.lo[a][i] = kids[i].lo[a];
          N.hi[a][i] = kids[i].hi[a];
        }
        if (inner(kids[i].ref)) {
          uint32_t below = 0;
          N.c[i] = convert(kids[i].ref, below);
          deepest = std::max(deepest, below);
        } else {
          N.c[i] = kids[i].ref;
        }
      } else {
        for (int a = 0; a < 3; ++a) {
          N.lo[a][i] = kHuge;
          N.hi[a][i] = -kHuge;
        }
        N.c[i] = RTC_NO_LEAF;
      }
    }
    const uint32_t k = static_cast<uint32_t>(kids.size());
    need = std::max(k, (k ? k - 1u : 0u) + deepest);
    out[me] = N;
    return me;
  }
};

// Collapses a binary tree of BvhBuilder into EIGHT-wide compressed nodes (Bvh8Node, rtc_device.h): the children of a node
// are its two children, the larger (by surface area) inner one of which is replaced by ITS two children until there are
// eight or none is left to open.  Child boxes are the binary tree's FP32 boxes quantised OUTWARD (lower planes down, upper
// planes up) on the node's grid, so whatever the binary tree enters, this one enters.  Inner children become consecutive
// nodes, leaf children consecutive leaf records (out_leaves: the binary builder's leaf list re-ordered), both in slot
// order; slots are dealt by octant around the node's centre (greedy assignment of the best remaining child / slot pair).
// Every box must be finite: leaves without a finite bound stay out of the tree (RootRec::always_*).
struct Bvh8Collapse {
  const std::vector<BvhNode>& in;
  const std::vector<uint32_t>& in_leaves;
  std::vector<Bvh8Node>& out;
  std::vector<uint32_t>& out_leaves;
  const std::vector<uint4>* leaf_meta = nullptr;  // per depth-first leaf: .x bit 8 = the shape casts a shadow (Bvh8Node::meta bit 7)
  uint32_t max_depth = 0;  // deepest node: the walk's stack holds at most one entry per level
  bool ok = true;          // false: a quantised box failed to contain its child (never expected; checked, not assumed)
  struct Kid {
    const float* lo;
    const float* hi;
    uint32_t ref;
  };
  static bool inner(uint32_t ref) { return ref != RTC_NO_LEAF && !(ref & RTC_NODE_BIT); }
  static double area(const Kid& k) {
    const double x = static_cast<double>(k.hi[0]) - k.lo[0], y = static_cast<double>(k.hi[1]) - k.lo[1], z = static_cast<double>(k.hi[2]) - k.lo[2];
    return (x < 0.0 || y < 0.0 || z < 0.0) ? 0.0 : 2.0 * (x * y + y * z + z * x);
  }
  // ---- which descendants of a binary node become the (up to eight) children of its wide node.
  // RTC_COLLAPSE_DP == 0: greedily - the node's two children, the larger (by surface area) inner one of which is replaced
  // by ITS two children until there are eight or none is left to open.
  // RTC_COLLAPSE_DP == 1: the cut that minimises the surface-area cost of the WIDE tree (Ylitie, Karras, Laine 2017,
  // section 3.2), by dynamic programming over the binary tree: F(n, i) = the cheapest way to cover the subtree of n with
  // at most i elements, an element being a wide inner node (its cost T(n) = area(n) + the cheapest cover of its two
  // children with eight elements in all) or a leaf child of up to four records (cost kPrim * area(n) * records);
  // F(n, 1) = min(T, leaf), F(n, i) = min(F(n, i - 1), min over k of F(left, k) + F(right, i - k)).
  struct Cover {
    float F[8];        // F(n, 1 .. 8)
    uint8_t k[8];      // i >= 2: the left child's share of the split that gives F(n, i); 0: F(n, i) = F(n, i - 1)
    uint8_t k_node;    // the left child's share of the eight elements when n becomes a wide inner node
    uint8_t as_leaf;   // F(n, 1) is the leaf form
    uint32_t first_leaf, records;  // the binary leaf list's entries below n (depth-first: contiguous)
  };
  std::vector<Cover> cover;  // per binary node (RTC_COLLAPSE_DP)
  static constexpr double kPrim = RTC_COLLAPSE_PRIM_COST;
  double areaOf(const float* lo, const float* hi) const { return area(Kid{lo, hi, 0u}); }
  double coverCost(uint32_t ref, const float* lo, const float* hi, int i) const {  // F(child, i); a binary leaf reference is a leaf whatever i
    if (ref == RTC_NO_LEAF) return 0.0;
    if (ref & RTC_NODE_BIT) return kPrim * areaOf(lo, hi) * ((ref & 7u) + 1u);
    return cover[ref].F[i - 1];
  }
  void prepareCover(uint32_t n, const float* lo, const float* hi) {  // post-order over the binary tree below n (its own box: lo, hi)
    const BvhNode& N = in[n];
    if (inner(N.c0)) prepareCover(N.c0, N.lo0, N.hi0);
    if (inner(N.c1)) prepareCover(N.c1, N.lo1, N.hi1);
    Cover& C = cover[n];
    auto recordsOf = [&](uint32_t ref) -> uint32_t { return ref == RTC_NO_LEAF ? 0u : ((ref & RTC_NODE_BIT) ? (ref & 7u) + 1u : cover[ref].records); };
    auto firstOf = [&](uint32_t ref) -> uint32_t { return (ref & RTC_NODE_BIT) ? ((ref & ~RTC_NODE_BIT) >> 3) : cover[ref].first_leaf; };
    C.records = recordsOf(N.c0) + recordsOf(N.c1);
    C.first_leaf = N.c0 != RTC_NO_LEAF ? firstOf(N.c0) : (N.c1 != RTC_NO_LEAF ? firstOf(N.c1) : 0u);
    const bool contiguous = N.c0 != RTC_NO_LEAF && N.c1 != RTC_NO_LEAF && firstOf(N.c0) + recordsOf(N.c0) == firstOf(N.c1);
    const double a = areaOf(lo, hi);
    double node_cost = INFINITY;
    C.k_node = 1;
    for (int k = 1; k <= 7; ++k) {
      const double v = coverCost(N.c0, N.lo0, N.hi0, k) + coverCost(N.c1, N.lo1, N.hi1, 8 - k);
      if (v < node_cost) {
        node_cost = v;
        C.k_node = static_cast<uint8_t>(k);
      }
    }
    node_cost += a;
    const double leaf_cost = (contiguous && C.records <= 4u) ? kPrim * a * C.records : INFINITY;
    C.as_leaf = leaf_cost < node_cost;
    C.F[0] = static_cast<float>(std::fmin(node_cost, leaf_cost));
    C.k[0] = 0;
    for (int i = 2; i <= 8; ++i) {
      double best = C.F[i - 2];
      C.k[i - 1] = 0;
      for (int k = 1; k < i; ++k) {
        const double v = coverCost(N.c0, N.lo0, N.hi0, k) + coverCost(N.c1, N.lo1, N.hi1, i - k);
        if (v < best) {
          best = v;
          C.k[i - 1] = static_cast<uint8_t>(k);
        }
      }
      C.F[i - 1] = static_cast<float>(best);
    }
  }
  void expand(uint32_t ref, const float* lo, const float* hi, int i, std::vector<Kid>& kids) const {  // the cover F(ref, i), element by element
    if (ref == RTC_NO_LEAF) return;
    if (ref & RTC_NODE_BIT) {
      kids.push_back({lo, hi, ref});
      return;
    }
    const Cover& C = cover[ref];
    while (i > 1 && C.k[i - 1] == 0) --i;
    if (i == 1) {
      kids.push_back({lo, hi, C.as_leaf ? (RTC_NODE_BIT | (C.first_leaf << 3) | (C.records - 1u)) : ref});
      return;
    }
    const BvhNode& N = in[ref];
    expand(N.c0, N.lo0, N.hi0, C.k[i - 1], kids);
    expand(N.c1, N.lo1, N.hi1, i - C.k[i - 1], kids);
  }
  uint32_t convertRoot(uint32_t n) {
    const uint32_t me = static_cast<uint32_t>(out.size());
    out.emplace_back();
#if RTC_COLLAPSE_DP
    cover.assign(in.size(), Cover{});
    {
      const BvhNode& R = in[n];
      float lo[3], hi[3];  // the root's own box: the union of its children's
      for (int a = 0; a < 3; ++a) {
        lo[a] = std::fmin(R.c0 != RTC_NO_LEAF ? R.lo0[a] : INFINITY, R.c1 != RTC_NO_LEAF ? R.lo1[a] : INFINITY);
        hi[a] = std::fmax(R.c0 != RTC_NO_LEAF ? R.hi0[a] : -INFINITY, R.c1 != RTC_NO_LEAF ? R.hi1[a] : -INFINITY);
      }
      prepareCover(n, lo, hi);
    }
#endif
    fill(me, n, 1);
    return me;
  }
  void fill(uint32_t me, uint32_t n, uint32_t depth) {
    max_depth = std::max(max_depth, depth);
    std::vector<Kid> kids;
#if RTC_COLLAPSE_DP
    {
      const BvhNode& N = in[n];
      expand(N.c0, N.lo0, N.hi0, cover[n].k_node, kids);
      expand(N.c1, N.lo1, N.hi1, 8 - cover[n].k_node, kids);
    }
#else
    auto add = [&](const BvhNode& N, int which) {
      const uint32_t ref = which == 0 ? N.c0 : N.c1;
      if (ref != RTC_NO_LEAF) kids.push_back({which == 0 ? N.lo0 : N.lo1, which == 0 ? N.hi0 : N.hi1, ref});
    };
    add(in[n], 0);
    add(in[n], 1);
    while (kids.size() < 8) {
      int open = -1;
      for (size_t i = 0; i < kids.size(); ++i)
        if (inner(kids[i].ref) && (open < 0 || area(kids[i]) > area(kids[open]))) open = static_cast<int>(i);
      if (open < 0) break;
      const BvhNode& C = in[kids[open].ref];
      kids.erase(kids.begin() + open);
      add(C, 0);
      add(C, 1);
    }
#endif
    Bvh8Node N;
    std::memset(&N, 0, sizeof N);
    for (int a = 0; a < 3; ++a)
      for (int s = 0; s < 8; ++s) {  // empty slots: lo above hi, never entered
        N.q[8 * a + s] = 255;
        N.q[24 + 8 * a + s] = 0;
      }
    const size_t nk = kids.size();
    // the node's grid: origin = lower corner of the union, step per axis = the power of two with 255 steps >= the extent
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (int a = 0; a < 3; ++a) {
      lo[a] = nk ? kids[0].lo[a] : 0.0;
      hi[a] = nk ? kids[0].hi[a] : 0.0;
      for (size_t i = 1; i < nk; ++i) {
        lo[a] = std::fmin(lo[a], kids[i].lo[a]);
        hi[a] = std::fmax(hi[a], kids[i].hi[a]);
      }
    }
    const float origin[3] = {static_cast<float>(lo[0]), static_cast<float>(lo[1]), static_cast<float>(lo[2])};  // (floats already: exact)
    double step[3];
    uint8_t* const ebyte[3] = {&N.ex, &N.ey, &N.ez};
    for (int a = 0; a < 3; ++a) {
      const double extent = hi[a] - static_cast<double>(origin[a]);
      int e = extent > 0.0 ? static_cast<int>(std::ceil(std::log2(extent / 255.0))) : -126;
      e = std::max(-126, std::min(127, e));
      while (e < 127 && std::ldexp(255.0, e) < extent) ++e;
      step[a] = std::ldexp(1.0, e);
      *ebyte[a] = static_cast<uint8_t>(e + 127);
    }
    N.ox = origin[0];
    N.oy = origin[1];
    N.oz = origin[2];
    // slots by octant: cost(child, slot) = sum over axes of +-(child centre - node centre); best remaining pair first
    int slot_of[8];
    {
      double centre[3], cost[8][8];
      for (int a = 0; a < 3; ++a) centre[a] = 0.5 * (lo[a] + hi[a]);
      for (size_t i = 0; i < nk; ++i)
        for (int s = 0; s < 8; ++s) {
          double c = 0.0;
          for (int a = 0; a < 3; ++a) {
            const double off = 0.5 * (static_cast<double>(kids[i].lo[a]) + kids[i].hi[a]) - centre[a];
            c += ((s >> a) & 1) ? off : -off;
          }
          cost[i][s] = c;
        }
      bool kid_done[8] = {}, slot_used[8] = {};
      for (size_t round = 0; round < nk; ++round) {
        int bi = -1, bs = -1;
        for (size_t i = 0; i < nk; ++i)
          for (int s = 0; s < 8; ++s)
            if (!kid_done[i] && !slot_used[s] && (bi < 0 || cost[i][s] > cost[bi][bs])) {
              bi = static_cast<int>(i);
              bs = s;
            }
        kid_done[bi] = true;
        slot_used[bs] = true;
        slot_of[bi] = bs;
      }
    }
    int kid_in[8];
    for (int s = 0; s < 8; ++s) kid_in[s] = -1;
    for (size_t i = 0; i < nk; ++i) kid_in[slot_of[i]] = static_cast<int>(i);
    // quantise; inner children and leaf records in slot order
    uint32_t n_inner = 0, n_recs = 0, lmask = 0;
    const uint32_t leaf_base = static_cast<uint32_t>(out_leaves.size());
    for (int s = 0; s < 8; ++s) {
      if (kid_in[s] < 0) continue;
      const Kid& K = kids[kid_in[s]];
      for (int a = 0; a < 3; ++a) {
        const double o = origin[a];
        double ql = std::floor((static_cast<double>(K.lo[a]) - o) / step[a]);
        double qh = std::ceil((static_cast<double>(K.hi[a]) - o) / step[a]);
        ql = std::fmax(0.0, std::fmin(255.0, ql));
        qh = std::fmax(0.0, std::fmin(255.0, qh));
        if (!(o + ql * step[a] <= K.lo[a] && o + qh * step[a] >= K.hi[a])) ok = false;
        N.q[8 * a + s] = static_cast<uint8_t>(ql);
        N.q[24 + 8 * a + s] = static_cast<uint8_t>(qh);
      }
      if (inner(K.ref)) {
        N.imask |= static_cast<uint8_t>(1u << s);
        n_inner++;
      } else {
        const uint32_t first = (K.ref & ~RTC_NODE_BIT) >> 3, count = (K.ref & 7u) + 1u;
        if (count > 4u || n_recs > 28u) ok = false;
        N.meta[s] = static_cast<uint8_t>((n_recs << 2) | (count - 1u));
        lmask |= 1u << s;
        bool dark = leaf_meta != nullptr;  // no shape of the range casts a shadow (a csg unit: not known here, it counts as casting)
        for (uint32_t k = 0; k < count && dark; ++k) {
          const uint32_t leaf = in_leaves[first + k];
          dark = !(leaf & RTC_NODE_BIT) && leaf < leaf_meta->size() && (((*leaf_meta)[leaf].x >> 8) & 1u) == 0u;
        }
        if (dark) N.meta[s] |= 0x80u;
        for (uint32_t k = 0; k < count; ++k) out_leaves.push_back(in_leaves[first + k]);
        n_recs += count;
      }
    }
    if (leaf_base >= (1u << 24)) ok = false;
    N.leaf_base_lmask = (leaf_base & 0xFFFFFFu) | (lmask << 24);
    const uint32_t child_base = static_cast<uint32_t>(out.size());
    N.child_base = child_base;
    out.resize(out.size() + n_inner);  // (invalidates references into `out`: N is a local copy)
    out[me] = N;
    uint32_t rank = 0;
    for (int s = 0; s < 8; ++s) {
      if (kid_in[s] < 0 || !inner(kids[kid_in[s]].ref)) continue;
      fill(child_base + rank, kids[kid_in[s]].ref, depth + 1);
      rank++;
    }
  }
};

bool affineRow(const double* m16) {  // last row must be exactly (0,0,0,1); -0 is accepted
  return m16[12] == 0.0 && m16[13] == 0.0 && m16[14] == 0.0 && m16[15] == 1.0;
}

}  // namespace
