// rtc_device.h — device-side scene layout shared by rtc_kernels.hip (kernels)
// and rtc_capi.hip (upload).  gfx950 only.
//
// HBM layout (everything read-only during a render, uploaded once per scene):
//
//   roots[]      u32      World.objects in order; high bit = group node
//   leaf_meta[]  uint4    {kind | casts_shadow<<8, xform, material, geom} per leaf, leaf index =
//                         position in the reference's depth-first order (the equal-t tie-break)
//   xf[]         12 f64   rows 0..2 of Shape._inverse_transform (row 3 is (0,0,0,1): validated at
//                         create).  The inverse-transpose used by normalToWorld (shape.zig:139) is
//                         the same 3x3 read column-wise, so it is not stored.
//   cyl[]        {min,max,closed}
//   tri[]        9 f64    p1,e1,e2 packed per triangle (AoS: a BVH leaf visit is a per-lane
//                         gather, one 72-byte record beats nine SoA lines)
//   trin[]       9 f64    n1,n2,n3 (flat triangles: n1 = stored face normal)
//   mat[]        8 f64    ambient,diffuse,specular,shininess,reflective,transparency,ior,pattern
//   pat[]        DevPattern, 144 B
//   node_box[]   6 f64    Group._bbox min,max; node_kids[] {first,count}; kids[] u32
//   light[]      6 f64    position, intensity
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RTC_NODE_BIT 0x80000000u
#define RTC_NO_LEAF 0xFFFFFFFFu
#define RTC_NO_ITEM 0xFFFFFFFFu
#define RTC_PACKET_ITEMS 16u
#define RTC_ITEM_MAX_CHUNKS 0x100000u  // chunk index bits of a schedule item

// per-lane traversal stack (node indices) and secondary-ray stack capacities
#ifndef RTC_TRAV_STACK
#define RTC_TRAV_STACK 64
#endif
#ifndef RTC_LDS_TRAV
#define RTC_LDS_TRAV 8  // entries of the BVH walk's stack kept in LDS (the rest in scratch memory)
#endif
#ifndef RTC_MAX_DEPTH
#define RTC_MAX_DEPTH 16
#endif
#define RTC_RAY_STACK (RTC_MAX_DEPTH + 2)

struct DevMaterial {
  double ambient, diffuse, specular, shininess, reflective, transparency, ior;
  uint32_t pattern;
  uint32_t shininess_int;  // shininess as an integer if it is one in 2 .. 2^20, else 0: zig_pow's plain squaring loop (pow_small_int)
};

struct DevCyl {
  double ymin, ymax;
  uint32_t closed;
  uint32_t pad;
};

// One record per World.objects entry, flattened so the wave-uniform root loop needs no dependent
// (pointer-chasing) loads.  Split in two tables, both staged once per work-group into LDS:
//   RootCullPair 32 B  conservative world-space bounding spheres of TWO roots in FP32 (r2 == +inf: no finite bound),
//                      component by component: the two roots are the two lanes of packed-FP32 instructions
//   RootRec     144 B  what the exact test needs
struct RootCullPair {
  typedef float Pair __attribute__((ext_vector_type(2)));
  Pair cx, cy, cz, r2;  // centre rounded to nearest, r2 rounded UP; phase 1 of the root loop is FP32
};
// The same bound as an axis-aligned WORLD-space box (round 5): what the render kernels' root loop tests since the box of an
// axis-aligned cube IS the cube (its bounding sphere lets half of the rays through that miss it), a group's box is far
// tighter than the sphere around it, and the interval a ray spends inside a box also says whether the root lies behind
// the origin or beyond a shadow ray's light.  Two roots per record, axis by axis: [lo_x][hi_x][lo_x] [lo_y][hi_y][lo_y]
// [lo_z][hi_z][lo_z], each a pair of floats (root 0, root 1).  A lane reads the NEAR and the FAR plane of an axis as two
// NEIGHBOURING pairs from the offset its ray's direction sign picks - lo, hi for a ray that travels up the axis, hi, lo
// (the repeated lo) for one that travels down: no selects, no min / max, ONE per-lane address per axis for a whole block of
// records (the rest are immediate offsets), and the two roots are the two halves of packed FP32 FMAs.  Planes rounded
// outward; no finite bound: -3e38 / +3e38 (always kept); table padding (to a multiple of EIGHT roots): +3e38 / -3e38 (never kept).
struct RootBoxPair {  // 80 B
  typedef float Pair __attribute__((ext_vector_type(2)));
  Pair x[3], y[3], z[3];  // lo, hi, lo
  Pair line_only;  // != 0: entries of this root may lie outside its box (a cone in a group): only "the line misses the box" culls it
};
struct alignas(16) RootRec {
  double inv[12];        // rows 0..2 of the leaf's inverse; a group: a copy of the root Bvh8Node of its candidate BVH (80 bytes: a walk's first node comes from this record - in LDS - not from the node table)
  double ymin, ymax;     // cylinder / cone
  uint32_t kind_flags;   // kind | casts_shadow<<8 | closed<<9 | room<<10 | is_group<<15
  uint32_t index;        // leaf index (depth-first) or group node index
  uint32_t material;
  uint32_t geom;         // a group: root node of its candidate BVH
  uint32_t always_first, always_count;  // a group: BvhLeafRecs of the leaves below it that no box bounds (planes, cones): visited before the walk
  uint32_t pad_[2];      // 144-byte stride: per-lane LDS reads of different records spread over the banks
};
#define RTC_ROOT_IS_GROUP 0x8000u
#define RTC_ROOT_IS_CSG 0x4000u  // with IS_GROUP: the root is a csg unit, `index` its node
#define RTC_ROOT_ROOM 0x400u     // a top-level cube with every light of the world inside it: most shadow rays never reach its faces (trace())
// Small-world limits: scenes within all four get their tables staged in LDS (50.7 KB per work-group with the mailbox);
// anything larger runs the same kernel reading the tables from memory.
#ifndef RTC_LDS_ROOTS
#define RTC_LDS_ROOTS 128
#endif
#ifndef RTC_LDS_MATERIALS
#define RTC_LDS_MATERIALS 64
#endif
#ifndef RTC_LDS_PATTERNS
#define RTC_LDS_PATTERNS 48
#endif
// (the three-waves-per-SIMD variant of the simple kernel: 52 KB per work-group with two levels of pending rays.  That
// is the most three work-groups can have: at 53 KB the occupancy query still says three per CU, the hardware runs two
// and the third of the persistent work-groups starts when the others are done - cover 0.55 -> 0.75 ms)
#define RTC_LDS3_ROOTS 32
#define RTC_LDS3_MATERIALS 16
#define RTC_LDS3_PATTERNS 20
#define RTC_LDS3_LIGHTS 8
#define RTC_LDS_LIGHTS 16

struct DevPattern {      // 144 B
  double inv[12];        // rows 0..2 of Pattern._inverse_transform
  double rgb[3];         // solid colour
  uint32_t kind, a, b;   // RTC_PAT_*, sub-pattern indices
  uint32_t pad_[3];
};

// Acceleration structure the kernel actually walks for a World.objects entry that is a Group: a binary
// SAH BVH over the WORLD-space boxes of all leaves under that group, FP32, both child boxes in the
// parent (one 64-byte fetch per step).  It only proposes candidates: a leaf's entries count only if the
// ray also passes the exact reference box test of every reference Group above that leaf
// (Group.localIntersect, group.zig:46-50), which is re-checked on the reference's own boxes.
struct BvhNode {
  float lo0[3], hi0[3];  // child 0
  float lo1[3], hi1[3];  // child 1
  uint32_t c0, c1;       // child: node index, RTC_NODE_BIT | first << 3 | (count - 1) into bvh_leaf[], or RTC_NO_LEAF (none)
  uint32_t pad_[2];
};

// What the kernel walks: the binary SAH tree collapsed to four children per node (half the dependent fetches per ray;
// the walk waits on memory latency, not on box tests).  Child boxes component by component, so that two children are
// the two lanes of packed FP32 instructions.  128 B.
struct Bvh4Node {
  float lo[3][4], hi[3][4];  // [axis][child]; an unused slot has lo = +huge, hi = -huge (never entered)
  uint32_t c[4];             // as BvhNode::c0
  uint32_t pad_[4];
};

// What the kernel walks since round 4 (RTC_BVH8): the binary SAH tree collapsed to EIGHT children per node, the child
// boxes quantised to 8 bits per plane on a grid anchored at the node's own FP32 box (origin + per-axis power-of-two
// step, lower planes rounded down, upper planes up) - 80 bytes, five 16-byte fetches for eight children where the
// four-wide node took seven for four.  Children are not referenced one by one: the inner children of a node are
// consecutive nodes (in slot order) starting at child_base, its leaf children's records consecutive BvhLeafRecs starting
// at leaf_base, so a whole node's worth of pending children is ONE stack entry (base + bit masks), and a child's index
// is base + the number of set mask bits below its slot.  Slots are dealt by octant (slot bit 0 / 1 / 2 set: the child
// lies towards +x / +y / +z of the node's centre), so that `slot XOR the ray's direction signs` is a front-to-back
// order without any sorting.  (The layout follows Ylitie, Karras, Laine, "Efficient Incoherent Ray Traversal on GPUs
// Through Compressed Wide BVHs", 2017, with explicit leaf ranges instead of triangle blocks.)
#ifndef RTC_BVH8
#define RTC_BVH8 1
#endif
struct Bvh8Node {              // 80 B = 5 x 16
  float ox, oy, oz;            // origin of the quantisation grid: the node box's lower corner
  uint8_t ex, ey, ez;          // per axis: the step as an IEEE exponent byte, step = bit_cast<float>(e << 23)
  uint8_t imask;               // bit s: slot s holds an inner node
  uint32_t child_base;         // node index of the first inner child
  uint32_t leaf_base_lmask;    // bits 0..23: first BvhLeafRec of the leaf children; bits 24..31: bit s: slot s holds a leaf range
  uint8_t meta[8];             // per slot with a leaf range: (offset from leaf_base) << 2 | (records - 1); records <= 4, offset <= 28; bit 7: none of the range's shapes casts a shadow (shadow traces skip the slot)
  uint8_t q[48];               // lo_x[8] lo_y[8] lo_z[8] hi_x[8] hi_y[8] hi_z[8]; an empty slot has lo = 255, hi = 0
};

// Everything a visit of one BVH leaf needs, in BVH order: one fetch where leaf index -> leaf_meta -> tri would be three
// dependent ones (the walk waits on memory latency).
struct BvhLeafRec {
  uint32_t leaf;        // depth-first leaf index, or RTC_NODE_BIT | node of a csg unit
  uint32_t kind_flags;  // leaf_meta.x
  uint32_t xform;       // leaf_meta.y
  uint32_t material;    // leaf_meta.z
  uint32_t geom;        // leaf_meta.w
  uint32_t parent;      // leaf_parent[leaf]: where the replay of the reference's box chain starts
  uint32_t pad_[2];
  double tri[9];        // p1, e1, e2 of a (smooth) triangle
};

// One pending secondary ray on a lane's stack (see DevPixelMap::ray_stack): origin, direction, weight,
// remaining depth; exactly one 64-byte line.
struct __attribute__((aligned(64))) PendingRec {
  double v[7];
  uint32_t remaining, pad_;
};

// Texture maps (patterns/texture_map.zig); in memory, not LDS: few scenes have them.
struct DevTexMap {   // 32 B
  uint32_t mapping;  // RTC_TEX_*
  uint32_t uv[6];    // DevUv per face (Cubic.Face order), entry 0 for the other mappings
  uint32_t pad_;
};
struct DevUv {       // 64 B
  double width, height;      // UvCheckers
  uint32_t kind, interp, image, pad_;
  uint32_t sub[5];           // pattern indices
  uint32_t pad2_[3];
};
struct DevImage {    // 16 B
  uint64_t offset;   // first pixel in img_rgb
  uint32_t width, height;
};

// One intersection of a csg unit under evaluation (csg.zig:74-95 builds, sorts and filters such a list);
// the per-lane lists live in DevPixelMap::csg_buf as [wave][entry][lane].
struct __attribute__((aligned(32))) CsgRec {
  double t, u, v;
  uint32_t leaf;
  uint32_t flags;  // bit 0: survives every csg filter above it; bit 1: already handed to the visitor
};
#define RTC_CSG_ENTRIES 32u       // entries of a lane's list a handle starts with (DevScene::csg_entries) ...
#define RTC_CSG_ENTRIES_MAX 1024u // ... and how far the synchronous entry points grow it when a render reports an overflow
#define RTC_PATTERN_STACK 8       // gradient / blend patterns nested inside one another: frames of pattern_tree's stack

struct DevScene {
  const RootRec* __restrict__ root_recs;
  const RootCullPair* __restrict__ root_cull;  // (n_roots + 3) / 4 * 2 pairs, padded with never-kept spheres (rtc_estimate_kernel)
  const RootBoxPair* __restrict__ root_box;    // the same roots' world boxes: what the render kernels' root loop tests
  const float* __restrict__ root_weight;       // per root: what a chunk that looks at it costs (rtc_estimate_kernel), in packer ticks
  const uint32_t* __restrict__ roots;
  const uint4* __restrict__ leaf_meta;
  const double* __restrict__ xf;        // [n_xforms][12]
  const DevCyl* __restrict__ cyl;
  const double* __restrict__ tri;       // [n_tris][9]
  const double* __restrict__ trin;      // [n_tris][9]
  const DevMaterial* __restrict__ mat;
  const DevPattern* __restrict__ pat;
  const Bvh4Node* __restrict__ bvh;     // all groups' BVHs; RootRec::geom = root node of a group's BVH (RTC_BVH8 == 0)
  const Bvh8Node* __restrict__ bvh8;    // ... as eight-wide compressed nodes (RTC_BVH8 == 1)
  uint32_t n_bvh_nodes, n_bvh_leaves;   // sizes of bvh / bvh8 and bvh_leaf (the RTC_PROFILE build checks every reference against them)
  const BvhLeafRec* __restrict__ bvh_leaf; // the leaves referenced by the nodes' leaf ranges
  const uint32_t* __restrict__ leaf_parent;  // reference Group node directly above each leaf (RTC_NO_LEAF: none)
  const uint32_t* __restrict__ node_parent;  // reference Group above each Group node (RTC_NO_LEAF: none)
  // csg (csg.zig): per node  op (RTC_CSG_*, bits 0..1) | slot << 8 (index of a csg node inside its unit, < 32)
  //                | side << 16 (1: this node is the `right` child of its parent) | is_unit << 17;
  // node_range = {first depth-first leaf below the node, count}.  leaf_meta.x bit 10 = the leaf's own side.
  const uint32_t* __restrict__ node_info;
  const uint2* __restrict__ node_range;
  const DevTexMap* __restrict__ tex;
  const DevUv* __restrict__ uv;
  const DevImage* __restrict__ img;
  const float* __restrict__ img_rgb;  // [pixels][3]
  CsgRec* csg_buf;  // per-launch scratch of the csg units' intersection lists; null unless the scene has csg nodes
  uint32_t csg_entries;  // entries per lane in csg_buf
  const double* __restrict__ node_box;  // [n_nodes][6]
  const uint2* __restrict__ node_kids;  // {first, count}
  const uint32_t* __restrict__ kids;
  const double* __restrict__ light;     // [n_lights][6]
  uint32_t n_roots, n_leaves, n_nodes, n_lights, n_materials, n_patterns;
  // root_recs / root_cull / root_box are sorted by kind: [spheres][cubes][every other leaf kind and the groups][planes] (the
  // order inside a kind is World.objects order; nothing depends on the table order, see trace(); the planes, which have no
  // bound, are the tail that phase 1 of the root loop does not read)
  uint32_t n_root_planes, n_root_spheres, n_root_cubes;
  float cull_cmax;  // max over bounded roots of |centre|: scale of the FP32 rounding margin
  float cull_bmax;  // max |coordinate| of any finite root box: scale of the box test's FP32 margin
  float cull_par;   // 1.2e-5 x the largest scale of any cube in the scene: the reference's "parallel" rule (cube.zig:28-35 ignores a
                    // direction component below 1e-5 in object space, so an entry may lie that far - times the ray parameter - outside the cube)
  float bvh_mag;    // max |coordinate| of any finite BVH box: scale of the FP32 traversal margin
  uint32_t chain_nested;  // every reference Group box lies inside its parent's: a ray that passes the innermost passes all
  uint32_t all_solid;     // every pattern of the world is a solid colour: a hit's colour needs no object-space point (cover, dragons, groups)
};

struct DevCamera {
  double half_width, half_height, pixel_size;
  double inv[12];  // rows 0..2 of Camera._inverse_transform
  uint32_t hsize, vsize;
};

// How pixels are enumerated and where results go.  Work is handed out in 8x8-pixel CHUNKS, numbered
// 0..n_chunks-1 and pulled by the persistent waves from DevStats::next_chunk.
//   mode 0: rectangle [x0,x0+w) x [y0,y0+h) -> out[(y-y0)*w + (x-x0)]
//   mode 1: interleaved tiles (multi-GPU): region i is tile first_tile + i*tile_stride of a
//           tile_w x tile_h tiling of the image -> out[(i*tile_h + ly)*tile_w + lx]
//   mode 2: the same with an explicit list: region i is tile tile_list[i] (a rank's share of a cost-balanced split;
//           first_tile carries the generation of the list, so that a new list is a new map for the schedule)
struct DevPixelMap {
  uint32_t mode;
  uint32_t x0, y0, w, h;
  uint32_t tile_w, tile_h, first_tile, tile_stride, n_my_tiles;
  uint32_t tiles_x;             // tiles per image row (mode 1)
  uint32_t chunks_x;            // chunks per row of the rectangle / of one tile
  uint32_t chunks_per_region;   // chunks in the rectangle / in one tile
  uint32_t n_chunks;            // total
  const uint32_t* __restrict__ tile_list;  // mode 2 (device memory, owned by the scene handle)
  uint32_t n_units;             // what the work counter runs over: n_chunks, or the number of packets in `order`
  // Optional schedule: PACKETS of RTC_PACKET_ITEMS items, one packet per pull of the work counter.  An item
  // names a run of pixels of one chunk: chunk | start << 20 | (len - 1) << 26, RTC_NO_ITEM = unused slot.
  // The host packs them so that every packet costs about the same (runs of expensive pixels - ray trees
  // through glass - padded with cheap ones) and hands the most expensive packets out first.
  // Scheduling only: results do not depend on it.
  const uint32_t* __restrict__ order;
  // With a schedule packed ON THE DEVICE (rtc_pack_kernel) the host does not know how many packets it has: the kernel
  // then reads the count from here (DevSchedInfo::n_units of the schedule in use); null: n_units above is exact.
  const uint32_t* __restrict__ n_units_dev;
  // Optional per-pixel cost output (rays traced for the pixel, indexed like the canvas), zeroed before
  // the launch; the host packs the next frames' schedule by it.
  uint32_t* __restrict__ cost;
  // Optional, with `cost`: per PACKET of the schedule in use, the time (s_memtime ticks / 16) the wave that pulled
  // it needed for it.  Costs say how work is distributed inside a chunk; times say how long a chunk really takes.
  uint32_t* __restrict__ packet_time;
  // The lanes' stacks of pending secondary rays: [resident waves][ray_stack_levels][64 lanes] records of
  // 64 bytes, owned by the scene handle and sized for the launch (max_depth + 2 levels).
  PendingRec* __restrict__ ray_stack;
  uint32_t ray_stack_levels;
  uint32_t pull_min_idle;  // a wave pulls its next packet only when at least this many lanes are idle (or none has a ray)
  uint32_t row_packets;    // without a schedule (`order` null): != 0: packet c is ROW c % 8 of chunk c / 8 (n_units = 8 n_chunks), else chunk c whole
};

// Per chunk, what cutting it into runs of pixels needs to know about how its rays are spread (rtc_chunk_cost_kernel, from
// the per-pixel costs of a measured frame; the packer's emit step reads it for the few chunks it cuts).
struct DevChunkShape {
  float rays;     // rays its pixels' trees had, about (cost / 5: closest-hit and containers traces count 2, shadow rays 1); 0: not measured
  float depth;    // the deepest tree among them, in levels (<= max_depth + 1, <= its rays)
  uint32_t parts; // packets of the measured frame's schedule that held pixels of this chunk (rtc_chunk_time_kernel counts them)
  uint8_t q[16];  // q[j]: one past the pixel (row-major in the 8x8) at which the running sum of rays passes j/16 of the total; q[0] = 0
};

// What rtc_pack_kernel leaves beside a schedule it packed.
struct DevSchedInfo {
  uint32_t n_units;      // packets in the schedule (DevPixelMap::n_units_dev points here)
  uint32_t heaviest;     // longest chunk time
  uint32_t pad_[2];
  unsigned long long total;  // sum of the chunk times
};

// Scratch of the packer's launches (rtc_chunk_cost_kernel clears it).
#define RTC_PACK_CLASSES 136  // 4 * log2(1 + t) of a 32-bit time is at most 128
struct DevPackState {
  uint32_t cnt[RTC_PACK_CLASSES];     // chunks per class
  uint32_t cursor[RTC_PACK_CLASSES];  // chunks of the class already placed by the sort
  unsigned long long total;           // sum of the chunk times
  uint32_t heaviest;
  uint32_t parts_cursor;              // packets handed out to the runs of cut chunks (the front of the schedule)
  unsigned long long extra[4];        // time the cuts add (every run pays the depth of its deepest tree again): [i] as found in round i
};

// Zero at the start of every launch.  A scene owns TWO of these and alternates: launch n counts in [n & 1] and clears
// [(n + 1) & 1] for its successor (stream order makes that safe), so no launch needs a memset of its own.
// Every counter sits in a cache line of its own.  Atomics on one line are served one after the other, and a frame ends
// with every wave adding its counts and making its last (failing) pull within a few microseconds of each other: with
// all of it in one 48-byte block - 2048 waves x 4 counters + the pulls - the queue on that line WAS the end of the frame
// (cover.json 1080p 0.70 ms; 0.66 with the work counter by chance in the other half of the line; 0.605 with a line per
// counter; 0.575 with the counts summed per work-group first, a quarter of the atomics).
struct DevStats {
  alignas(128) unsigned int next_chunk;  // work counter of the persistent waves
  alignas(128) unsigned long long primary;
  alignas(128) unsigned long long secondary;
  alignas(128) unsigned long long shadow_calls;
  alignas(128) unsigned long long shadow_traced;
  alignas(128) unsigned long long overflow;  // lanes that ran out of a per-lane stack or csg list
  unsigned int csg_needed;               // ... if a csg list was what ran out: the longest list any lane needed (an upper bound)
  unsigned int stolen;                   // rays handed from one lane to another (diagnostic)
#ifdef RTC_PROFILE          // diagnostic builds only (cleared by a host memset there)
  unsigned long long prof[16];  // wave cycles per section
  unsigned long long prof2[8];  // trace invocations (wave level) and active lanes: closest, shadow, behind
  unsigned long long prof3[21]; // wave cycles in traces by lanes with a ray: [closest, shadow, behind][1-2, 3-4, 5-8, 9-16, 17-32, 33-48, 49-64]
  unsigned long long prof5[8];  // [5], [6]: wave cycles of all group walks at nodes / at leaves (the other slots: free)
  unsigned long long prof6[24]; // group walks by kind of trace [closest, shadow, containers][walks, lanes, node steps, leaf steps, lanes at nodes, lanes at leaves, wave cycles at nodes, at leaves]
  unsigned long long prof4[8];  // group walks: walks of a wave, their lanes, wave steps at nodes, at leaves, lanes at nodes, at leaves (summed over the steps), walks that reach no leaf, their node steps
  unsigned long long prof_t0, prof_t1, prof_busy;  // shortest / longest / summed wave lifetime
  unsigned long long prof_log[4096][4];            // per wave: lifetime, iterations, units, first<<32|last unit
  unsigned long long prof_last[4096][16];          // per wave: prof[] of its last packet only
#endif
};
