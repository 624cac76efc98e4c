/*
 * rtc.h — C ABI of the MI355X render path (librtc_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of SinclaM/ray-tracer-challenge:
 *
 *     Camera(T).render(self, allocator, world) !Canvas(T)      src/raytracer/camera.zig:80-125
 *       -> rayForPixel                                         src/raytracer/camera.zig:64-76
 *       -> World.colorAt(ray, 5)                               src/raytracer/world.zig:111-121
 *            intersect / shadeHit / isShadowed /
 *            reflectedColor / refractedColor                   src/raytracer/world.zig:71-189
 *
 * The reference has no FFI for this path (it is pure Zig); the only C-ABI
 * precedent is the WASM export table src/lib.zig:233-309 (global renderer,
 * error returned as a NUL-terminated error *name*).  A Zig host binds the
 * functions below with `extern fn` declarations and calls
 * them from the body of Camera.render; the JSON scene loader and the
 * canvas/PPM writer stay untouched.
 *
 * Conventions
 *   - all reals are IEEE binary64 (the reference renders scenes in f64:
 *     src/main.zig:71, src/lib.zig:194,223);
 *   - matrices are row-major 4x4 (src/raytracer/matrix.zig:15), 16 doubles;
 *   - all input arrays are caller-owned and are copied by rtc_scene_create();
 *     the library never retains a host pointer;
 *   - every function returns an rtc_status; the message/name of the last
 *     failure on the calling thread is available from rtc_last_error();
 *   - a handle may be used from one thread at a time; distinct handles are
 *     independent;
 *   - renders on ONE handle run one after the other, whatever streams they are
 *     enqueued on: a handle owns one set of counters, work counter and per-lane
 *     scratch, so every launch records an event and a launch on a different
 *     stream makes that stream wait for it first (no host synchronisation).
 *     Two handles on two streams overlap freely.
 *
 * Limits (a scene beyond them is refused or reported, never rendered differently):
 *   - more than 8 gradient / radial-gradient / blend patterns nested inside one
 *     another: rtc_scene_create returns RTC_ERR_UNSUPPORTED;
 *   - pattern select-chains (stripes / checkers / rings / perturb / texture map)
 *     deeper than 64: RTC_ERR_UNSUPPORTED at create;
 *   - more than 64 csg nodes under one csg: RTC_ERR_UNSUPPORTED at create;
 *   - a lane's list of the intersections of one ray with one csg unit (the list
 *     Csg.filterIntersections works on, csg.zig:51-95) starts with 32 slots: when a
 *     frame needs more, the frame itself says how many (the longest list any lane
 *     wanted): rtc_render / rtc_render_rgba8 size the lists for that and render
 *     again (up to 1024; the handle keeps the longer lists), while the asynchronous
 *     entry points count the lanes that ran out in rtc_stats::overflow - callers
 *     check rtc_get_stats and call rtc_grow_csg_lists, then render the frame again
 *     (rtc_multi.h does that for its handles); beyond 1024 entries, or a group tree
 *     deeper than the traversal stack: RTC_ERR_OVERFLOW, never a truncated image;
 *   - two leaves with the same Shape.id: RTC_ERR_UNSUPPORTED (identity in the
 *     containers walk is the leaf);
 *   - reproducibility: geometry and every branch are bit-identical from run to
 *     run; the colour of a pixel whose ray tree was shared between lanes is the
 *     sum of the lanes' shares in completion order (f64 atomic adds), so two
 *     renders of one frame agree to ~1e-15, not bitwise.  A pixel whose tree
 *     stayed in one lane (every pixel of a scene without transparent
 *     reflective materials) is bitwise reproducible.
 */
#ifndef RTC_H
#define RTC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTC_ABI_VERSION 3u

/* ---- status codes; names mirror the reference's Zig error names where one exists ---- */
typedef enum rtc_status {
  RTC_OK = 0,
  RTC_ERR_INVALID_ARGUMENT = 1, /* null pointer, zero-sized image, tile out of range ...           */
  RTC_ERR_OUT_OF_MEMORY = 2,    /* Zig: error.OutOfMemory -> @panic in camera.zig:118               */
  RTC_ERR_NOT_INVERTIBLE = 3,   /* Zig: MatrixError.NotInvertible, matrix.zig:7                     */
  RTC_ERR_UNSUPPORTED = 4,      /* a shape/pattern kind this build of the kernel does not implement */
  RTC_ERR_BAD_INDEX = 5,        /* an index array points outside its table                          */
  RTC_ERR_NO_DEVICE = 6,        /* no HIP device / HIP runtime failure; message carries hipError    */
  RTC_ERR_NOT_AFFINE = 7,       /* a shape/pattern inverse whose last row is not (0,0,0,1)          */
  RTC_ERR_OVERFLOW = 8          /* a per-lane device stack or csg intersection list overflowed        */
} rtc_status;

/* ---- leaf kinds: the Shape(T).Variant tags that can be hit (shape.zig:99-111) ---- */
enum {
  RTC_SPHERE = 0,          /* shapes/sphere.zig   */
  RTC_PLANE = 1,           /* shapes/plane.zig    */
  RTC_CUBE = 2,            /* shapes/cube.zig     */
  RTC_CYLINDER = 3,        /* shapes/cylinder.zig */
  RTC_TRIANGLE = 4,        /* shapes/triangle.zig:17-81   */
  RTC_SMOOTH_TRIANGLE = 5, /* shapes/triangle.zig:210-274 */
  RTC_CONE = 6             /* shapes/cone.zig; geometry in the cyl_* tables like a cylinder */
};

/* ---- pattern kinds: Pattern(T).Variant tags (patterns/pattern.zig:34-45) ---- */
enum {
  RTC_PAT_SOLID = 0,           /* patterns/solid.zig    */
  RTC_PAT_STRIPES = 1,         /* patterns/stripes.zig  */
  RTC_PAT_RINGS = 2,           /* patterns/rings.zig    */
  RTC_PAT_GRADIENT = 3,        /* patterns/gradient.zig */
  RTC_PAT_RADIAL_GRADIENT = 4, /* patterns/gradient.zig */
  RTC_PAT_CHECKERS = 5,        /* patterns/checkers.zig */
  RTC_PAT_BLEND = 6,           /* patterns/blend.zig    */
  RTC_PAT_PERTURB = 7,         /* patterns/perturb.zig: pat_a = the perturbed pattern, pat_rgb = PerturbInfo
                                  {scale_value, octaves, persistence} (defaults 0.3, 3, 0.8) */
  RTC_PAT_TEXTURE_MAP = 8,     /* patterns/texture_map.zig: pat_a = index into the tex_* tables */
  RTC_PAT_TEST = 9             /* TestPattern, pattern.zig:136-150: colour = pattern-space point */
};

/* rtc_scene_desc::tex_mapping: TextureMap variants (texture_map.zig:173-305) */
#define RTC_TEX_SPHERICAL 0u
#define RTC_TEX_PLANAR 1u
#define RTC_TEX_CYLINDRICAL 2u
#define RTC_TEX_CUBIC 3u
/* rtc_scene_desc::uv_kind: UvPattern variants (texture_map.zig:9-121) */
#define RTC_UV_ALIGN_CHECK 0u /* uv_sub = central, upper-left, upper-right, bottom-left, bottom-right */
#define RTC_UV_CHECKERS 1u    /* uv_size = width, height; uv_sub[0..1] = a, b                        */
#define RTC_UV_IMAGE 2u       /* uv_image = image index, uv_interp = 0 none / 1 bilinear             */
#define RTC_UV_TEST 3u        /* UvTestPattern: colour = (u, v, 0)                                   */

/* rtc_scene_desc::node_op (shapes/csg.zig:16-20) */
#define RTC_CSG_NONE 0u
#define RTC_CSG_UNION 1u
#define RTC_CSG_INTERSECTION 2u
#define RTC_CSG_DIFFERENCE 3u

/* Children / roots are encoded as one u32: high bit set = group node index, else leaf index. */
#define RTC_CHILD_NODE_BIT 0x80000000u

/* Number of doubles per material row: ambient, diffuse, specular, shininess,
 * reflective, transparency, refractive_index (material.zig:18-25).              */
#define RTC_MAT_STRIDE 7

/*
 * Flattened World(T) (world.zig:24-25): SoA tables, caller-owned.
 *
 * Leaves are the hit-able Shapes; `leaf_*` arrays are indexed by leaf.  The
 * order in which leaves are reached by a depth-first walk of roots[] /
 * children[] is the order the reference's nested stable sorts preserve for
 * equal-t intersections (world.zig:81, group.zig:59) and is used as the
 * tie-break; the leaf arrays themselves may be in any order.
 *
 * Groups store no transform of their own (shape.zig:286-296): every leaf holds
 * the full world<->object matrices, so only the inverse and its transpose are
 * needed on the path (shape.zig:133-145, 313-318).  Transforms are a table so
 * that the thousands of triangles of one OBJ instance share one entry.
 */
typedef struct rtc_scene_desc {
  uint32_t abi_version;      /* RTC_ABI_VERSION */

  uint32_t n_xforms;
  const double *xf_inv;      /* [n_xforms][16]  Shape._inverse_transform           */
  const double *xf_inv_t;    /* [n_xforms][16]  Shape._inverse_transform_transpose */

  uint32_t n_leaves;
  const uint8_t *leaf_kind;      /* RTC_SPHERE ...                                        */
  const uint32_t *leaf_xform;    /* index into xf_*                                       */
  const uint32_t *leaf_material; /* index into mat_*                                      */
  const uint8_t *leaf_shadow;    /* Shape.casts_shadow (shape.zig:119)                    */
  const uint32_t *leaf_id;       /* Shape.id (shape.zig:113): containers walk identity    */
  const uint32_t *leaf_geom;     /* cylinder/cone: index into cyl_*; triangles: into tri_* */

  uint32_t n_cyls;               /* cylinder.zig:26-28 (also cones) */
  const double *cyl_min;
  const double *cyl_max;
  const uint8_t *cyl_closed;

  uint32_t n_tris;               /* triangle.zig:21-26, 214-221 */
  const double *tri_p1;          /* [n_tris][3] */
  const double *tri_e1;          /* [n_tris][3] p2 - p1 */
  const double *tri_e2;          /* [n_tris][3] p3 - p1 */
  const double *tri_n1;          /* [n_tris][3] smooth: n1; flat: the stored face normal (shape.zig:190) */
  const double *tri_n2;          /* [n_tris][3] smooth only */
  const double *tri_n3;          /* [n_tris][3] smooth only */

  uint32_t n_materials;
  const double *mat_params;      /* [n_materials][RTC_MAT_STRIDE] */
  const uint32_t *mat_pattern;   /* index into pat_*              */

  uint32_t n_patterns;
  const uint8_t *pat_kind;       /* RTC_PAT_*                                              */
  const double *pat_inv;         /* [n_patterns][16] Pattern._inverse_transform            */
  const double *pat_rgb;         /* [n_patterns][3]  solid colour                          */
  const uint32_t *pat_a;         /* sub-pattern a (stripes/checkers/...), else 0           */
  const uint32_t *pat_b;         /* sub-pattern b                                          */

  uint32_t n_nodes;              /* one node per reference Group (group.zig:17-23)         */
  const double *node_min;        /* [n_nodes][3] Group._bbox min (bounding_box.zig:21)     */
  const double *node_max;        /* [n_nodes][3]                                           */
  const uint32_t *node_first;    /* first entry of this group's children in children[]     */
  const uint32_t *node_count;    /* Group.children.items.len; exactly 2 for a csg node     */
  const uint8_t *node_op;        /* [n_nodes] RTC_CSG_NONE for a Group, else the Csg.operation
                                    (csg.zig:16-20): children[first] is `left`, [first+1] `right`,
                                    node_min/max the csg's _bbox (shape.zig:257-261)          */
  uint32_t n_children;
  const uint32_t *children;      /* mixed list, RTC_CHILD_NODE_BIT marks a sub-group       */

  uint32_t n_roots;              /* World.objects (world.zig:24), in order                 */
  const uint32_t *roots;         /* same encoding as children[]                            */

  uint32_t n_lights;             /* World.lights (world.zig:25), point lights (light.zig)  */
  const double *light_pos;       /* [n_lights][3] */
  const double *light_rgb;       /* [n_lights][3] */

  /* Texture maps (patterns/texture_map.zig); all counts may be 0 and the pointers NULL.    */
  uint32_t n_texmaps;            /* a pattern of kind RTC_PAT_TEXTURE_MAP names one by pat_a */
  const uint8_t *tex_mapping;    /* RTC_TEX_*                                              */
  const uint32_t *tex_uv;        /* [n_texmaps][6] uv-pattern per face, Cubic.Face order front, back, left,
                                    right, up, down (texture_map.zig:216); other mappings use entry 0 */
  uint32_t n_uvs;
  const uint8_t *uv_kind;        /* RTC_UV_*                                               */
  const double *uv_size;         /* [n_uvs][2] UvCheckers width, height                    */
  const uint32_t *uv_sub;        /* [n_uvs][5] pattern indices (see RTC_UV_*)              */
  const uint32_t *uv_image;      /* [n_uvs] image index (RTC_UV_IMAGE)                     */
  const uint8_t *uv_interp;      /* [n_uvs] UvImage.Interpolation: 0 None, 1 Bilinear      */
  uint32_t n_images;             /* Canvas(T) behind a UvImage (canvas.zig:34-46)          */
  const uint32_t *img_width;
  const uint32_t *img_height;
  const uint64_t *img_offset;    /* first pixel of the image in img_rgb                    */
  const float *img_rgb;          /* [pixels][3], row-major: the f32 colour zigimg's iterator yields, which
                                    canvas.zig:41 widens to T                               */
} rtc_scene_desc;

/* Camera(T) after Camera.new + setTransform (camera.zig:18-61). */
typedef struct rtc_camera {
  uint32_t hsize, vsize;
  double half_width, half_height, pixel_size; /* camera.zig:33-52 */
  double inv_view[16];                        /* Camera._inverse_transform */
} rtc_camera;

/* Ray counters of the most recent render on a handle ("ray" = one World.intersect call). */
typedef struct rtc_stats {
  uint64_t primary;       /* camera.zig:117-118: one per pixel                                  */
  uint64_t secondary;     /* reflectedColor + refractedColor calls that recurse (world.zig:164,186) */
  uint64_t shadow_calls;  /* isShadowed calls the reference makes (world.zig:92)                */
  uint64_t shadow_traced; /* shadow rays this library actually traced (skips provably inert ones) */
  uint64_t overflow;      /* lanes that overflowed a stack / csg list (must be 0)                */
} rtc_stats;

typedef struct rtc_scene rtc_scene; /* opaque; owns the device copies and one HIP stream */

/* Validates `desc`, copies it into HBM on the current HIP device. */
int rtc_scene_create(const rtc_scene_desc *desc, rtc_scene **out);
void rtc_scene_destroy(rtc_scene *scene);

/*
 * A second handle on the SAME scene: shares the source's device copy of the scene (no upload, no BVH build; freed
 * with the last handle that uses it) and has a stream, launch counters, schedule and measurements of its own.  A
 * handle runs one frame at a time; independent frames - an orbit's, an animation's, a rank's share of a split frame,
 * which is too short to fill a GPU by itself - go to a handle and its clones in turn, each on a stream of its own,
 * and the work-groups of a frame start on the CUs the frame before has left (bench.py --inflight, rtc_multi.h's
 * RTC_MULTI_FRAMES).  Same device as the source (made current here); either may be destroyed first.
 */
int rtc_scene_clone(const rtc_scene *source, rtc_scene **out);

/*
 * Replaces Camera.render (camera.zig:80-125) for the tile [x0,x0+w) x [y0,y0+h):
 * rgb_out[(y-y0)*w + (x-x0)][0..2] = colorAt(rayForPixel(x,y), max_depth).
 * The reference value of max_depth is 5 (camera.zig:118).  `rgb_out` is host
 * memory, [h][w][3] doubles.  Synchronous, and without side effects on the
 * caller's memory: the frame is copied into `rgb_out`, nothing is remembered
 * about the pointer.  (Into pageable memory that copy runs at a fifth of the
 * link's rate; see rtc_canvas_register.)  The copy of a frame takes as long as
 * its render or longer, so a large frame (from 400 000 pixels) is rendered in
 * two or four horizontal bands one after the other, each copied while the next
 * renders (1080p into a registered canvas: 1.1 ms instead of 1.45; the lower
 * bands run on clones of the handle, made on first use; rtc_get_stats sums the
 * bands; option "host_bands" forces the count).
 */
int rtc_render(rtc_scene *scene, const rtc_camera *cam, uint32_t max_depth,
               uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, double *rgb_out);

/*
 * Optional, for a host that renders frame after frame into ONE canvas (the interactive seam, lib.zig:135-190): pins
 * the caller's buffer for the HIP runtime (hipHostRegister), so that rtc_render's / rtc_multi_render's copy into it
 * runs at link speed (1080p: 1.1 ms per call instead of 3.8-5.6).  Explicit on purpose: the registration belongs to the
 * MEMORY, not to a scene handle, and the caller - who knows when the canvas is freed - drops it with
 * rtc_canvas_unregister BEFORE freeing or reallocating the buffer.  Both are process-wide and need no scene.
 */
int rtc_canvas_register(void *canvas, size_t bytes);
int rtc_canvas_unregister(void *canvas);

/*
 * The same frame as the RGBA8 framebuffer of the reference's interactive seam (Renderer, src/lib.zig:135-164):
 * rgba_out[((y-y0)*w + (x-x0))*4 + 0..2] = clamp(channel) of src/raytracer/color.zig:61-71 (@round of
 * channel * 255, clamped to 0..255), [+3] = 255.  Clamped on the device: 4 bytes per pixel cross the link
 * instead of 24.  `rgba_out` is host memory, [h][w][4] bytes.  Synchronous.
 */
int rtc_render_rgba8(rtc_scene *scene, const rtc_camera *cam, uint32_t max_depth,
                     uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint8_t *rgba_out);

/* The clamp alone, device to device: d_rgba[i] = R | G << 8 | B << 16 | 255 << 24 of pixel i of an [n_pixels][3] f64
 * canvas on the current device; asynchronous on `hip_stream` (not NULL).  Needs no scene. */
int rtc_rgba8_device(const double *d_canvas, size_t n_pixels, uint32_t *d_rgba, void *hip_stream);

/*
 * Same, but `d_rgb_out` is device memory on the scene's device and the work is
 * enqueued on `hip_stream` (a hipStream_t; NULL = the handle's own stream, which
 * is made at the first call that needs it: a host that always passes its streams
 * keeps the device's few hardware queues for them - INTEGRATION.md, "Streams and
 * hardware queues") without synchronising.  This is the entry point the multi-GPU driver uses
 * with device buffers owned by its RCCL communicator.
 */
int rtc_render_device(rtc_scene *scene, const rtc_camera *cam, uint32_t max_depth,
                      uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                      double *d_rgb_out, void *hip_stream);

/*
 * Multi-GPU tile partition: the image is cut into tile_w x tile_h tiles
 * numbered row-major; this call renders tiles first_tile, first_tile+stride,
 * ... (n_my_tiles of them) into the compact buffer
 * d_rgb_out[k][tile_h][tile_w][3]; pixels of edge tiles that fall outside the
 * image are NOT written (the library never clears the caller's buffer: zero it
 * once if the padding is read, e.g. by a gather).  Asynchronous on `hip_stream`
 * like rtc_render_device.
 */
int rtc_render_tiles_device(rtc_scene *scene, const rtc_camera *cam, uint32_t max_depth,
                            uint32_t tile_w, uint32_t tile_h, uint32_t first_tile,
                            uint32_t tile_stride, uint32_t n_my_tiles,
                            double *d_rgb_out, void *hip_stream);

/*
 * The same with an explicit list of tiles (host memory, copied): region k of d_rgb_out is tile tiles[k].  This is
 * what a cost-balanced split uses (rtc_assign_tiles below); a handle keeps the last list on the device, so
 * rendering the same list again copies nothing.
 */
int rtc_render_tile_list_device(rtc_scene *scene, const rtc_camera *cam, uint32_t max_depth,
                                uint32_t tile_w, uint32_t tile_h, const uint32_t *tiles,
                                uint32_t n_my_tiles, double *d_rgb_out, void *hip_stream);

/*
 * What the regions (tiles) of the most recent MEASURING tile-mode render on this handle took, in the library's
 * wave-time units (only ratios mean anything): cost_out[k] for region k.  A render measures when its schedule
 * was not measured on its own view, i.e. the first frame of a tile list and every frame of a moving camera.
 * Synchronises.  RTC_ERR_INVALID_ARGUMENT if there is no such measurement of n_regions regions.
 */
int rtc_get_tile_costs(rtc_scene *scene, double *cost_out, uint32_t n_regions);

/*
 * Cost-balanced split of the image's tiles over `world` ranks (host only, deterministic: every rank that calls it
 * with the same costs gets the same table): longest tile first onto the least-loaded rank that still has a free
 * slot, every rank holding at most padded = ceil(n_tiles / world) tiles so that one equal-count gather moves
 * them.  rank_of_tile[t] = the rank that renders tile t; slot_of_tile[t] = rank * padded + k, the tile's place in
 * the gathered buffer [world][padded][tile_h][tile_w][3] (a rank renders its tiles in increasing tile order).
 */
int rtc_assign_tiles(const double *tile_cost, uint32_t n_tiles, uint32_t world,
                     uint32_t *rank_of_tile, uint32_t *slot_of_tile);

/* rtc_assemble_tiles_device for such a split: d_slot_of_tile is rtc_assign_tiles' table in device memory. */
int rtc_assemble_tile_list_device(const double *d_gathered, const uint32_t *d_slot_of_tile,
                                  uint32_t tile_w, uint32_t tile_h, uint32_t hsize, uint32_t vsize,
                                  double *d_canvas, void *hip_stream);

/* The same for shares that were clamped before they were gathered (rtc_rgba8_device on a rank's tile buffer: 4 bytes
 * per pixel cross the links instead of 24 - an 8-GPU host of lib.zig's RGBA8 framebuffer is otherwise bound by the gather,
 * not the render): d_gathered_rgba[world][padded][tile_h][tile_w] -> d_rgba[vsize][hsize]. */
int rtc_assemble_tile_list_rgba8_device(const uint32_t *d_gathered_rgba, const uint32_t *d_slot_of_tile,
                                        uint32_t tile_w, uint32_t tile_h, uint32_t hsize, uint32_t vsize,
                                        uint32_t *d_rgba, void *hip_stream);

/*
 * Rank 0 of the tile partition, after the gather: copies the ranks' compact tile
 * buffers d_gathered[world][padded_tiles][tile_h][tile_w][3] (rank r's k-th tile
 * is tile r + k * world of the row-major tiling) into the row-major canvas
 * d_canvas[vsize][hsize][3].  Device to device on the current device,
 * asynchronous on `hip_stream` (which must not be NULL).  Needs no scene.
 * Replaces the row-major Canvas the reference fills in place (camera.zig:121).
 */
int rtc_assemble_tiles_device(const double *d_gathered, uint32_t world, uint32_t padded_tiles,
                              uint32_t tile_w, uint32_t tile_h, uint32_t hsize, uint32_t vsize,
                              double *d_canvas, void *hip_stream);

/*
 * A rank's compact tiles (as rtc_render_tile_list_device leaves them: d_tiles[k] is tile d_tile_list[k]) written
 * straight to their places in a row-major canvas [vsize][hsize][3] - `canvas` being any memory the current device can
 * write: device memory, or the caller's host canvas after rtc_canvas_register (pinned and mapped).  With a registered
 * host canvas every GPU of a split frame sends its own share over its own host link; nothing funnels through rank 0
 * (rtc_multi.h does this for its host forms).  Pixels of edge tiles outside the image are skipped.  The _rgba8 form
 * clamps on the way (color.zig:61-71) into [vsize][hsize] RGBA8.  d_tile_list is device memory.  Asynchronous on
 * `hip_stream` (not NULL), on the current device.  Needs no scene.  Replaces Canvas writes of camera.zig:119-121.
 */
int rtc_scatter_tile_list_device(const double *d_tiles, const uint32_t *d_tile_list, uint32_t n_tiles, uint32_t tile_w,
                                 uint32_t tile_h, uint32_t hsize, uint32_t vsize, double *canvas, void *hip_stream);
int rtc_scatter_tile_list_rgba8_device(const double *d_tiles, const uint32_t *d_tile_list, uint32_t n_tiles,
                                       uint32_t tile_w, uint32_t tile_h, uint32_t hsize, uint32_t vsize, uint32_t *rgba,
                                       void *hip_stream);

/* Waits for the work enqueued on the handle's own stream (nothing to wait for if it has never used one). */
int rtc_scene_synchronize(rtc_scene *scene);

/* Counters of the last render that was enqueued on this handle (synchronises). */
int rtc_get_stats(rtc_scene *scene, rtc_stats *out);

/*
 * For callers of the asynchronous entry points, after a frame whose rtc_stats::overflow is not 0 on a scene with csg
 * nodes (Csg.filterIntersections' list, csg.zig:51-95, is of any length in the reference; a lane's list here is a
 * buffer): waits for the handle's last launch and, if what ran out was a csg intersection list, sizes the handle's lists
 * for what that frame needed (in one step, up to 1024 entries).  RTC_OK: the lists are longer now (or nothing had
 * overflowed) - render the frame again; RTC_ERR_OVERFLOW: the overflow was not a csg list's, or the lists are at their
 * maximum; RTC_ERR_OUT_OF_MEMORY (at the next render): the longer lists do not fit, the handle keeps the old ones.
 * rtc_render and rtc_render_rgba8 do this by themselves.
 */
int rtc_grow_csg_lists(rtc_scene *scene);

/* (Diagnostics and tuning - rtc_set_option, rtc_get_schedule, rtc_get_chunk_times, rtc_last_kernel_name - are declared in
 * rtc_diag.h: a host that binds the render path does not need them.) */

/* Thread-local, static storage; "" when the last call on this thread succeeded. */
const char *rtc_last_error(void);
/* "NotInvertible", "OutOfMemory", ... (Zig error-name style, cf. lib.zig:226-227). */
const char *rtc_status_name(int status);

#ifdef __cplusplus
}
#endif
#endif /* RTC_H */
