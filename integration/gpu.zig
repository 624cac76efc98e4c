//! gpu.zig — the reference-side binding of librtc_hip.so (include/rtc.h): extern declarations, the Flattener that
//! turns a `World(f64)` into the flat `rtc_scene_desc`, and `render`, the replacement body of `Camera(T).render`
//! (src/raytracer/camera.zig:80-125).
//!
//! NOT COMPILED HERE: the build image has no Zig toolchain (SURVEY F8), so this file is shipped as source for a
//! maintainer of SinclaM/ray-tracer-challenge to drop into src/raytracer/ (it is written against the reference at the
//! commit under /root/reference: field and variant names are the reference's own, cited below).  The same ABI is
//! exercised from C++ (ray-tracer-challenge_amd/host/rtc_flatten.cpp is this file's twin, statement by statement)
//! and from Python, which is what the test suite runs; tests/test_abi.py checks that the extern struct below lists
//! rtc.h's fields in order.
//!
//! build.zig:
//!     exe.addIncludePath(.{ .path = "path/to/include" });
//!     exe.addLibraryPath(.{ .path = "path/to/ray-tracer-challenge_amd/lib" });
//!     exe.linkSystemLibrary("rtc_hip");   // librtc_hip.so (links libamdhip64)
//!     exe.linkLibC();

const std = @import("std");
const Allocator = std.mem.Allocator;
const ArrayList = std.ArrayList;

const Tuple = @import("tuple.zig").Tuple;
const Matrix = @import("matrix.zig").Matrix;
const Color = @import("color.zig").Color;
const Canvas = @import("canvas.zig").Canvas;
const Shape = @import("shapes/shape.zig").Shape;
const Material = @import("material.zig").Material;
const Pattern = @import("patterns/pattern.zig").Pattern;
const UvPattern = @import("patterns/texture_map.zig").UvPattern;
const Light = @import("light.zig").Light;
const World = @import("world.zig").World;
const Camera = @import("camera.zig").Camera;

// ------------------------------------------------------------------------------------------------ include/rtc.h

pub const RTC_ABI_VERSION: u32 = 3;
pub const RTC_CHILD_NODE_BIT: u32 = 0x80000000;
pub const RTC_MAT_STRIDE = 7;

/// rtc.h: leaf kinds (Shape(T).Variant tags that can be hit, shape.zig:99-111)
pub const LeafKind = enum(u8) { sphere = 0, plane = 1, cube = 2, cylinder = 3, triangle = 4, smooth_triangle = 5, cone = 6 };
/// rtc.h: pattern kinds (Pattern(T).Variant tags, pattern.zig:34-45)
pub const PatKind = enum(u8) {
    solid = 0, stripes = 1, rings = 2, gradient = 3, radial_gradient = 4, checkers = 5, blend = 6, perturb = 7,
    texture_map = 8, test_pattern = 9,
};
pub const TexMapping = enum(u8) { spherical = 0, planar = 1, cylindrical = 2, cubic = 3 };
pub const UvKind = enum(u8) { align_check = 0, checkers = 1, image = 2, test_pattern = 3 };
/// rtc.h: node_op (csg.zig:16-20)
pub const CsgOp = enum(u8) { none = 0, @"union" = 1, intersection = 2, difference = 3 };

pub const RtcSceneDesc = extern struct {
    abi_version: u32,
    n_xforms: u32, xf_inv: [*]const f64, xf_inv_t: [*]const f64,
    n_leaves: u32, leaf_kind: [*]const u8, leaf_xform: [*]const u32, leaf_material: [*]const u32,
    leaf_shadow: [*]const u8, leaf_id: [*]const u32, leaf_geom: [*]const u32,
    n_cyls: u32, cyl_min: [*]const f64, cyl_max: [*]const f64, cyl_closed: [*]const u8,
    n_tris: u32, tri_p1: [*]const f64, tri_e1: [*]const f64, tri_e2: [*]const f64,
    tri_n1: [*]const f64, tri_n2: [*]const f64, tri_n3: [*]const f64,
    n_materials: u32, mat_params: [*]const f64, mat_pattern: [*]const u32,
    n_patterns: u32, pat_kind: [*]const u8, pat_inv: [*]const f64, pat_rgb: [*]const f64,
    pat_a: [*]const u32, pat_b: [*]const u32,
    n_nodes: u32, node_min: [*]const f64, node_max: [*]const f64, node_first: [*]const u32, node_count: [*]const u32,
    node_op: [*]const u8,
    n_children: u32, children: [*]const u32,
    n_roots: u32, roots: [*]const u32,
    n_lights: u32, light_pos: [*]const f64, light_rgb: [*]const f64,
    n_texmaps: u32, tex_mapping: [*]const u8, tex_uv: [*]const u32,
    n_uvs: u32, uv_kind: [*]const u8, uv_size: [*]const f64, uv_sub: [*]const u32, uv_image: [*]const u32, uv_interp: [*]const u8,
    n_images: u32, img_width: [*]const u32, img_height: [*]const u32, img_offset: [*]const u64, img_rgb: [*]const f32,
};
pub const RtcCamera = extern struct {
    hsize: u32, vsize: u32, half_width: f64, half_height: f64, pixel_size: f64, inv_view: [16]f64,
};
pub const RtcStats = extern struct { primary: u64, secondary: u64, shadow_calls: u64, shadow_traced: u64, overflow: u64 };
pub const RtcScene = opaque {};

pub extern fn rtc_scene_create(desc: *const RtcSceneDesc, out: *?*RtcScene) c_int;
pub extern fn rtc_scene_clone(source: ?*const RtcScene, out: *?*RtcScene) c_int;
pub extern fn rtc_scene_destroy(scene: ?*RtcScene) void;
pub extern fn rtc_render(scene: *RtcScene, cam: *const RtcCamera, max_depth: u32,
                         x0: u32, y0: u32, w: u32, h: u32, rgb_out: [*]f64) c_int;
pub extern fn rtc_render_rgba8(scene: *RtcScene, cam: *const RtcCamera, max_depth: u32,
                               x0: u32, y0: u32, w: u32, h: u32, rgba_out: [*]u8) c_int; // lib.zig's framebuffer
pub extern fn rtc_render_device(scene: *RtcScene, cam: *const RtcCamera, max_depth: u32, x0: u32, y0: u32, w: u32, h: u32,
                                d_rgb_out: [*]f64, hip_stream: ?*anyopaque) c_int; // output stays in HBM
pub extern fn rtc_canvas_register(canvas: *anyopaque, bytes: usize) c_int; // pin a canvas that is rendered into again and again ...
pub extern fn rtc_canvas_unregister(canvas: *anyopaque) c_int; // ... and drop the pin BEFORE the canvas is freed
pub extern fn rtc_rgba8_device(d_canvas: [*]const f64, n_pixels: usize, d_rgba: [*]u32, hip_stream: ?*anyopaque) c_int;
pub extern fn rtc_scene_synchronize(scene: *RtcScene) c_int;
pub extern fn rtc_get_stats(scene: *RtcScene, out: *RtcStats) c_int;
pub extern fn rtc_grow_csg_lists(scene: *RtcScene) c_int; // after an asynchronous frame whose stats.overflow != 0 on a csg scene: longer lists, render again
pub extern fn rtc_scatter_tile_list_device(d_tiles: [*]const f64, d_tile_list: [*]const u32, n_tiles: u32, tile_w: u32, tile_h: u32,
                                           hsize: u32, vsize: u32, canvas: [*]f64, hip_stream: ?*anyopaque) c_int; // a rank's tiles straight into a (registered host) canvas
pub extern fn rtc_scatter_tile_list_rgba8_device(d_tiles: [*]const f64, d_tile_list: [*]const u32, n_tiles: u32, tile_w: u32, tile_h: u32,
                                                 hsize: u32, vsize: u32, rgba: [*]u32, hip_stream: ?*anyopaque) c_int;
pub extern fn rtc_last_error() [*:0]const u8;
pub extern fn rtc_status_name(status: c_int) [*:0]const u8;

// include/rtc_diag.h (diagnostics and tuning: not needed to render; the same library exports them)
pub extern fn rtc_set_option(name: [*:0]const u8, value: f64) c_int; // tuning / test options; the library reads no environment
pub extern fn rtc_get_chunk_times(scene: *RtcScene, cam: *const RtcCamera, estimated: ?[*]u32, measured: ?[*]u32, capacity: usize, n_chunks: *u32) c_int; // diagnostic
pub extern fn rtc_last_kernel_name(scene: *const RtcScene) [*:0]const u8;
pub extern fn rtc_get_schedule(scene: *RtcScene, items: ?[*]u32, capacity_items: usize, n_packets: *u32) c_int;

// include/rtc_multi.h (librtc_multi.so: all GPUs of the node behind one call, one ncclGather per frame)
pub const RtcMulti = opaque {};
pub extern fn rtc_multi_create(desc: *const RtcSceneDesc, n_gpus: u32, flags: u32, out: *?*RtcMulti) c_int;
pub extern fn rtc_multi_destroy(m: ?*RtcMulti) void;
pub extern fn rtc_multi_render(m: *RtcMulti, cam: *const RtcCamera, max_depth: u32, rgb_out: [*]f64) c_int;
pub extern fn rtc_multi_render_rgba8(m: *RtcMulti, cam: *const RtcCamera, max_depth: u32, rgba_out: [*]u8) c_int; // lib.zig's framebuffer
pub extern fn rtc_multi_render_device(m: *RtcMulti, cam: *const RtcCamera, max_depth: u32, d_canvas: *?[*]const f64) c_int; // stays on GPU 0
pub extern fn rtc_multi_render_rgba8_device(m: *RtcMulti, cam: *const RtcCamera, max_depth: u32, d_rgba: *?[*]const u32) c_int; // ... as RGBA8
pub extern fn rtc_multi_synchronize(m: *RtcMulti) c_int;
pub extern fn rtc_multi_stream(m: *RtcMulti) ?*anyopaque;
pub extern fn rtc_multi_last_error() [*:0]const u8;

pub const GpuError = error{ GpuSceneRejected, GpuRenderFailed };

// ------------------------------------------------------------------------------------------------ Flattener

/// Walks `World.objects` depth-first in the order `World.intersect` / `Group.localIntersect` visit shapes
/// (world.zig:74, group.zig:52): the position of a leaf in that walk is what the reference's nested stable sorts give an
/// intersection among equal t's, and the library uses it as the tie-break.  One node per Group / Csg with the shape's own
/// `_bbox` (group.zig:23, csg.zig:32): identical boxes give identical candidate sets.
pub const Flattener = struct {
    const Self = @This();
    const T = f64;

    allocator: Allocator,
    xf_inv: ArrayList(f64), xf_inv_t: ArrayList(f64),
    leaf_kind: ArrayList(u8), leaf_xform: ArrayList(u32), leaf_material: ArrayList(u32), leaf_shadow: ArrayList(u8),
    leaf_id: ArrayList(u32), leaf_geom: ArrayList(u32),
    cyl_min: ArrayList(f64), cyl_max: ArrayList(f64), cyl_closed: ArrayList(u8),
    tri_p1: ArrayList(f64), tri_e1: ArrayList(f64), tri_e2: ArrayList(f64),
    tri_n1: ArrayList(f64), tri_n2: ArrayList(f64), tri_n3: ArrayList(f64),
    mat_params: ArrayList(f64), mat_pattern: ArrayList(u32),
    pat_kind: ArrayList(u8), pat_inv: ArrayList(f64), pat_rgb: ArrayList(f64), pat_a: ArrayList(u32), pat_b: ArrayList(u32),
    node_min: ArrayList(f64), node_max: ArrayList(f64), node_first: ArrayList(u32), node_count: ArrayList(u32),
    node_op: ArrayList(u8), children: ArrayList(u32), roots: ArrayList(u32),
    light_pos: ArrayList(f64), light_rgb: ArrayList(f64),
    tex_mapping: ArrayList(u8), tex_uv: ArrayList(u32),
    uv_kind: ArrayList(u8), uv_size: ArrayList(f64), uv_sub: ArrayList(u32), uv_image: ArrayList(u32), uv_interp: ArrayList(u8),
    img_width: ArrayList(u32), img_height: ArrayList(u32), img_offset: ArrayList(u64), img_rgb: ArrayList(f32),

    pub fn init(allocator: Allocator) Self {
        var self: Self = undefined;
        self.allocator = allocator;
        inline for (std.meta.fields(Self)) |field| {
            if (comptime std.mem.eql(u8, field.name, "allocator")) continue;
            @field(self, field.name) = field.type.init(allocator);
        }
        return self;
    }

    pub fn deinit(self: *Self) void {
        inline for (std.meta.fields(Self)) |field| {
            if (comptime std.mem.eql(u8, field.name, "allocator")) continue;
            @field(self, field.name).deinit();
        }
    }

    fn push3(list: *ArrayList(f64), t: Tuple(T)) !void {
        try list.appendSlice(&[_]f64{ t.x, t.y, t.z });
    }

    fn pushMatrix(list: *ArrayList(f64), m: Matrix(T, 4)) !void { // row-major, matrix.zig:15
        for (m.data) |row| try list.appendSlice(&row);
    }

    /// Shape._inverse_transform / _inverse_transform_transpose (shape.zig:115-116); the thousands of triangles of one
    /// OBJ instance share one entry: a linear search over the LAST entry suffices for that (consecutive leaves).
    fn internXform(self: *Self, s: *const Shape(T)) !u32 {
        const n = self.xf_inv.items.len / 16;
        if (n > 0) {
            const last = self.xf_inv.items[(n - 1) * 16 ..][0..16];
            var same = true;
            for (s._inverse_transform.data, 0..) |row, r| {
                for (row, 0..) |v, c| same = same and (@as(u64, @bitCast(v)) == @as(u64, @bitCast(last[r * 4 + c])));
            }
            if (same) return @intCast(n - 1);
        }
        try pushMatrix(&self.xf_inv, s._inverse_transform);
        try pushMatrix(&self.xf_inv_t, s._inverse_transform_transpose);
        return @intCast(n);
    }

    /// Pattern(T) (pattern.zig:34-49) -> one table entry; higher-order patterns name their children by index
    /// (stripes.zig:20-21, gradient.zig:20-21, rings.zig:20-21, checkers.zig:16-17, blend.zig:16-17, perturb.zig:18).
    fn internPattern(self: *Self, p: *const Pattern(T)) !u32 {
        var kind: PatKind = .solid;
        var a: u32 = 0;
        var b: u32 = 0;
        var rgb = [3]f64{ 0.0, 0.0, 0.0 };
        switch (p.variant) {
            .solid => |v| { kind = .solid; rgb = .{ v.a.r, v.a.g, v.a.b }; },
            .test_pattern => kind = .test_pattern,
            .stripes => |v| { kind = .stripes; a = try self.internPattern(v.a); b = try self.internPattern(v.b); },
            .rings => |v| { kind = .rings; a = try self.internPattern(v.a); b = try self.internPattern(v.b); },
            .gradient => |v| { kind = .gradient; a = try self.internPattern(v.a); b = try self.internPattern(v.b); },
            .radial_gradient => |v| { kind = .radial_gradient; a = try self.internPattern(v.a); b = try self.internPattern(v.b); },
            .checkers => |v| { kind = .checkers; a = try self.internPattern(v.a); b = try self.internPattern(v.b); },
            .blend => |v| { kind = .blend; a = try self.internPattern(v.a); b = try self.internPattern(v.b); },
            .perturb => |v| { // PerturbInfo travels in the colour slot (rtc.h: RTC_PAT_PERTURB)
                kind = .perturb;
                a = try self.internPattern(v.a);
                b = a;
                rgb = .{ v.info.scale_value, @floatFromInt(v.info.octaves), v.info.persistence };
            },
            .texture_map => |v| { kind = .texture_map; a = try self.internTextureMap(&v); },
        }
        const id: u32 = @intCast(self.pat_kind.items.len);
        try self.pat_kind.append(@intFromEnum(kind));
        try pushMatrix(&self.pat_inv, p._inverse_transform);
        try self.pat_rgb.appendSlice(&rgb);
        try self.pat_a.append(a);
        try self.pat_b.append(b);
        return id;
    }

    /// UvPattern(T) (texture_map.zig:107-165).
    fn internUv(self: *Self, uv: *const UvPattern(T)) !u32 {
        var kind: UvKind = .test_pattern;
        var sub = [5]u32{ 0, 0, 0, 0, 0 };
        var size = [2]f64{ 0.0, 0.0 };
        var image: u32 = 0;
        var interp: u8 = 0;
        switch (uv.variant) {
            .uv_test_pattern => kind = .test_pattern,
            .uv_align_check => |v| {
                kind = .align_check;
                sub = .{
                    try self.internPattern(v.central), try self.internPattern(v.upper_left), try self.internPattern(v.upper_right),
                    try self.internPattern(v.bottom_left), try self.internPattern(v.bottom_right),
                };
            },
            .uv_checkers => |v| {
                kind = .checkers;
                size = .{ v.width, v.height };
                sub[0] = try self.internPattern(v.a);
                sub[1] = try self.internPattern(v.b);
            },
            .uv_image => |v| { // Canvas(T) behind a UvImage (canvas.zig:34-46): [h][w][3], as f32 (what zigimg yielded)
                kind = .image;
                interp = if (v.interpolation == .Bilinear) 1 else 0;
                image = @intCast(self.img_width.items.len);
                try self.img_width.append(@intCast(v.canvas.width));
                try self.img_height.append(@intCast(v.canvas.height));
                try self.img_offset.append(self.img_rgb.items.len / 3);
                for (v.canvas.pixels) |px| try self.img_rgb.appendSlice(&[_]f32{ @floatCast(px.r), @floatCast(px.g), @floatCast(px.b) });
            },
        }
        const id: u32 = @intCast(self.uv_kind.items.len);
        try self.uv_kind.append(@intFromEnum(kind));
        try self.uv_size.appendSlice(&size);
        try self.uv_sub.appendSlice(&sub);
        try self.uv_image.append(image);
        try self.uv_interp.append(interp);
        return id;
    }

    /// TextureMap(T) (texture_map.zig:167-305); Cubic.Face order front, back, left, right, up, down (:216).
    fn internTextureMap(self: *Self, tm: anytype) !u32 {
        var mapping: TexMapping = .spherical;
        var uv = [6]u32{ 0, 0, 0, 0, 0, 0 };
        switch (tm.*) {
            .spherical => |v| { mapping = .spherical; uv[0] = try self.internUv(&v.uv_pattern); },
            .planar => |v| { mapping = .planar; uv[0] = try self.internUv(&v.uv_pattern); },
            .cylindrical => |v| { mapping = .cylindrical; uv[0] = try self.internUv(&v.uv_pattern); },
            .cubic => |v| {
                mapping = .cubic;
                for (v.face_patterns, 0..) |*face, f| uv[f] = try self.internUv(face);
            },
        }
        if (mapping != .cubic) for (uv[1..]) |*u| { u.* = uv[0]; };
        const id: u32 = @intCast(self.tex_mapping.items.len);
        try self.tex_mapping.append(@intFromEnum(mapping));
        try self.tex_uv.appendSlice(&uv);
        return id;
    }

    /// Material(T) (material.zig:18-25).
    fn internMaterial(self: *Self, m: *const Material(T)) !u32 {
        const pat = try self.internPattern(&m.pattern);
        const id: u32 = @intCast(self.mat_pattern.items.len);
        try self.mat_params.appendSlice(&[RTC_MAT_STRIDE]f64{
            m.ambient, m.diffuse, m.specular, m.shininess, m.reflective, m.transparency, m.refractive_index,
        });
        try self.mat_pattern.append(pat);
        return id;
    }

    /// Returns the encoded child reference (leaf index, or RTC_CHILD_NODE_BIT | node), or null for a shape that can
    /// never be hit (test_shape: shape.zig:411-420; a bare bounding_box is not a World object).
    fn visit(self: *Self, s: *const Shape(T)) !?u32 {
        var kind: LeafKind = .sphere;
        var geom: u32 = 0;
        switch (s.variant) {
            .group => |g| return RTC_CHILD_NODE_BIT | try self.visitNode(g._bbox, .none, g.children.items, null),
            .csg => |c| {
                const op: CsgOp = switch (c.operation) { // csg.zig:16-20
                    .Union => .@"union",
                    .Intersection => .intersection,
                    .Difference => .difference,
                };
                return RTC_CHILD_NODE_BIT | try self.visitNode(c._bbox, op, &[_]Shape(T){}, .{ c.left, c.right });
            },
            .test_shape, .bounding_box => return null,
            .sphere => kind = .sphere,
            .plane => kind = .plane,
            .cube => kind = .cube,
            .cylinder => |c| {
                kind = .cylinder;
                geom = @intCast(self.cyl_min.items.len);
                try self.cyl_min.append(c.min);
                try self.cyl_max.append(c.max);
                try self.cyl_closed.append(@intFromBool(c.closed));
            },
            .cone => |c| {
                kind = .cone;
                geom = @intCast(self.cyl_min.items.len);
                try self.cyl_min.append(c.min);
                try self.cyl_max.append(c.max);
                try self.cyl_closed.append(@intFromBool(c.closed));
            },
            .triangle => |t| { // triangle.zig:21-26; the stored face normal goes where a smooth triangle has n1
                kind = .triangle;
                geom = @intCast(self.tri_p1.items.len / 3);
                try push3(&self.tri_p1, t.p1);
                try push3(&self.tri_e1, t.e1);
                try push3(&self.tri_e2, t.e2);
                try push3(&self.tri_n1, t.normal);
                try self.tri_n2.appendSlice(&[_]f64{ 0.0, 0.0, 0.0 });
                try self.tri_n3.appendSlice(&[_]f64{ 0.0, 0.0, 0.0 });
            },
            .smooth_triangle => |t| { // triangle.zig:214-221
                kind = .smooth_triangle;
                geom = @intCast(self.tri_p1.items.len / 3);
                try push3(&self.tri_p1, t.p1);
                try push3(&self.tri_e1, t.e1);
                try push3(&self.tri_e2, t.e2);
                try push3(&self.tri_n1, t.n1);
                try push3(&self.tri_n2, t.n2);
                try push3(&self.tri_n3, t.n3);
            },
        }
        const leaf: u32 = @intCast(self.leaf_kind.items.len);
        try self.leaf_kind.append(@intFromEnum(kind));
        try self.leaf_xform.append(try self.internXform(s));
        try self.leaf_material.append(try self.internMaterial(&s.material));
        try self.leaf_shadow.append(@intFromBool(s.casts_shadow));
        try self.leaf_id.append(@intCast(s.id)); // Shape.id (shape.zig:113): identity in the containers walk
        try self.leaf_geom.append(geom);
        return leaf;
    }

    /// One node per Group (children: its list, group.zig:21) or Csg (children: left, right, csg.zig:28-29), with the
    /// shape's own _bbox.  A node's children are contiguous in children[]; sub-nodes are appended first.
    fn visitNode(self: *Self, bbox: *const Shape(T), op: CsgOp, list: []const Shape(T), pair: ?[2]*const Shape(T)) !u32 {
        const node: u32 = @intCast(self.node_first.items.len);
        try push3(&self.node_min, bbox.variant.bounding_box.min);
        try push3(&self.node_max, bbox.variant.bounding_box.max);
        try self.node_first.append(0);
        try self.node_count.append(0);
        try self.node_op.append(@intFromEnum(op));
        var refs = ArrayList(u32).init(self.allocator);
        defer refs.deinit();
        if (pair) |lr| {
            for (lr) |child| {
                // a csg always has a left and a right: a child that can never be hit becomes an empty group
                const r = (try self.visit(child)) orelse (RTC_CHILD_NODE_BIT | try self.emptyGroup());
                try refs.append(r);
            }
        } else {
            for (list) |*child| if (try self.visit(child)) |r| try refs.append(r);
        }
        self.node_first.items[node] = @intCast(self.children.items.len);
        self.node_count.items[node] = @intCast(refs.items.len);
        try self.children.appendSlice(refs.items);
        return node;
    }

    fn emptyGroup(self: *Self) !u32 {
        const node: u32 = @intCast(self.node_first.items.len);
        const inf = std.math.inf(f64);
        try self.node_min.appendSlice(&[_]f64{ inf, inf, inf }); // BoundingBox default (bounding_box.zig:21-22)
        try self.node_max.appendSlice(&[_]f64{ -inf, -inf, -inf });
        try self.node_first.append(@intCast(self.children.items.len));
        try self.node_count.append(0);
        try self.node_op.append(@intFromEnum(CsgOp.none));
        return node;
    }

    /// World.objects entry (world.zig:24), in order.
    pub fn addRoot(self: *Self, object: *const Shape(T)) !void {
        if (try self.visit(object)) |r| try self.roots.append(r);
    }

    /// World.lights entry (world.zig:25, light.zig:14-15).
    pub fn addLight(self: *Self, light: Light(T)) !void {
        try push3(&self.light_pos, light.position);
        try self.light_rgb.appendSlice(&[_]f64{ light.intensity.r, light.intensity.g, light.intensity.b });
    }

    /// View over the lists; valid while `self` is alive and unmodified.  (A zero-length list's pointer is never read.)
    pub fn desc(self: *const Self) RtcSceneDesc {
        return .{
            .abi_version = RTC_ABI_VERSION,
            .n_xforms = @intCast(self.xf_inv.items.len / 16), .xf_inv = self.xf_inv.items.ptr, .xf_inv_t = self.xf_inv_t.items.ptr,
            .n_leaves = @intCast(self.leaf_kind.items.len), .leaf_kind = self.leaf_kind.items.ptr,
            .leaf_xform = self.leaf_xform.items.ptr, .leaf_material = self.leaf_material.items.ptr,
            .leaf_shadow = self.leaf_shadow.items.ptr, .leaf_id = self.leaf_id.items.ptr, .leaf_geom = self.leaf_geom.items.ptr,
            .n_cyls = @intCast(self.cyl_min.items.len), .cyl_min = self.cyl_min.items.ptr, .cyl_max = self.cyl_max.items.ptr,
            .cyl_closed = self.cyl_closed.items.ptr,
            .n_tris = @intCast(self.tri_p1.items.len / 3), .tri_p1 = self.tri_p1.items.ptr, .tri_e1 = self.tri_e1.items.ptr,
            .tri_e2 = self.tri_e2.items.ptr, .tri_n1 = self.tri_n1.items.ptr, .tri_n2 = self.tri_n2.items.ptr,
            .tri_n3 = self.tri_n3.items.ptr,
            .n_materials = @intCast(self.mat_pattern.items.len), .mat_params = self.mat_params.items.ptr,
            .mat_pattern = self.mat_pattern.items.ptr,
            .n_patterns = @intCast(self.pat_kind.items.len), .pat_kind = self.pat_kind.items.ptr, .pat_inv = self.pat_inv.items.ptr,
            .pat_rgb = self.pat_rgb.items.ptr, .pat_a = self.pat_a.items.ptr, .pat_b = self.pat_b.items.ptr,
            .n_nodes = @intCast(self.node_first.items.len), .node_min = self.node_min.items.ptr, .node_max = self.node_max.items.ptr,
            .node_first = self.node_first.items.ptr, .node_count = self.node_count.items.ptr, .node_op = self.node_op.items.ptr,
            .n_children = @intCast(self.children.items.len), .children = self.children.items.ptr,
            .n_roots = @intCast(self.roots.items.len), .roots = self.roots.items.ptr,
            .n_lights = @intCast(self.light_pos.items.len / 3), .light_pos = self.light_pos.items.ptr,
            .light_rgb = self.light_rgb.items.ptr,
            .n_texmaps = @intCast(self.tex_mapping.items.len), .tex_mapping = self.tex_mapping.items.ptr, .tex_uv = self.tex_uv.items.ptr,
            .n_uvs = @intCast(self.uv_kind.items.len), .uv_kind = self.uv_kind.items.ptr, .uv_size = self.uv_size.items.ptr,
            .uv_sub = self.uv_sub.items.ptr, .uv_image = self.uv_image.items.ptr, .uv_interp = self.uv_interp.items.ptr,
            .n_images = @intCast(self.img_width.items.len), .img_width = self.img_width.items.ptr,
            .img_height = self.img_height.items.ptr, .img_offset = self.img_offset.items.ptr, .img_rgb = self.img_rgb.items.ptr,
        };
    }
};

/// Camera(T) after Camera.new + setTransform (camera.zig:18-61).
pub fn flattenCamera(camera: Camera(f64)) RtcCamera {
    var cam = RtcCamera{
        .hsize = @intCast(camera.hsize), .vsize = @intCast(camera.vsize),
        .half_width = camera.half_width, .half_height = camera.half_height, .pixel_size = camera.pixel_size,
        .inv_view = undefined,
    };
    for (camera._inverse_transform.data, 0..) |row, r| {
        for (row, 0..) |v, c| cam.inv_view[r * 4 + c] = v;
    }
    return cam;
}

// ------------------------------------------------------------------------------------------------ Camera.render

/// Replaces the body of `Camera(T).render(self, allocator, world) !Canvas(T)` (camera.zig:80-125) for T == f64 (scenes
/// render in f64: main.zig:71, lib.zig:194).  `n_gpus` > 1 takes the multi-GPU library instead.
pub fn render(camera: Camera(f64), allocator: Allocator, world: World(f64), n_gpus: u32) !Canvas(f64) {
    var image = try Canvas(f64).new(allocator, camera.hsize, camera.vsize);
    errdefer image.destroy();

    var flat = Flattener.init(allocator);
    defer flat.deinit();
    for (world.objects.items) |*object| try flat.addRoot(object);
    for (world.lights.items) |light| try flat.addLight(light);
    const desc = flat.desc();
    var cam = flattenCamera(camera);

    const rgb = try allocator.alloc(f64, 3 * camera.hsize * camera.vsize);
    defer allocator.free(rgb);
    if (n_gpus <= 1) {
        var scene: ?*RtcScene = null;
        if (rtc_scene_create(&desc, &scene) != 0) { // the name of the failure: rtc_last_error()
            std.log.err("{s}", .{rtc_last_error()});
            return GpuError.GpuSceneRejected;
        }
        defer rtc_scene_destroy(scene);
        if (rtc_render(scene.?, &cam, 5, 0, 0, cam.hsize, cam.vsize, rgb.ptr) != 0) { // depth 5: camera.zig:118
            std.log.err("{s}", .{rtc_last_error()});
            return GpuError.GpuRenderFailed;
        }
    } else {
        var multi: ?*RtcMulti = null;
        if (rtc_multi_create(&desc, n_gpus, 0, &multi) != 0) {
            std.log.err("{s}", .{rtc_multi_last_error()});
            return GpuError.GpuSceneRejected;
        }
        defer rtc_multi_destroy(multi);
        if (rtc_multi_render(multi.?, &cam, 5, rgb.ptr) != 0) {
            std.log.err("{s}", .{rtc_multi_last_error()});
            return GpuError.GpuRenderFailed;
        }
    }
    // Color(T) / Tuple(T) are packed structs whose @sizeOf is not 3 * 8 bytes (SURVEY section 7): copy element-wise.
    for (image.pixels, 0..) |*pixel, i| pixel.* = Color(f64).new(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
    return image;
}
