"""Parity of the HIP render path against the CPU oracle, through the C ABI (include/rtc.h).

Tolerance: BASELINE.json's north_star — per-channel |delta colour| < 1e-5 (f64).  The kernel keeps
the reference's evaluation order with FMA contraction off, so geometry is bit-identical and only
pow() (specular, schlick) can differ in the last ulps; the measured max delta is printed.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob

pytestmark = pytest.mark.gpu

TOL = 1e-5
# Two renders of the same frame agree to the last bits, not bitwise: a pixel whose ray tree is shared
# between lanes is the SUM of the lanes' shares in completion order (f64 atomic adds).
REPEAT_TOL = 1e-13

# (scene, width, height, depth): sizes the oracle finishes in seconds
CASES = [
    ("fresnel.json", 300, 300, 5),                      # BASELINE configs[0]
    ("cover.json", 192, 108, 5),                        # configs[1] at 1/10 scale
    ("cover.json", 160, 200, 5),                        # aspect < 1 branch of Camera.new
    ("reflection_and_refraction.json", 192, 108, 8),    # configs[2]: depth 8 extension
    ("reflection_and_refraction.json", 96, 54, 5),
    ("teapot.json", 192, 108, 5),                       # configs[3]: OBJ triangles + BVH
    ("dragons.json", 192, 108, 5),                      # configs[4]: smooth triangles, nested groups
    ("cubes.json", 150, 75, 5),
    ("cylinders.json", 160, 80, 5),
    ("groups.json", 150, 50, 5),                        # cones + cylinder in divided groups
    ("xyz.json", 160, 90, 5),                           # gradient + rings patterns, shininess 1600
    ("perturb_demo.json", 160, 90, 5),                  # perturb (Perlin noise) over every other pattern kind
    ("nefertiti.json", 90, 150, 5),                     # 99 944-triangle OBJ, perturbed gradient
    ("csg.json", 160, 90, 5),                           # three nested csg levels, unbounded cylinders
    ("csg_demo.json", 160, 90, 5),                      # csg in a group, group in a csg, glass lens, stale csg box
    ("align_check.json", 200, 100, 5),                  # cubic texture maps, align-check uv patterns
    ("earth.json", 200, 100, 5),                        # spherical map of a PNG, bilinear
    ("texture_demo.json", 160, 90, 5),                  # all four mappings, uv checkers / images, maps under other patterns
    ("skybox_demo.json", 200, 100, 5),                  # skybox.json's structure: camera inside a 1000x cube of image faces
    ("cover.json", 33, 17, 0),                          # depth 0: no secondary rays at all
    ("fresnel.json", 17, 33, 1),
]


def _both(rtc, scene, w, h, depth, tile=None):
    hs = rtc.HostScene.from_file(scene)
    cam = hs.camera(w, h)
    gpu = rtc.GpuScene(hs.desc)
    got = gpu.render(cam, depth, tile)
    stats = gpu.stats()
    want, counters = ob.OracleScene(hs.desc).render(cam, depth, tile)
    return got, want, stats, counters


@pytest.mark.parametrize("scene,w,h,depth", CASES)
def test_scene_parity(rtc, scene, w, h, depth):
    got, want, stats, counters = _both(rtc, scene, w, h, depth)
    delta = np.abs(got - want)
    worst = np.unravel_index(np.argmax(delta), delta.shape)
    print(f"{scene} {w}x{h} d{depth}: max|delta|={delta.max():.3e} at {worst}, "
          f"bit-identical pixels={np.mean(np.all(got == want, axis=2)) * 100:.2f}%")
    assert np.isfinite(got).all()
    assert delta.max() < TOL, (scene, delta.max(), worst, got[worst[:2]], want[worst[:2]])
    # ray counters are deterministic per scene/res/depth and must agree exactly
    assert stats["overflow"] == 0
    assert stats["primary"] == counters["primary"] == w * h
    assert stats["secondary"] == counters["secondary"]
    assert stats["shadow_calls"] == counters["shadow"]
    assert stats["shadow_traced"] <= stats["shadow_calls"]


@pytest.mark.parametrize("scene,w,h,depth", [("dragons.json", 96, 54, 5), ("teapot.json", 96, 54, 5), ("groups.json", 90, 30, 5),
                                             ("csg_demo.json", 96, 54, 5), ("cover.json", 96, 54, 5)])
def test_gpu_against_the_world_the_oracle_built_itself(rtc, scene, w, h, depth):
    """Everywhere else in this file the oracle is fed the PRODUCT loader's description (common mode: a loader bug moves
    both sides).  Here the oracle builds its own World from the scene JSON and OBJ bytes (oracle/rtc_oracle_scene.hpp,
    no product code) and that render is what the GPU - fed the product loader's description - must match.  The two
    builds are also compared table by table on the CPU (tests/test_oracle_scene_cpu.py)."""
    import os
    with open(os.path.join(rtc.SCENE_DIR, scene)) as f:
        js = f.read()
    hs = rtc.HostScene(js, rtc.DATA_DIR + os.sep)
    gpu = rtc.GpuScene(hs.desc)
    got = gpu.render(hs.camera(w, h), depth)
    stats = gpu.stats()
    built = ob.BuiltScene(js, rtc.DATA_DIR + os.sep, w, h)
    want, counters = built.render(depth)
    delta = np.abs(got - want)
    print(f"{scene} {w}x{h}: GPU vs oracle-built world max|delta|={delta.max():.3e}")
    assert delta.max() < TOL
    assert stats["overflow"] == 0 and stats["primary"] == counters["primary"] == w * h
    assert stats["secondary"] == counters["secondary"] and stats["shadow_calls"] == counters["shadow"]


def test_tile_render_matches_full_frame(rtc):
    hs = rtc.HostScene.from_file("cover.json")
    cam = hs.camera(120, 90)
    gpu = rtc.GpuScene(hs.desc)
    full = gpu.render(cam, 5)
    for tile in [(0, 0, 120, 90), (7, 3, 50, 41), (119, 89, 1, 1), (0, 80, 120, 10)]:
        x0, y0, w, h = tile
        part = gpu.render(cam, 5, tile)
        assert np.abs(part - full[y0:y0 + h, x0:x0 + w]).max() < REPEAT_TOL, tile


def test_interleaved_tiles_reassemble(rtc):
    """The multi-GPU partition on one GPU: render each rank's tiles, un-permute."""
    torch = pytest.importorskip("torch")
    hs = rtc.HostScene.from_file("fresnel.json")
    cam = hs.camera(100, 70)
    gpu = rtc.GpuScene(hs.desc)
    full = gpu.render(cam, 5)
    tw, th, world = 32, 16, 3
    tx, ty = rtc.tile_grid(cam.hsize, cam.vsize, tw, th)
    n_tiles = tx * ty
    padded = (n_tiles + world - 1) // world
    gathered = np.zeros((world, padded, th, tw, 3))
    for rank in range(world):
        first, stride, count, _ = rtc.tiles_of_rank(n_tiles, rank, world)
        buf = torch.zeros((padded, th, tw, 3), dtype=torch.float64, device="cuda")
        gpu.render_tiles_device(cam, buf.data_ptr(), tw, th, first, stride, count, 5,
                                torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        gathered[rank] = buf.cpu().numpy()
    img = rtc.assemble_tiles(gathered, cam.hsize, cam.vsize, tw, th, world)
    assert np.abs(img - full).max() < REPEAT_TOL
    # the device-side un-permute rank 0 runs after the gather (rtc_assemble_tiles_device): same bytes
    stream = torch.cuda.Stream()
    d_gathered = torch.from_numpy(gathered).cuda()
    d_canvas = torch.full((cam.vsize, cam.hsize, 3), -1.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    rtc.assemble_tiles_device(d_gathered.data_ptr(), world, padded, tw, th, cam.hsize, cam.vsize,
                              d_canvas.data_ptr(), stream.cuda_stream)
    stream.synchronize()
    assert np.array_equal(d_canvas.cpu().numpy(), img)
    with pytest.raises(rtc.RtcError) as e:   # too few tiles for the image
        rtc.assemble_tiles_device(d_gathered.data_ptr(), world, 1, tw, th, cam.hsize, cam.vsize,
                                  d_canvas.data_ptr(), stream.cuda_stream)
    assert e.value.name == "InvalidArgument"


def test_repeat_renders_agree(rtc):
    hs = rtc.HostScene.from_file("reflection_and_refraction.json")
    cam = hs.camera(96, 54)
    gpu = rtc.GpuScene(hs.desc)
    a = gpu.render(cam, 5)
    b = gpu.render(cam, 5)
    assert np.abs(a - b).max() < REPEAT_TOL
    # a scene whose ray trees never branch (no transparent material) has nothing to share: bitwise equal
    hs2 = rtc.HostScene.from_file("groups.json")
    gpu2 = rtc.GpuScene(hs2.desc)
    cam2 = hs2.camera(150, 50)
    assert np.array_equal(gpu2.render(cam2, 5), gpu2.render(cam2, 5))


def test_canvas_is_fully_written(rtc):
    """The library never clears the caller's canvas: every pixel is stored once, or zeroed by the lane that first
    shares its ray tree and then added to.  Rendering over NaNs must give the same image, launch after launch
    (first launch: heuristic schedule; second: re-packed; third: steady state)."""
    torch = pytest.importorskip("torch")
    for scene, w, h, depth in (("reflection_and_refraction.json", 320, 180, 5), ("fresnel.json", 150, 150, 5),
                               ("cover.json", 97, 61, 5)):
        hs = rtc.HostScene.from_file(scene)
        cam = hs.camera(w, h)
        gpu = rtc.GpuScene(hs.desc)
        want, _ = ob.OracleScene(hs.desc).render(cam, depth)
        for launch in range(3):
            canvas = torch.full((h, w, 3), float("nan"), dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            gpu.render_device(cam, canvas.data_ptr(), depth, None, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            got = canvas.cpu().numpy()
            assert np.isfinite(got).all(), (scene, launch, int(np.isnan(got).sum()))
            assert np.abs(got - want).max() < TOL, (scene, launch)
    # tile mode: the pixels of the rank's tiles inside the image are all written; the padding is left alone
    hs = rtc.HostScene.from_file("fresnel.json")
    cam = hs.camera(100, 70)
    gpu = rtc.GpuScene(hs.desc)
    full = gpu.render(cam, 5)
    tw = th = 32
    tx, ty = rtc.tile_grid(cam.hsize, cam.vsize, tw, th)
    buf = torch.full((tx * ty, th, tw, 3), float("nan"), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    gpu.render_tiles_device(cam, buf.data_ptr(), tw, th, 0, 1, tx * ty, 5, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    tiles = buf.cpu().numpy()
    for t in range(tx * ty):
        y0, x0 = (t // tx) * th, (t % tx) * tw
        hh, ww = min(th, cam.vsize - y0), min(tw, cam.hsize - x0)
        assert np.abs(tiles[t, :hh, :ww] - full[y0:y0 + hh, x0:x0 + ww]).max() < REPEAT_TOL, t
        assert np.isnan(tiles[t, hh:]).all() and np.isnan(tiles[t, :, ww:]).all()


def test_group_boxes_that_do_not_nest(rtc):
    """The chain replay stops at the innermost Group box only if rtc_scene_create finds every box inside its
    parent's.  A description whose tables break that (here: inner boxes of groups.json and teapot.json blown up or
    parents shrunk, so that rays pass an inner box and miss an outer one) must take the full replay and still match
    the oracle, which tests every box on the way down like the reference (group.zig:46-50)."""
    import ctypes as C
    for scene, w, h in (("groups.json", 150, 50), ("teapot.json", 96, 54)):
        hs = rtc.HostScene.from_file(scene)
        d = type(hs.desc)()  # a shallow copy: same arrays, except the two replaced below
        C.memmove(C.byref(d), C.byref(hs.desc), C.sizeof(d))
        n = d.n_nodes
        assert n > 3
        lo = np.ctypeslib.as_array(d.node_min, (n, 3)).copy()
        hi = np.ctypeslib.as_array(d.node_max, (n, 3)).copy()
        rng = np.random.default_rng(7)
        finite = np.isfinite(lo).all(axis=1) & np.isfinite(hi).all(axis=1)
        picks = rng.choice(np.flatnonzero(finite), size=max(2, int(finite.sum()) // 3), replace=False)
        for k in picks:  # shrink some boxes towards their centre: their children now stick out
            c = 0.5 * (lo[k] + hi[k])
            lo[k] = c + 0.6 * (lo[k] - c)
            hi[k] = c + 0.6 * (hi[k] - c)
        d.node_min = lo.ctypes.data_as(C.POINTER(C.c_double))
        d.node_max = hi.ctypes.data_as(C.POINTER(C.c_double))
        cam = hs.camera(w, h)
        gpu = rtc.GpuScene(d)
        got = gpu.render(cam, 5)
        want, counters = ob.OracleScene(d).render(cam, 5)
        base, _ = ob.OracleScene(hs.desc).render(cam, 5)
        assert np.abs(want - base).max() > 1e-3, "the doctored boxes must change the image, or the test tests nothing"
        assert np.abs(got - want).max() < TOL, scene
        st = gpu.stats()
        assert st["overflow"] == 0 and st["secondary"] == counters["secondary"] and st["shadow_calls"] == counters["shadow"]


def test_default_world_kat_through_the_abi(rtc):
    """camera.zig:171-187: default world, 11x11, fov pi/2, from (0,0,-5): pixel (5,5) = (0.38066, 0.47583, 0.2855)."""
    scene = """{"camera":{"width":11,"height":11,"field-of-view":1.5707963267948966,"from":[0,0,-5],"to":[0,0,0],"up":[0,1,0]},
      "lights":[{"point-light":{"position":[-10,10,-10],"intensity":[1,1,1]}}],
      "objects":[{"type":{"sphere":{}},"material":{"pattern":{"type":{"solid":[0.8,1.0,0.6]}},"diffuse":0.7,"specular":0.2}},
                 {"type":{"sphere":{}},"transform":[{"scale":[0.5,0.5,0.5]}]}]}"""
    hs = rtc.HostScene(scene)
    img = rtc.GpuScene(hs.desc).render(hs.camera(), 5)
    assert np.allclose(img[5, 5], [0.38066, 0.47583, 0.2855], atol=1e-5)


def test_test_pattern_refraction_kat(rtc):
    """world.zig:751-778 geometry through the ABI, GPU vs oracle (TestPattern, nested glass)."""
    scene = """{"camera":{"width":40,"height":40,"field-of-view":0.8,"from":[0,0,-3],"to":[0,0,0],"up":[0,1,0]},
      "lights":[{"point-light":{"position":[-10,10,-10],"intensity":[1,1,1]}}],
      "objects":[{"type":{"sphere":{}},"material":{"pattern":{"type":{"solid":[0.8,1.0,0.6]}},"ambient":1.0,"diffuse":0.7,"specular":0.2,
                   "transparency":0.9,"refractive-index":1.3,"reflective":0.3}},
                 {"type":{"sphere":{}},"transform":[{"scale":[0.5,0.5,0.5]}],"material":{"transparency":1.0,"refractive-index":1.5}},
                 {"type":{"cube":{}},"transform":[{"scale":[0.2,0.2,0.2]},{"translate":[0.1,0,0]}],"material":{"transparency":0.5,"reflective":0.5,"refractive-index":2.0}}]}"""
    hs = rtc.HostScene(scene)
    # make the outer sphere's pattern the reference's TestPattern (colour = pattern-space point)
    kinds = hs.array("pat_kind", hs.desc.n_patterns)
    mats = hs.array("mat_pattern", hs.desc.n_materials)
    kinds[mats[hs.array("leaf_material", hs.desc.n_leaves)[0]]] = 9
    cam = hs.camera()
    got = rtc.GpuScene(hs.desc).render(cam, 5)
    want, _ = ob.OracleScene(hs.desc).render(cam, 5)
    assert np.abs(got - want).max() < TOL


def test_full_size_cover_properties(rtc):
    """BASELINE configs[1] at full size (1920x1080): size-independent properties + the WHOLE image against the oracle."""
    hs = rtc.HostScene.from_file("cover.json")
    cam = hs.camera(1920, 1080)
    gpu = rtc.GpuScene(hs.desc)
    full = gpu.render(cam, 5)
    st = gpu.stats()
    assert gpu.last_kernel_name() == _simple_names(hs)[1]   # (what bench.py times)
    assert st["primary"] == 1920 * 1080 and st["overflow"] == 0
    assert np.isfinite(full).all() and full.min() >= 0.0
    # idempotence
    assert np.abs(full - gpu.render(cam, 5)).max() < REPEAT_TOL
    # tiles of the frame equal the frame (pixels are independent: camera.zig:116-121)
    for tile in [(640, 360, 333, 211), (0, 1079, 1920, 1)]:
        x0, y0, w, h = tile
        assert np.abs(gpu.render(cam, 5, tile) - full[y0:y0 + h, x0:x0 + w]).max() < REPEAT_TOL
    # depth linearity: depth 0 equals the surface term only, and is <= full colour where all weights >= 0
    d0 = gpu.render(cam, 0)
    assert gpu.stats()["secondary"] == 0
    assert (full - d0 >= -1e-12).all()
    # every pixel and every ray counter against the oracle (0.7 s on the box's 16 cores)
    want, counters = ob.OracleScene(hs.desc).render(cam, 5)
    assert np.abs(full - want).max() < TOL
    assert [st["secondary"], st["shadow_calls"]] == [counters["secondary"], counters["shadow"]]


# ---------------------------------------------------------------- the launches the bench times
# One handle renders a frame three times: launch 1 runs the geometric heuristic schedule and measures, launch 2 runs
# the schedule packed from those measurements (whole chunks, or runs of pixels where a chunk exceeds a wave's share),
# launch 3 is the steady state bench.py times.  Then the camera moves: the schedule is kept for a while, re-measured at
# launch 16 of the pixel map and re-packed by whichever launch finds the measurements on the host.  Pixels are
# independent (camera.zig:116-121), so EVERY one of these launches must give the oracle's image and ray counts.
SCHEDULE_CASES = [
    ("cover.json", 640, 360, 5),                        # rtc_render_kernel_simple, enough chunks for whole-chunk packing
    ("fresnel.json", 150, 150, 5),                      # simple kernel, fewer chunks than waves: chunks cut into runs
    ("reflection_and_refraction.json", 192, 108, 8),    # simple kernel, depth 8
    ("teapot.json", 192, 108, 5),                       # rtc_render_kernel (BVH)
    ("dragons.json", 192, 108, 5),                      # rtc_render_kernel, deep reference tree
    ("groups.json", 150, 50, 5),                        # cones in divided groups
    ("csg_demo.json", 160, 90, 5),                      # rtc_render_kernel_ext (csg, and texture-free groups)
    ("texture_demo.json", 160, 90, 5),                  # rtc_render_kernel_flat_ext (texture maps, no groups)
    ("cylinders.json", 160, 80, 5),                     # rtc_render_kernel_flat (cylinders at top level, no groups)
    ("skybox_demo.json", 200, 100, 5),                  # rtc_render_kernel_simple_ext (cubes and spheres with texture maps)
    ("earth.json", 200, 100, 5),                        # a texture-mapped sphere on a cylinder
]


def _check_launch(gpu, cam, depth, want, counters, what):
    got = gpu.render(cam, depth)
    st = gpu.stats()
    assert np.isfinite(got).all(), what
    assert np.abs(got - want).max() < TOL, (what, float(np.abs(got - want).max()))
    assert st["overflow"] == 0, what
    assert [st["primary"], st["secondary"], st["shadow_calls"]] == [counters["primary"], counters["secondary"], counters["shadow"]], what


@pytest.mark.parametrize("scene,w,h,depth", SCHEDULE_CASES)
def test_every_schedule_renders_the_same_image(rtc, scene, w, h, depth):
    hs = rtc.HostScene.from_file(scene)
    gpu = rtc.GpuScene(hs.desc)
    osc = ob.OracleScene(hs.desc)
    cam = hs.camera(w, h)
    want, counters = osc.render(cam, depth)
    for launch in range(1, 4):          # heuristic + measuring, packed, steady state
        _check_launch(gpu, cam, depth, want, counters, (scene, "launch", launch))
    hs.rotate_camera(0.3)               # lib.zig:166-178; the schedule in use now belongs to another view
    cam2 = hs.camera(w, h)
    want2, counters2 = osc.render(cam2, depth)
    assert np.abs(want2 - want).max() > 1e-3
    for launch in range(4, 30):         # launch 16 re-measures, a later one re-packs (at the latest 8 launches on)
        _check_launch(gpu, cam2, depth, want2, counters2, (scene, "moved camera, launch", launch))


@pytest.fixture
def simple3_always(rtc):
    """rtc_set_option("simple3_min_chunks", 0) for one test; the library's own choice (-1) afterwards."""
    rtc.set_option("simple3_min_chunks", 0)
    yield
    rtc.set_option("simple3_min_chunks", -1)


SIMPLE3_CASES = [("cover.json", 640, 360, 5), ("fresnel.json", 150, 150, 5), ("reflection_and_refraction.json", 192, 108, 8),
                 ("cubes.json", 200, 100, 5)]


@pytest.mark.parametrize("scene,w,h,depth", SIMPLE3_CASES)
def test_three_wave_simple_kernel_renders_the_same_image(rtc, scene, w, h, depth, simple3_always):
    """rtc_render_kernel_simple3 (168 VGPRs, three waves per SIMD) is what a large launch on a world of planes, spheres
    and cubes runs - cover.json at 1080p, the bench - while the small launches of this suite run the two-wave kernel.
    The option simple3_min_chunks = 0 (read per launch) makes every launch take it: the same launches as above, first
    frame on the estimate, packed, steady state, moved camera, each against the oracle."""
    hs = rtc.HostScene.from_file(scene)
    gpu = rtc.GpuScene(hs.desc)
    osc = ob.OracleScene(hs.desc)
    cam = hs.camera(w, h)
    want, counters = osc.render(cam, depth)
    for launch in range(1, 4):
        _check_launch(gpu, cam, depth, want, counters, (scene, "launch", launch))
        assert gpu.last_kernel_name() == _simple_names(hs)[1]
    hs.rotate_camera(0.3)
    cam2 = hs.camera(w, h)
    want2, counters2 = osc.render(cam2, depth)
    for launch in range(4, 12):
        _check_launch(gpu, cam2, depth, want2, counters2, (scene, "moved camera, launch", launch))
    rtc.set_option("simple3_min_chunks", 1e9)   # ... and back on the same handle: the two-wave kernel
    _check_launch(gpu, cam2, depth, want2, counters2, (scene, "two-wave kernel again"))
    assert gpu.last_kernel_name() == _simple_names(hs)[0]


def test_three_wave_simple_kernel_random_scenes(rtc, simple3_always):
    ran = 0
    for seed in range(40):
        hs = rtc.HostScene(_random_flat_scene(seed, simple=True, max_objects=3 + seed % 5))   # (few objects: nested patterns fill its 24-entry table)
        cam = hs.camera()
        gpu = rtc.GpuScene(hs.desc)
        got = gpu.render(cam, 5)
        fits = hs.desc.n_roots <= 32 and hs.desc.n_materials <= 16 and hs.desc.n_patterns <= 20 and hs.desc.n_lights <= 8   # RTC_LDS3_*
        assert gpu.last_kernel_name().startswith("rtc_render_kernel_simple3") == bool(fits), seed   # (larger worlds: other kernels; _b: mostly cubes)
        ran += fits
        want, counters = ob.OracleScene(hs.desc).render(cam, 5)
        st = gpu.stats()
        assert np.abs(got - want).max() < TOL, seed
        assert [st["secondary"], st["shadow_calls"], st["overflow"]] == [counters["secondary"], counters["shadow"], 0], seed
        gpu.close()
    assert ran >= 10


def _check_schedule(sched, n_chunks, what):
    """Every pixel of every chunk in exactly one item; returns the number of items that are runs (not whole chunks)."""
    seen = np.zeros((n_chunks, 64), dtype=np.int32)
    items = sched[sched != 0xFFFFFFFF]
    chunk, start, length = items & 0xFFFFF, (items >> 20) & 63, (items >> 26) + 1
    assert (chunk < n_chunks).all() and (start + length <= 64).all(), what
    for c, a, n in zip(chunk.tolist(), start.tolist(), length.tolist()):
        seen[c, a:a + n] += 1
    assert (seen == 1).all(), (what, int((seen != 1).sum()))
    return int((length < 64).sum())


SCHEDULE_SHAPES = [("cover.json", 1920, 1080, None, None), ("cover.json", 300, 200, None, True), ("fresnel.json", 150, 150, None, True),
                   ("reflection_and_refraction.json", 250, 130, None, True), ("teapot.json", 320, 180, None, True),
                   ("cover.json", 1920, 1080, (560, 720, 256, 256), True), ("cover.json", 1920, 1080, (3, 5, 131, 77), True)]


@pytest.mark.parametrize("scene,w,h,rect,expect_runs", SCHEDULE_SHAPES)
def test_device_schedules_hand_out_every_pixel_once(rtc, scene, w, h, rect, expect_runs):
    """The schedule is made on the device (rtc_kernels.hip: estimate or measurement -> classes -> sort -> packets, chunks
    above a wave's share cut into runs of pixels): read it back (rtc_get_schedule) after the first launch (packed from the
    estimate), the second (from the first frame's measurement) and after the camera moved (from a frame that ran CUT
    chunks, whose times are summed over their runs), and check its structure - every pixel of every 8x8 chunk in exactly
    one item, small launches actually cut - beside the image."""
    torch = pytest.importorskip("torch")
    hs = rtc.HostScene.from_file(scene)
    cam = hs.camera(w, h)
    gpu = rtc.GpuScene(hs.desc)
    x0, y0, rw, rh = rect if rect else (0, 0, w, h)
    n_chunks = ((rw + 7) // 8) * ((rh + 7) // 8)
    want = None
    if w * h <= 300 * 200:
        want, _ = ob.OracleScene(hs.desc).render(cam, 5)
    canvas = torch.empty((rh, rw, 3), dtype=torch.float64, device="cuda")
    runs = []
    for launch in range(4):
        if launch == 3:
            hs.rotate_camera(0.05)
            cam = hs.camera(w, h)
            want = None
        canvas.fill_(float("nan"))
        torch.cuda.synchronize()
        gpu.render_device(cam, canvas.data_ptr(), 5, rect, torch.cuda.current_stream().cuda_stream)
        st = gpu.stats()
        got = canvas.cpu().numpy()
        assert np.isfinite(got).all() and st["primary"] == rw * rh and st["overflow"] == 0, (scene, launch)
        if want is not None:
            assert np.abs(got - want[y0:y0 + rh, x0:x0 + rw]).max() < TOL, (scene, launch)
        sched = gpu.schedule()                       # what the NEXT launch would run
        assert len(sched) > 0
        runs.append(_check_schedule(sched, n_chunks, (scene, rect, launch)))
    if expect_runs:                                  # (fewer chunks than waves: most chunks exceed a wave's share)
        assert min(runs) > 0, runs


def test_every_order_of_the_packets_is_a_schedule(rtc):
    """Option "sched_mix" = a | b << 8: behind every wave's first packet the schedule takes a packets from its long end,
    then b from its short end, ... (rtc_pack_emit_kernel; 5 : 3 by default, 0 = longest first throughout).  Whatever the
    ratio - also lopsided ones that leave a long middle over - every pixel is in exactly one item and the image is the same."""
    torch = pytest.importorskip("torch")
    hs = rtc.HostScene.from_file("cover.json")
    w, h = 1280, 720                                 # 14 400 chunks: several packets per wave
    cam = hs.camera(w, h)
    n_chunks = ((w + 7) // 8) * ((h + 7) // 8)
    want = rtc.GpuScene(hs.desc).render(cam, 5)
    canvas = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
    try:
        for mix in (0, 1 | 1 << 8, 5 | 3 << 8, 2 | 7 << 8, 16 | 1 << 8, 255 | 255 << 8):
            rtc.set_option("sched_mix", mix)
            gpu = rtc.GpuScene(hs.desc)
            for launch in range(3):                  # estimate-packed, then packed from the measurement
                canvas.fill_(float("nan"))
                torch.cuda.synchronize()
                gpu.render_device(cam, canvas.data_ptr(), 5, None, torch.cuda.current_stream().cuda_stream)
                st = gpu.stats()
                assert st["primary"] == w * h and st["overflow"] == 0, (mix, launch)
                assert np.abs(canvas.cpu().numpy() - want).max() < REPEAT_TOL, (mix, launch)
                _check_schedule(gpu.schedule(), n_chunks, (mix, launch))
            gpu.close()
    finally:
        rtc.set_option("sched_mix", 5 | 3 << 8)


def test_first_launches_write_every_pixel(rtc):
    """The first frame of a pixel map runs a schedule packed from rtc_estimate_kernel's guesses (which roots a chunk's
    pixels can see), the second one packed from the first's measurements.  Whatever the guesses, every pixel is handed
    out exactly once: render over NaNs with every kind of kernel and in tile mode (edge tiles stick out of the image),
    and count the primary rays."""
    torch = pytest.importorskip("torch")
    for scene, w, h, depth in (("cover.json", 192, 108, 5), ("teapot.json", 96, 54, 5), ("csg_demo.json", 96, 54, 5),
                               ("reflection_and_refraction.json", 150, 97, 8)):
        hs = rtc.HostScene.from_file(scene)
        cam = hs.camera(w, h)
        gpu = rtc.GpuScene(hs.desc)
        want, counters = ob.OracleScene(hs.desc).render(cam, depth)
        for launch in range(1, 4):
            canvas = torch.full((h, w, 3), float("nan"), dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            gpu.render_device(cam, canvas.data_ptr(), depth, None, torch.cuda.current_stream().cuda_stream)
            st = gpu.stats()
            got = canvas.cpu().numpy()
            assert np.isfinite(got).all(), (scene, launch)
            assert np.abs(got - want).max() < TOL, (scene, launch)
            assert [st["primary"], st["secondary"], st["shadow_calls"], st["overflow"]] == [counters["primary"], counters["secondary"], counters["shadow"], 0]
    # tile mode (one rank's share of a frame): edge tiles stick out of the image
    hs = rtc.HostScene.from_file("fresnel.json")
    cam = hs.camera(200, 140)
    full = rtc.GpuScene(hs.desc).render(cam, 5)
    tw = th = 32
    tx, ty = rtc.tile_grid(cam.hsize, cam.vsize, tw, th)
    gpu = rtc.GpuScene(hs.desc)
    for launch in range(2):
        buf = torch.full((tx * ty, th, tw, 3), float("nan"), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        gpu.render_tiles_device(cam, buf.data_ptr(), tw, th, 0, 1, tx * ty, 5, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert gpu.stats()["primary"] == 200 * 140
        tiles = buf.cpu().numpy()
        for t in range(tx * ty):
            y0, x0 = (t // tx) * th, (t % tx) * tw
            hh, ww = min(th, cam.vsize - y0), min(tw, cam.hsize - x0)
            assert np.abs(tiles[t, :hh, :ww] - full[y0:y0 + hh, x0:x0 + ww]).max() < REPEAT_TOL, (launch, t)


def test_every_schedule_big_world_kernels(rtc):
    """The table-in-memory variants (rtc_render_kernel_bigworld and its _ext form) under packed schedules."""
    import json
    base = json.loads(_many_objects_scene(150))
    ext = json.loads(_many_objects_scene(150))
    ext["objects"].append({"type": {"csg": {"operation": "difference",
                                            "left": {"type": {"cube": {}}, "transform": [{"translate": [0, 1.2, 0]}]},
                                            "right": {"type": {"sphere": {}}, "transform": [{"scale": [1.3, 1.3, 1.3]}, {"translate": [0, 1.2, 0]}]}}}})
    for scene in (base, ext):
        hs = rtc.HostScene(json.dumps(scene))
        cam = hs.camera()
        gpu = rtc.GpuScene(hs.desc)
        want, counters = ob.OracleScene(hs.desc).render(cam, 5)
        for launch in range(1, 4):
            _check_launch(gpu, cam, 5, want, counters, ("big world", len(scene["objects"]), launch))


# BASELINE configs[2..4] at their full sizes (configs[1]: test_full_size_cover_properties): launches 1-3 on one handle,
# every k-th row of each against the oracle, and the packed launches against the first bit for bit where nothing is
# shared between lanes.
# Whole images, every pixel (the oracle takes 1.2 s, 1.6 s and 25 s for these on the box's 16 cores).
FULL_SIZE = [
    ("reflection_and_refraction.json", 1920, 1080, 8, 1),
    ("teapot.json", 1920, 1080, 5, 1),
    ("dragons.json", 3840, 2160, 5, 1),
]


@pytest.mark.parametrize("scene,w,h,depth,row_step", FULL_SIZE)
def test_full_size_configs(rtc, scene, w, h, depth, row_step):
    hs = rtc.HostScene.from_file(scene)
    cam = hs.camera(w, h)
    gpu = rtc.GpuScene(hs.desc)
    rows = np.arange(0, h, row_step)
    want, counters = ob.OracleScene(hs.desc).render(cam, depth, row_step=row_step)
    frames = []
    for launch in range(1, 4):
        got = gpu.render(cam, depth)
        st = gpu.stats()
        assert st["primary"] == w * h and st["overflow"] == 0
        if row_step == 1:
            assert [st["secondary"], st["shadow_calls"]] == [counters["secondary"], counters["shadow"]], (scene, launch)
        assert np.isfinite(got).all() and got.min() >= 0.0
        assert np.abs(got[rows] - want[rows]).max() < TOL, (scene, launch)
        frames.append((got, st))
    for got, st in frames[1:]:          # the schedule changes nothing: same image, same ray counts
        assert np.abs(got - frames[0][0]).max() < REPEAT_TOL
        assert st == frames[0][1]
    # a tile of the frame equals the frame
    x0, y0, tw, th = w // 3, h // 2, 257, 129
    assert np.abs(gpu.render(cam, depth, (x0, y0, tw, th)) - frames[0][0][y0:y0 + th, x0:x0 + tw]).max() < REPEAT_TOL


def test_full_size_tile_path(rtc):
    """The multi-GPU partition at the headline size on one GPU: cover 1920x1080 cut into 64x64 tiles dealt to 4 "ranks",
    each rank's tiles rendered three times (heuristic, packed, steady state), un-permuted on the device, compared with
    the plain render and with sampled oracle rows."""
    torch = pytest.importorskip("torch")
    hs = rtc.HostScene.from_file("cover.json")
    W, H, T, world = 1920, 1080, 64, 4
    cam = hs.camera(W, H)
    plain = rtc.GpuScene(hs.desc).render(cam, 5)
    tx, ty = rtc.tile_grid(W, H, T, T)
    n_tiles = tx * ty
    padded = (n_tiles + world - 1) // world
    stream = torch.cuda.Stream()
    for launch in range(3):
        gathered = torch.zeros((world, padded, T, T, 3), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        if launch == 0:
            scenes = [rtc.GpuScene(hs.desc) for _ in range(world)]   # one handle per rank, as in bench.py
        for rank in range(world):
            first, stride, count, _ = rtc.tiles_of_rank(n_tiles, rank, world)
            scenes[rank].render_tiles_device(cam, gathered[rank].data_ptr(), T, T, first, stride, count, 5, stream.cuda_stream)
        canvas = torch.full((H, W, 3), float("nan"), dtype=torch.float64, device="cuda")
        rtc.assemble_tiles_device(gathered.data_ptr(), world, padded, T, T, W, H, canvas.data_ptr(), stream.cuda_stream)
        stream.synchronize()
        got = canvas.cpu().numpy()
        assert np.abs(got - plain).max() < REPEAT_TOL, launch
    want, _ = ob.OracleScene(hs.desc).render(cam, 5, row_step=60)
    rows = np.arange(0, H, 60)
    assert np.abs(got[rows] - want[rows]).max() < TOL


def test_cost_balanced_tile_split(rtc):
    """The multi-GPU split by measured cost on one GPU: frame 1 dealt round-robin and measured, rtc_get_tile_costs,
    rtc_assign_tiles, then every "rank" renders its tile LIST; un-permuted by the table on the device; equal to the plain
    render, and better balanced than round-robin by the tiles' own measured costs."""
    torch = pytest.importorskip("torch")
    hs = rtc.HostScene.from_file("cover.json")
    W, H, T, world = 960, 540, 64, 4
    cam = hs.camera(W, H)
    plain = rtc.GpuScene(hs.desc).render(cam, 5)
    tx, ty = rtc.tile_grid(W, H, T, T)
    n_tiles = tx * ty
    padded = (n_tiles + world - 1) // world
    stream = torch.cuda.Stream()
    scenes = [rtc.GpuScene(hs.desc) for _ in range(world)]
    cost = np.zeros(n_tiles)
    for rank in range(world):                                # frame 1: round-robin, measured
        first, stride, count, _ = rtc.tiles_of_rank(n_tiles, rank, world)
        buf = torch.zeros((padded, T, T, 3), dtype=torch.float64, device="cuda")
        scenes[rank].render_tiles_device(cam, buf.data_ptr(), T, T, first, stride, count, 5, stream.cuda_stream)
        cost[first::stride] = scenes[rank].tile_costs(count)
    assert (cost > 0).all()
    rank_of, slot_of = rtc.assign_tiles(cost, world)
    rr_load = np.bincount(np.arange(n_tiles) % world, weights=cost, minlength=world)
    load = np.bincount(rank_of, weights=cost, minlength=world)
    assert load.max() < rr_load.max()
    d_slot = torch.from_numpy(slot_of.astype(np.int32)).cuda()
    for launch in range(3):
        gathered = torch.zeros((world, padded, T, T, 3), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        for rank in range(world):
            tiles = np.flatnonzero(rank_of == rank).astype(np.uint32)
            scenes[rank].render_tile_list_device(cam, gathered[rank].data_ptr(), T, T, tiles, 5, stream.cuda_stream)
        canvas = torch.full((H, W, 3), float("nan"), dtype=torch.float64, device="cuda")
        rtc.assemble_tile_list_device(gathered.data_ptr(), d_slot.data_ptr(), T, T, W, H, canvas.data_ptr(), stream.cuda_stream)
        stream.synchronize()
        assert np.abs(canvas.cpu().numpy() - plain).max() < REPEAT_TOL, launch
    with pytest.raises(rtc.RtcError):                        # a tile twice in a list
        scenes[0].render_tile_list_device(cam, gathered[0].data_ptr(), T, T, np.array([1, 1], dtype=np.uint32), 5, stream.cuda_stream)
    with pytest.raises(rtc.RtcError):                        # no measurement of that many regions
        scenes[0].tile_costs(3)


def test_single_process_multi_gpu_render(rtc):
    """include/rtc_multi.h on this one-GPU box: four VIRTUAL ranks (all on device 0, device copies in place of RCCL: the
    split, the gather layout, the un-permute and the re-balance by measured cost) and ONE real rank through RCCL
    (ncclCommInitAll + ncclGather).  Every frame is the oracle's image; the camera moves in between."""
    hs = rtc.HostScene.from_file("cover.json")
    osc = ob.OracleScene(hs.desc)
    for n, virtual in ((4, True), (3, True), (1, False)):
        hs = rtc.HostScene.from_file("cover.json")
        multi = rtc.MultiGpu(hs.desc, n, virtual)
        for frame in range(4):
            cam = hs.camera(400, 230)
            want, counters = osc.render(cam, 5)
            got = multi.render(cam, 5)
            st = multi.stats()
            assert np.abs(got - want).max() < TOL, (n, frame)
            assert [st["primary"], st["secondary"], st["shadow_calls"], st["overflow"]] == [counters["primary"], counters["secondary"], counters["shadow"], 0]
            tiles, ratio = multi.balance()
            assert tiles.sum() == 7 * 4                      # 400x230 in 64x64 tiles
            if n > 1 and frame >= 1:
                assert 0.99 < ratio < 1.6, (n, frame, ratio)  # re-dealt by measured cost after the first frame
            if frame == 1:
                hs.rotate_camera(0.4)
        other = hs.camera(130, 70)                           # another image size on the same object
        want_other = osc.render(other, 5)[0]
        assert np.abs(multi.render(other, 5) - want_other).max() < TOL
        # the RGBA8 framebuffer of the interactive seam (lib.zig:146-153), clamped on GPU 0 ...
        assert np.array_equal(multi.render_rgba8(other, 5), rtc.canvas_rgba8(multi.render(other, 5)))
        # ... and the frame left on the device: nothing copied, two canvases alternate
        ptrs = []
        for _ in range(3):
            ptrs.append(multi.render_device(other, 5))
            multi.synchronize()
            got = np.empty_like(want_other)
            import ctypes   # (hipMemcpy of the ONE runtime already in the process: the global symbol scope, not another dlopen)
            assert ctypes.CDLL(None).hipMemcpy(ctypes.c_void_p(got.ctypes.data), ctypes.c_void_p(ptrs[-1]), ctypes.c_size_t(got.nbytes), 2) == 0
            assert np.abs(got - want_other).max() < TOL
        assert ptrs[0] != ptrs[1] and ptrs[0] == ptrs[2]
        # an image no GPU can hold: refused (or out of memory), and the handle renders on afterwards
        huge = hs.camera(130, 70)
        huge.hsize, huge.vsize = 60000, 60000
        with pytest.raises(rtc.RtcError):
            multi.render_device(huge, 5)
        assert np.abs(multi.render(other, 5) - want_other).max() < TOL
        multi.close()


def test_frames_in_flight(rtc):
    """rtc_scene_clone and RTC_MULTI_FRAMES: independent frames on a handle and its clones (one device copy of the scene,
    a stream, schedule and counters per handle), enqueued without waiting for each other; the clone outlives the handle
    it was made from; four virtual ranks with three frame slots render an orbit three frames at a time."""
    torch = pytest.importorskip("torch")
    import ctypes
    hs = rtc.HostScene.from_file("teapot.json")          # (a mesh: the shared tables include the BVH)
    w, h = 240, 135
    osc = ob.OracleScene(hs.desc)
    first = rtc.GpuScene(hs.desc)
    handles = [first, first.clone(), first.clone()]
    streams = [torch.cuda.Stream() for _ in handles]
    cams, wants = [], []
    for k in range(3):
        cams.append(hs.camera(w, h))
        wants.append(osc.render(cams[-1], 5)[0])
        hs.rotate_camera(0.3)
    for round_ in range(3):
        outs = [torch.full((h, w, 3), float("nan"), dtype=torch.float64, device="cuda") for _ in handles]
        torch.cuda.synchronize()
        for k, g in enumerate(handles):                   # three frames in flight, three views
            g.render_device(cams[(k + round_) % 3], outs[k].data_ptr(), 5, None, streams[k].cuda_stream)
        torch.cuda.synchronize()
        for k in range(3):
            assert np.abs(outs[k].cpu().numpy() - wants[(k + round_) % 3]).max() < TOL, (round_, k)
    first.close()                                         # the clones keep the scene alive
    for k in (1, 2):
        assert np.abs(handles[k].render(cams[k], 5) - wants[k]).max() < TOL
        handles[k].close()

    # a scene with several handles is taken to have frames in flight: throughput over latency in the kernel choice
    hs = rtc.HostScene.from_file("cover.json")
    cam = hs.camera(720, 400)                             # 4500 chunks: between one and four per resident wave
    alone = rtc.GpuScene(hs.desc)
    want = alone.render(cam, 5)
    assert alone.last_kernel_name() == _simple_names(hs)[0]
    twin = alone.clone()
    for g in (alone, twin):
        assert np.abs(g.render(cam, 5) - want).max() < REPEAT_TOL and g.last_kernel_name() == _simple_names(hs)[1]
    twin.close()
    alone.render(cam, 5)
    assert alone.last_kernel_name() == _simple_names(hs)[0]
    alone.close()

    hs = rtc.HostScene.from_file("cover.json")
    osc = ob.OracleScene(hs.desc)
    multi = rtc.MultiGpu(hs.desc, 4, virtual=True, frames=3)
    seen = []
    for batch in range(4):                                # (the re-deal by measured cost happens in between)
        cams, ptrs = [], []
        for k in range(3):
            cams.append(hs.camera(400, 230))
            ptrs.append(multi.render_device(cams[-1], 5))
            hs.rotate_camera(0.05)
        multi.synchronize()
        for cam, ptr in zip(cams, ptrs):
            want = osc.render(cam, 5)[0]
            got = np.empty_like(want)
            assert ctypes.CDLL(None).hipMemcpy(ctypes.c_void_p(got.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(got.nbytes), 2) == 0
            assert np.abs(got - want).max() < TOL, batch
        seen.append(ptrs)
    assert len(set(seen[0])) == 3 and seen[0] == seen[1]  # three canvases take turns
    fb_ptrs, cams = [], []
    for k in range(3):                                    # the same as RGBA8 framebuffers: clamped by the rank that rendered the tile
        cams.append(hs.camera(400, 230))
        fb_ptrs.append(multi.render_rgba8_device(cams[-1], 5))
        hs.rotate_camera(0.05)
    multi.synchronize()
    for cam, ptr in zip(cams, fb_ptrs):
        got = np.empty((230, 400, 4), dtype=np.uint8)
        assert ctypes.CDLL(None).hipMemcpy(ctypes.c_void_p(got.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(got.nbytes), 2) == 0
        want = rtc.canvas_rgba8(multi.render(cam, 5))
        assert np.abs(got.astype(int) - want.astype(int)).max() <= 1 and (got != want).mean() < 1e-4   # (a channel at k + 0.5 within an ulp)
    assert len(set(fb_ptrs)) == 3
    tiles, ratio = multi.balance()
    assert tiles.sum() == 7 * 4 and 0.99 < ratio < 1.6
    assert np.abs(multi.render(cams[0], 5) - osc.render(cams[0], 5)[0]).max() < TOL   # the synchronous entry point on the same object
    multi.close()
    with pytest.raises(rtc.RtcError):
        rtc.MultiGpu(hs.desc, 2, virtual=True, frames=9)
    # one REAL rank with two slots: the RCCL form of the same (a comm stream that carries the gathers frame after frame,
    # the events between it and the slots' render streams)
    real = rtc.MultiGpu(hs.desc, 1, virtual=False, frames=2)
    cams, ptrs = [], []
    for k in range(4):
        cams.append(hs.camera(200, 120))
        ptrs.append(real.render_device(cams[-1], 5) if k % 2 == 0 else real.render_rgba8_device(cams[-1], 5))
        hs.rotate_camera(0.1)
        if k % 2 == 1:
            real.synchronize()
            want = osc.render(cams[-2], 5)[0]
            got = np.empty_like(want)
            assert ctypes.CDLL(None).hipMemcpy(ctypes.c_void_p(got.ctypes.data), ctypes.c_void_p(ptrs[-2]), ctypes.c_size_t(got.nbytes), 2) == 0
            assert np.abs(got - want).max() < TOL
            fb = np.empty((120, 200, 4), dtype=np.uint8)
            assert ctypes.CDLL(None).hipMemcpy(ctypes.c_void_p(fb.ctypes.data), ctypes.c_void_p(ptrs[-1]), ctypes.c_size_t(fb.nbytes), 2) == 0
            assert np.abs(fb.astype(int) - rtc.canvas_rgba8(osc.render(cams[-1], 5)[0]).astype(int)).max() <= 1
    real.close()


def test_launches_on_different_streams_are_ordered(rtc):
    """One handle, launches enqueued back to back on three different streams (a caller's, the handle's own through
    rtc_render, another caller's) without any host synchronisation in between: they share the handle's counters and
    scratch, so the library must order them (event + hipStreamWaitEvent); every canvas must be the oracle's image."""
    torch = pytest.importorskip("torch")
    hs = rtc.HostScene.from_file("reflection_and_refraction.json")
    w, h, depth = 320, 180, 5
    cam = hs.camera(w, h)
    want, counters = ob.OracleScene(hs.desc).render(cam, depth)
    gpu = rtc.GpuScene(hs.desc)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for round_ in range(4):
        a = torch.full((h, w, 3), float("nan"), dtype=torch.float64, device="cuda")
        b = torch.full((h, w, 3), float("nan"), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        gpu.render_device(cam, a.data_ptr(), depth, None, s1.cuda_stream)
        gpu.render_device(cam, b.data_ptr(), depth, None, s2.cuda_stream)
        c = gpu.render(cam, depth)                       # the handle's own stream, synchronous
        st = gpu.stats()
        torch.cuda.synchronize()
        for name, img in (("a", a.cpu().numpy()), ("b", b.cpu().numpy()), ("c", c)):
            assert np.isfinite(img).all(), (round_, name)
            assert np.abs(img - want).max() < TOL, (round_, name)
        assert [st["primary"], st["secondary"], st["shadow_calls"]] == [counters["primary"], counters["secondary"], counters["shadow"]]


def test_host_output_into_a_reused_canvas(rtc):
    """rtc_render into the same caller-owned buffer again and again (an interactive host), pageable and registered
    (rtc_canvas_register: explicit, tied to the memory, not to the handle); changing buffers and sizes in between must
    keep every image right, and so must a canvas that is freed and re-allocated at the same address between frames."""
    hs = rtc.HostScene.from_file("fresnel.json")
    gpu = rtc.GpuScene(hs.desc)
    cam = hs.camera(160, 120)
    want, _ = ob.OracleScene(hs.desc).render(cam, 5)
    a = np.full((120, 160, 3), np.nan)
    b = np.full((120, 160, 3), np.nan)
    rtc.canvas_register(a)
    for out in (a, a, a, b, a, b, b):
        out[:] = np.nan
        assert gpu.render_into(cam, out, 5) is out
        assert np.abs(out - want).max() < TOL
    rtc.canvas_unregister(a)
    with pytest.raises(rtc.RtcError):
        rtc.canvas_unregister(a)                          # not registered any more: an error, not a crash
    a[:] = np.nan
    gpu.render_into(cam, a, 5)                            # the same memory, pageable again
    assert np.abs(a - want).max() < TOL
    small = hs.camera(40, 30)
    for _ in range(3):                                    # a host that frees and re-allocates its canvas every frame
        c = np.full((30, 40, 3), np.nan)
        gpu.render_into(small, c, 5)
        assert np.abs(c - ob.OracleScene(hs.desc).render(small, 5)[0]).max() < TOL
        del c
    gpu.close()


def test_tile_mode_costs_when_the_image_is_not_a_multiple_of_the_tile(rtc):
    """Edge tiles reach past the image: their outside pixels are never rendered, and the per-pixel cost buffer the
    schedule and rtc_get_tile_costs are computed from must read zero there (it is cleared when it is allocated) - a
    tile's cost is then what its inside pixels took, and no packet of the schedule is made of outside chunks alone."""
    hs = rtc.HostScene.from_file("cover.json")
    torch = pytest.importorskip("torch")
    cam = hs.camera(200, 150)                              # 64x64 tiles: 4 x 3, the last column 8 wide, the last row 22 high
    tw = th = 64
    tx, ty = rtc.tile_grid(cam.hsize, cam.vsize, tw, th)
    want, _ = ob.OracleScene(hs.desc).render(cam, 5)
    costs = []
    for attempt in range(2):                               # two fresh handles: the same costs, not what the allocator left behind
        gpu = rtc.GpuScene(hs.desc)
        buf = torch.full((tx * ty, th, tw, 3), float("nan"), dtype=torch.float64, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        gpu.render_tiles_device(cam, buf.data_ptr(), tw, th, 0, 1, tx * ty, 5, stream)
        torch.cuda.synchronize()
        c = gpu.tile_costs(tx * ty)
        img = rtc.assemble_tiles(torch.nan_to_num(buf, nan=0.0).cpu().numpy()[None], cam.hsize, cam.vsize, tw, th, 1)
        assert np.abs(img - want).max() < TOL
        assert np.isfinite(c).all() and (c >= 0).all()
        costs.append(c)
        gpu.close()
    full, edge = costs[0].reshape(ty, tx)[0, 0], costs[0].reshape(ty, tx)[0, tx - 1]
    assert edge < 0.6 * full, (edge, full)                 # an 8-pixel-wide strip of sky costs less than a full tile of it
    ratio = costs[0] / np.maximum(costs[1], 1e-9)
    assert (ratio > 0.3).all() and (ratio < 3.0).all(), ratio   # (measured times: noisy, but not garbage)


def test_errors_through_the_abi(rtc):
    hs = rtc.HostScene.from_file("fresnel.json")
    gpu = rtc.GpuScene(hs.desc)
    cam = hs.camera(64, 64)
    with pytest.raises(rtc.RtcError) as e:
        gpu.render(cam, 5, (60, 0, 10, 10))   # tile outside the image
    assert e.value.name == "InvalidArgument"
    with pytest.raises(rtc.RtcError) as e:
        gpu.render(cam, 99)                    # deeper than the per-lane ray stack
    assert e.value.name == "InvalidArgument"


# ---------------------------------------------------------------- committed golden vectors
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden"))
from make_golden import CASES as GOLDEN_CASES, name_of as golden_name  # noqa: E402


@pytest.mark.parametrize("scene,w,h,depth", GOLDEN_CASES)
def test_gpu_matches_golden(rtc, scene, w, h, depth):
    g = np.load(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", golden_name(scene, w, h, depth)))
    hs = rtc.HostScene.from_file(scene)
    gpu = rtc.GpuScene(hs.desc)
    got = gpu.render(hs.camera(w, h), depth)
    assert np.abs(got - g["image"]).max() < TOL
    st = gpu.stats()
    assert [st["primary"], st["secondary"], st["shadow_calls"]] == g["counters"][:3].tolist()


def _many_objects_scene(n):
    """n top-level objects (> RTC_LDS_ROOTS = 128): exercises the table-in-memory kernel variant."""
    import json
    import math
    objs = [{"type": {"plane": {}}, "material": {"pattern": {"type": {"checkers": [
        {"type": {"solid": [0.2, 0.2, 0.2]}}, {"type": {"solid": [0.9, 0.9, 0.9]}}]}}, "reflective": 0.2}}]
    for i in range(n):
        a = 2.399963 * i
        r = 0.35 * math.sqrt(i + 1)
        kind = [{"sphere": {}}, {"cube": {}}, {"cylinder": {"min": -1, "max": 1, "closed": True}}][i % 3]
        objs.append({"type": kind,
                     "transform": [{"scale": [0.25, 0.25, 0.25]}, {"rotate-y": 0.1 * i},
                                   {"translate": [r * math.cos(a), 0.25 + 0.02 * (i % 7), r * math.sin(a)]}],
                     "material": {"pattern": {"type": {"solid": [(i % 5) / 5.0, (i % 3) / 3.0, (i % 7) / 7.0]}},
                                  "reflective": 0.3 if i % 4 == 0 else 0.0,
                                  "transparency": 0.6 if i % 9 == 0 else 0.0, "refractive-index": 1.4}})
    return json.dumps({"camera": {"width": 160, "height": 90, "field-of-view": 0.9, "from": [0, 4, -9], "to": [0, 0, 0],
                                  "up": [0, 1, 0]},
                       "lights": [{"point-light": {"position": [-6, 9, -6], "intensity": [1, 1, 1]}},
                                  {"point-light": {"position": [7, 6, -2], "intensity": [0.3, 0.3, 0.3]}}],
                       "objects": objs})


@pytest.mark.parametrize("n", [100, 200])
def test_many_top_level_objects(rtc, n):
    hs = rtc.HostScene(_many_objects_scene(n))
    assert hs.desc.n_roots == n + 1
    cam = hs.camera()
    gpu = rtc.GpuScene(hs.desc)
    got = gpu.render(cam, 5)
    want, counters = ob.OracleScene(hs.desc).render(cam, 5)
    assert np.abs(got - want).max() < TOL
    st = gpu.stats()
    assert st["secondary"] == counters["secondary"] and st["shadow_calls"] == counters["shadow"] and st["overflow"] == 0


def test_host_library_render_entry(rtc):
    """Camera.render(world) through the C++ host mirror (librtc_host -> librtc_hip), not the Python binding."""
    hs = rtc.HostScene.from_file("fresnel.json")
    out = np.empty((40, 40, 3))
    st = rtc.host_lib().rtch_scene_render(hs._h, 40, 40, 5, out.ctypes.data)
    assert st == 0, rtc.host_lib().rtch_last_error()
    want, _ = ob.OracleScene(hs.desc).render(hs.camera(40, 40), 5)
    assert np.abs(out - want).max() < TOL


def test_shapes_that_cast_no_shadow_inside_groups(rtc):
    """isShadowed (world.zig:136-147) counts an entry only if its shape casts a shadow.  Shadow traces skip such shapes
    without testing them: a top-level leaf by its flag, a leaf inside a group by its record's flag, and a whole leaf range
    of a group's BVH none of whose shapes casts one by a bit of the node (set at build time) - a display case around the
    other shapes, as in dragons.json, every second sphere, and a flagged shape beside an unflagged one in one group.
    The image (the shadows on the floor and between the shapes) and the counters are the oracle's on both general kernels."""
    import json
    def sphere(x, y, z, r, casts, **material):
        o = {"type": {"sphere": {}}, "transform": [{"scale": [r, r, r]}, {"translate": [x, y, z]}],
             "material": dict({"diffuse": 0.7, "specular": 0.3}, **material)}
        if not casts:
            o["casts-shadow"] = False
        return o
    kids = [sphere(-2.0 + 0.8 * (i % 6), 0.5 + 0.9 * (i // 6), 0.3 * (i % 3), 0.35, i % 2 == 0) for i in range(18)]
    case = {"type": {"cube": {}}, "transform": [{"scale": [3.2, 1.6, 1.2]}, {"translate": [0, 1.4, 0.3]}], "casts-shadow": False,
            "material": {"ambient": 0, "diffuse": 0.3, "specular": 0, "transparency": 0.7, "refractive-index": 1}}
    scene = {"camera": {"width": 160, "height": 100, "field-of-view": 1.0, "from": [1.5, 3.5, -8], "to": [0, 1, 0], "up": [0, 1, 0]},
             "lights": [{"point-light": {"position": [-6, 9, -6], "intensity": [0.7, 0.7, 0.7]}},
                        {"point-light": {"position": [7, 6, -3], "intensity": [0.4, 0.4, 0.4]}}],
             "objects": [{"type": {"plane": {}}, "material": {"diffuse": 0.8, "specular": 0}},
                         {"type": {"group": kids + [case]}},
                         {"type": {"cube": {}}, "transform": [{"scale": [0.5, 0.5, 0.5]}, {"translate": [3.5, 0.5, -1]}], "casts-shadow": False,
                          "material": {"diffuse": 0.6}},
                         {"type": {"group": [sphere(-3.5, 0.6, -1.5, 0.6, False), sphere(-3.3, 1.9, -1.5, 0.5, True)]}}]}
    hs = rtc.HostScene(json.dumps(scene))
    cam = hs.camera()
    want, counters = ob.OracleScene(hs.desc).render(cam, 5)
    for waves3 in (0, 1):
        rtc.set_option("waves3", waves3)
        try:
            gpu = rtc.GpuScene(hs.desc)
            got = gpu.render(cam, 5)
            st = gpu.stats()
            assert gpu.last_kernel_name() == ("rtc_render_kernel3" if waves3 else "rtc_render_kernel")
        finally:
            rtc.set_option("waves3", -1)
        assert np.abs(got - want).max() < TOL, waves3
        assert [st["overflow"], st["secondary"], st["shadow_calls"]] == [0, counters["secondary"], counters["shadow"]], waves3
    # the flags matter: with every shape casting, the image differs (the case and the flagged spheres throw shadows)
    for o in scene["objects"][1]["type"]["group"] + scene["objects"][2:3]:
        o.pop("casts-shadow", None)
    hs2 = rtc.HostScene(json.dumps(scene))
    assert np.abs(ob.OracleScene(hs2.desc).render(cam, 5)[0] - want).max() > 1e-3


@pytest.mark.parametrize("scene,w,h,depth", [("cover.json", 320, 180, 5), ("reflection_and_refraction.json", 240, 135, 8), ("cubes.json", 240, 135, 5),
                                            ("fresnel.json", 150, 150, 5), ("xyz.json", 200, 120, 5)])
def test_simple_worlds_on_boxes_and_on_spheres(rtc, scene, w, h, depth):
    """A simple world renders the oracle's image and counters whichever form of its kernels runs: roots rejected by their
    bounding spheres (rtc_render_kernel_simple / _simple3) or by their world boxes (_simple_b / _simple3_b) - forced
    either way by option, at two and at three waves per SIMD."""
    hs = rtc.HostScene.from_file(scene)
    cam = hs.camera(w, h)
    want, counters = ob.OracleScene(hs.desc).render(cam, depth)
    seen = set()
    for box in (0, 1):
        for three in (1e9, 0):
            rtc.set_option("box_cull", box)
            rtc.set_option("simple3_min_chunks", three)
            try:
                gpu = rtc.GpuScene(hs.desc)
                got = gpu.render(cam, depth)
                st = gpu.stats()
                seen.add(gpu.last_kernel_name())
            finally:
                rtc.set_option("box_cull", -1)
                rtc.set_option("simple3_min_chunks", -1)
            assert np.abs(got - want).max() < TOL, (scene, box, three)
            assert [st["overflow"], st["secondary"], st["shadow_calls"]] == [0, counters["secondary"], counters["shadow"]], (scene, box, three)
    assert seen == {"rtc_render_kernel_simple", "rtc_render_kernel_simple3", "rtc_render_kernel_simple_b", "rtc_render_kernel_simple3_b"}, seen


def _grazing_cubes_scene(variant):
    """Axis-aligned cubes whose faces, edges and corners lie exactly on pixel rays and on the shadow rays of a light, cubes
    stretched so far that the reference's "parallel" rule (cube.zig:28-35: a direction component below 1e-5 in object space
    is ignored) applies to ordinary rays, a rotated cube: what a box test that is tight on whole faces has to get right.
    variant: "simple" (top-level cubes and planes), "flat" (+ a cylinder), "group" (everything inside one group)."""
    import json
    def cube(sx, sy, sz, tx, ty, tz, extra=None, **material):
        o = {"type": {"cube": {}}, "transform": [{"scale": [sx, sy, sz]}] + (extra or []) + [{"translate": [tx, ty, tz]}],
             "material": dict({"diffuse": 0.7, "specular": 0.2, "reflective": 0.2}, **material)}
        return o
    objs = [
        cube(1, 1, 1, 1, 0, 0),                    # face x = 0: the image's centre column looks along it
        cube(1, 1, 1, -1, 2, 3),                   # edge x = 0, y = 1 behind it
        cube(0.5, 0.5, 0.5, 0, -1.5, -2),          # top face y = -1: the centre row's rays pass just above / along it
        cube(200000, 0.5, 0.5, 0, 3.5, 6),         # stretched 2e5 along x: d_obj_x = d_x / 2e5 < 1e-5 for every ray - the parallel rule
        cube(0.5, 200000, 0.5, -4, 0, 8),          # ... and along y
        cube(1, 1, 1, 3, 1, 4, extra=[{"rotate-y": 0.7853981633974483}, {"rotate-x": 0.6154797086703874}]),   # a corner towards the camera
        cube(2, 0.01, 2, 0, -3, 2, transparency=0.5, **{"refractive-index": 1.3}),   # a thin glass slab: entries 0.02 apart
    ]
    floor = {"type": {"plane": {}}, "transform": [{"translate": [0, -4, 0]}], "material": {"diffuse": 0.8, "specular": 0}}
    if variant == "flat":
        objs.append({"type": {"cylinder": {"min": -1, "max": 1, "closed": True}}, "transform": [{"translate": [-3, -2, 1]}]})
    if variant == "group":
        objs = [{"type": {"group": objs}}]
    # lights: in the plane of the first cube's face x = 0 and of the slab's top face; ordinary
    lights = [{"point-light": {"position": [0, 6, -3], "intensity": [0.5, 0.5, 0.5]}},
              {"point-light": {"position": [5, -2.99, -6], "intensity": [0.3, 0.3, 0.3]}},
              {"point-light": {"position": [-7, 8, -5], "intensity": [0.4, 0.4, 0.4]}}]
    return json.dumps({"camera": {"width": 121, "height": 81, "field-of-view": 1.2, "from": [0, 0, -10], "to": [0, 0, 0], "up": [0, 1, 0]},
                       "lights": lights, "objects": objs + [floor]})


@pytest.mark.parametrize("variant,kernel", [("simple", "rtc_render_kernel_simple_b"), ("flat", "rtc_render_kernel_flat"), ("group", "rtc_render_kernel")])
def test_rays_along_cube_faces(rtc, variant, kernel):
    """The root loop rejects World.objects entries (and the walk a group's leaves) by FP32 world boxes that are exactly as
    large as an axis-aligned cube: rays that run along faces, through edges and corners, shadow rays in the plane of a
    face, and cubes for which the reference ignores a direction component must all come out as in the oracle - whole image
    and every counter, on the two- and (where there is one) the three-wave kernel, odd image size (a pixel centre on the axis)."""
    hs = rtc.HostScene(_grazing_cubes_scene(variant))
    cam = hs.camera()
    want, counters = ob.OracleScene(hs.desc).render(cam, 5)
    three = {"simple": ("simple3_min_chunks", 0), "group": ("waves3", 1)}.get(variant)
    for force in (None, three):
        if force is None and three is None and False:
            continue
        if force is not None:
            rtc.set_option(force[0], force[1])
        try:
            gpu = rtc.GpuScene(hs.desc)
            got = gpu.render(cam, 5)
            st = gpu.stats()
            name = gpu.last_kernel_name()
        finally:
            if force is not None:
                rtc.set_option(force[0], -1)
        if force is None:
            assert name == kernel, name
        elif variant == "simple":
            assert name == "rtc_render_kernel_simple3_b", name
        else:
            assert name == "rtc_render_kernel3", name
        delta = np.abs(got - want)
        assert delta.max() < TOL, (variant, name, delta.max(), np.unravel_index(np.argmax(delta), delta.shape))
        assert [st["overflow"], st["primary"], st["secondary"], st["shadow_calls"]] == [0, counters["primary"], counters["secondary"], counters["shadow"]], (variant, name)
        if three is None:
            break


def _room_of_planes_scene(variant):
    """Planes where a plane test decides without its quotient (round 5: the division is skipped when the entry provably lies
    behind the ray or beyond the trace's limit): walls of a room around the lights (no shadow ray ever reaches one), a wall THROUGH
    a light (t equals the light's distance to the last bit or so), a plane through the camera's origin and along its central
    row (o.y = 0, d.y = 0 in the plane's space), planes a hair in front of and behind a ray's origin, a tilted mirror, a glass
    sheet (the containers pass looks at entries BEHIND the origin), and enough other shapes that the planes are the tail of
    a table with steps of four.  variant: "simple", "flat" (+ cylinder, cone), "group" (+ a group, so the general kernel runs)."""
    import json
    def plane(transform, **material):
        return {"type": {"plane": {}}, "transform": transform, "material": dict({"diffuse": 0.7, "specular": 0.3}, **material)}
    half_pi = 1.5707963267948966
    objs = [
        plane([{"translate": [0, -2, 0]}], reflective=0.3),                                   # floor
        plane([{"translate": [0, 9, 0]}]),                                                    # ceiling, through the first light
        plane([{"rotate-x": half_pi}, {"translate": [0, 0, 12]}], reflective=0.2),            # back wall
        plane([{"rotate-x": half_pi}, {"translate": [0, 0, -10]}]),                           # the wall the camera stands in
        plane([{"rotate-z": half_pi}, {"translate": [-8, 0, 0]}]),                            # left wall
        plane([{"rotate-z": half_pi}, {"translate": [8, 0, 0]}], transparency=0.6, **{"refractive-index": 1.2}),   # right wall: glass
        plane([{"translate": [0, 1e-9, 0]}], transparency=0.9, reflective=0.1, **{"refractive-index": 1.0}),       # a sheet a hair above y = 0
        plane([{"rotate-z": 0.3}, {"rotate-x": 0.2}, {"translate": [0, 4, 6]}], reflective=0.8, diffuse=0.1),      # tilted mirror
        {"type": {"sphere": {}}, "transform": [{"translate": [-2, 0, 2]}], "material": {"transparency": 0.8, "refractive-index": 1.5, "reflective": 0.3}},
        {"type": {"sphere": {}}, "transform": [{"scale": [0.5, 0.5, 0.5]}, {"translate": [2, -1.5, 0]}]},
        {"type": {"cube": {}}, "transform": [{"translate": [3, -1, 4]}], "material": {"reflective": 0.4}},
        {"type": {"cube": {}}, "transform": [{"scale": [0.5, 2, 0.5]}, {"translate": [-4, 0, 5]}]},
        {"type": {"sphere": {}}, "transform": [{"translate": [0, 3, 4]}], "casts-shadow": False},
    ]
    if variant in ("flat", "group"):
        objs.append({"type": {"cylinder": {"min": -2, "max": 1, "closed": True}}, "transform": [{"translate": [5, 0, 2]}]})
        objs.append({"type": {"cone": {"min": -1, "max": 0, "closed": True}}, "transform": [{"translate": [-5, -1, 0]}]})
    if variant == "group":
        objs.append({"type": {"group": [{"type": {"sphere": {}}, "transform": [{"translate": [1, 1, 1]}]},
                                        plane([{"translate": [0, -1.5, 0]}]),
                                        {"type": {"cube": {}}, "transform": [{"translate": [-1, 0, 7]}]}]}})
    lights = [{"point-light": {"position": [0, 9, 1], "intensity": [0.5, 0.5, 0.5]}},       # ON the ceiling plane
              {"point-light": {"position": [-3, 5, -4], "intensity": [0.4, 0.4, 0.4]}},
              {"point-light": {"position": [8, 2, 3], "intensity": [0.3, 0.3, 0.3]}}]        # ON the glass wall
    return json.dumps({"camera": {"width": 121, "height": 81, "field-of-view": 1.3, "from": [0, 0, -10], "to": [0, 0, 0], "up": [0, 1, 0]},
                       "lights": lights, "objects": objs})


@pytest.mark.parametrize("variant,kernel", [("simple", "rtc_render_kernel_simple"), ("flat", "rtc_render_kernel_flat"), ("group", "rtc_render_kernel")])
def test_planes_decided_without_their_quotient(rtc, variant, kernel):
    """Whole image and every counter against the oracle, on the sphere- and the box-culling simple kernels, two and three
    waves per SIMD, the flat and the general kernel (the planes are the tail of the World.objects table and phase 1 of the
    root loop does not look at them; a plane test skips its division where the entry cannot matter)."""
    hs = rtc.HostScene(_room_of_planes_scene(variant))
    cam = hs.camera()
    want, counters = ob.OracleScene(hs.desc).render(cam, 6)
    forms = {"simple": [(), (("box_cull", 1),), (("simple3_min_chunks", 0),), (("box_cull", 1), ("simple3_min_chunks", 0))],
             "flat": [()], "group": [(), (("waves3", 1),)]}[variant]
    names = set()
    for options in forms:
        for name, value in options:
            rtc.set_option(name, value)
        try:
            gpu = rtc.GpuScene(hs.desc)
            got = gpu.render(cam, 6)
            st = gpu.stats()
            names.add(gpu.last_kernel_name())
        finally:
            for name, _ in options:
                rtc.set_option(name, -1)
        if not options:
            assert gpu.last_kernel_name() == kernel, gpu.last_kernel_name()
        delta = np.abs(got - want)
        assert delta.max() < TOL, (variant, options, delta.max(), np.unravel_index(np.argmax(delta), delta.shape))
        assert [st["overflow"], st["primary"], st["secondary"], st["shadow_calls"]] == \
            [0, counters["primary"], counters["secondary"], counters["shadow"]], (variant, options)
    if variant == "simple":
        assert names == {"rtc_render_kernel_simple", "rtc_render_kernel_simple_b", "rtc_render_kernel_simple3", "rtc_render_kernel_simple3_b"}, names
    if variant == "group":
        assert names == {"rtc_render_kernel", "rtc_render_kernel3"}, names


def _rooms_scene(variant):
    """Cubes with every light inside (rtc_scene_create marks them as rooms: a shadow ray that starts inside one and whose
    light is nearer than the faces ahead skips the cube's six quotients): a big room, a tilted room inside it, a glass
    case around one light only (NOT a room: the other lights are outside), a light a hair from a wall, things on the
    walls, in the corners and outside the rooms.  variant: "simple", "flat" (+ a cylinder), "group" (+ a group)."""
    import json
    def cube(scale, translate, extra=None, **material):
        return {"type": {"cube": {}}, "transform": [{"scale": scale}] + (extra or []) + [{"translate": translate}],
                "material": dict({"diffuse": 0.7, "specular": 0.2}, **material)}
    objs = [
        cube([10, 6, 12], [0, 4, 0], pattern={"type": {"checkers": [{"type": {"solid": [0.9, 0.9, 0.9]}}, {"type": {"solid": [0.3, 0.3, 0.4]}}]},
                                              "transform": [{"scale": [0.1, 0.2, 0.1]}]}),               # the room
        cube([7, 5, 8], [0, 4, 1], extra=[{"rotate-y": 0.3}], transparency=0.7, reflective=0.1, **{"refractive-index": 1.05}),   # a tilted glass room inside it
        cube([0.5, 0.5, 0.5], [-3, 7, -2], transparency=0.9, **{"refractive-index": 1.0}),              # a case around the first light only
        cube([1, 1, 1], [9, -1, 11]),                                                                    # in a corner of the room
        cube([1, 0.2, 1], [0, -1.8, 0], reflective=0.5),                                                 # on the floor
        {"type": {"sphere": {}}, "transform": [{"translate": [2, 0, 2]}], "material": {"reflective": 0.3}},
        {"type": {"sphere": {}}, "transform": [{"scale": [0.7, 0.7, 0.7]}, {"translate": [-2, -1.3, 3]}], "material": {"transparency": 0.8, "refractive-index": 1.5}},
        {"type": {"sphere": {}}, "transform": [{"translate": [0, 30, 0]}]},                               # outside every room
        {"type": {"plane": {}}, "transform": [{"translate": [0, -2.5, 0]}]},                              # below the room's floor
    ]
    if variant in ("flat", "group"):
        objs.append({"type": {"cylinder": {"min": -2, "max": 2, "closed": True}}, "transform": [{"scale": [0.5, 1, 0.5]}, {"translate": [4, 0, -3]}]})
    if variant == "group":
        objs.append({"type": {"group": [cube([0.5, 0.5, 0.5], [-4, -1.5, 5]), {"type": {"sphere": {}}, "transform": [{"translate": [-5, 0, 6]}]}]}})
    lights = [{"point-light": {"position": [-3, 7, -2], "intensity": [0.5, 0.5, 0.5]}},
              {"point-light": {"position": [4, 5, -6], "intensity": [0.4, 0.4, 0.4]}},
              {"point-light": {"position": [0, 9.99999, 3], "intensity": [0.3, 0.3, 0.3]}}]    # a hair under the room's ceiling (y = 10)
    return json.dumps({"camera": {"width": 121, "height": 81, "field-of-view": 1.4, "from": [0, 3, -11], "to": [0, 2, 0], "up": [0, 1, 0]},
                       "lights": lights, "objects": objs})


@pytest.mark.parametrize("variant", ["simple", "flat", "group"])
def test_rooms_around_the_lights(rtc, variant):
    """Whole image and every counter against the oracle for worlds inside cubes that contain the lights, on every kernel
    form the world can run."""
    hs = rtc.HostScene(_rooms_scene(variant))
    cam = hs.camera()
    want, counters = ob.OracleScene(hs.desc).render(cam, 6)
    forms = {"simple": [(), (("box_cull", 0),), (("simple3_min_chunks", 0),), (("box_cull", 0), ("simple3_min_chunks", 0))],
             "flat": [()], "group": [(), (("waves3", 1),)]}[variant]
    for options in forms:
        for name, value in options:
            rtc.set_option(name, value)
        try:
            gpu = rtc.GpuScene(hs.desc)
            got = gpu.render(cam, 6)
            st = gpu.stats()
        finally:
            for name, _ in options:
                rtc.set_option(name, -1)
        delta = np.abs(got - want)
        assert delta.max() < TOL, (variant, options, gpu.last_kernel_name(), delta.max(), np.unravel_index(np.argmax(delta), delta.shape))
        assert [st["overflow"], st["primary"], st["secondary"], st["shadow_calls"]] == \
            [0, counters["primary"], counters["secondary"], counters["shadow"]], (variant, options)
        assert st["shadow_traced"] > 0


def _random_scene(seed):
    """Random world through the JSON loader: every in-scope primitive, nested groups (some large enough to be
    divided), planes inside groups, glass inside glass, every pattern kind the kernel implements."""
    import json
    import random
    rnd = random.Random(seed)

    def solid():
        return {"type": {"solid": [round(rnd.random(), 3) for _ in range(3)]}}

    def pattern(depth=0):
        kinds = ["solid", "solid", "stripes", "checkers", "rings", "gradient", "radial-gradient", "blend"]
        if seed > 16:   # later seeds add Perlin-noise perturbation (seeds 1..16 keep the scenes they always had)
            kinds.append("perturb")
        k = rnd.choice(kinds)
        if k == "solid" or depth >= 2:
            return solid()
        if k == "perturb":
            inner = pattern(depth + 1)
            return {"type": {"perturb": inner}, "transform": [{"scale": [rnd.uniform(0.3, 2.0)] * 3}]}
        sub = (lambda: solid()) if k in ("gradient", "radial-gradient", "blend") else (lambda: pattern(depth + 1))
        p = {"type": {k: [sub(), sub()]}}
        if rnd.random() < 0.7:
            p["transform"] = [{"scale": [rnd.uniform(0.2, 1.5)] * 3}, {"rotate-y": rnd.uniform(0, 3)}]
        return p

    def material():
        m = {"pattern": pattern(), "ambient": rnd.uniform(0, 0.4), "diffuse": rnd.uniform(0.2, 0.9),
             "specular": rnd.choice([0, 0.3, 0.9]), "shininess": rnd.choice([10, 200])}
        r = rnd.random()
        if r < 0.25:
            m.update({"reflective": rnd.uniform(0.1, 0.9)})
        elif r < 0.5:
            m.update({"reflective": rnd.uniform(0.1, 0.9), "transparency": rnd.uniform(0.3, 1.0),
                      "refractive-index": rnd.choice([1.0, 1.33, 1.5, 2.4])})
        elif r < 0.6:
            m.update({"transparency": rnd.uniform(0.3, 1.0), "refractive-index": 1.5})
        return m

    def transform(spread):
        return [{"scale": [rnd.uniform(0.3, 1.2) for _ in range(3)]},
                {"rotate-x": rnd.uniform(0, 6.28)}, {"rotate-z": rnd.uniform(0, 6.28)},
                {"translate": [rnd.uniform(-spread, spread), rnd.uniform(0, spread), rnd.uniform(-spread, spread)]}]

    def leaf(spread):
        k = rnd.choice(["sphere", "cube", "cylinder", "cone", "triangle", "sphere", "cube"])
        if k in ("sphere", "cube"):
            t = {k: {}}
        elif k == "triangle":
            t = {"triangle": {"p1": [0, 1, 0], "p2": [-1, 0, 0.2], "p3": [1, 0, -0.3]}}
        else:
            lo = rnd.uniform(-1.5, -0.2)
            t = {k: {"min": lo, "max": lo + rnd.uniform(0.5, 2.0), "closed": rnd.random() < 0.6}}
        o = {"type": t, "transform": transform(spread), "material": material()}
        if rnd.random() < 0.15:
            o["casts-shadow"] = False
        return o

    def csg(depth):
        def side():
            r = rnd.random()
            if depth < 2 and r < 0.25:
                return csg(depth + 1)
            if depth < 2 and r < 0.4:
                return group(depth + 1, rnd.randint(2, 4))
            return leaf(1.2)
        o = {"type": {"csg": {"operation": rnd.choice(["union", "intersection", "difference"]),
                              "left": side(), "right": side()}}}
        if rnd.random() < 0.3:   # a transform pushed through the csg: its box stays where it was built
            o["transform"] = [{"translate": [rnd.uniform(-0.4, 0.4), rnd.uniform(0, 0.3), rnd.uniform(-0.4, 0.4)]}]
        if rnd.random() < 0.4:
            o["material"] = material()
        return o

    def group(depth, n):
        kids = []
        for _ in range(n):
            kids.append(group(depth + 1, rnd.randint(2, 5)) if depth < 2 and rnd.random() < 0.25 else leaf(3.0))
        if depth == 0 and rnd.random() < 0.5:   # a plane inside a group: its box has infinite / NaN extents
            kids.append({"type": {"plane": {}}, "transform": [{"translate": [0, -2.5, 0]}], "material": material()})
        g = {"type": {"group": kids}}
        if rnd.random() < 0.8:
            g["transform"] = [{"rotate-y": rnd.uniform(0, 6.28)}, {"translate": [rnd.uniform(-2, 2), 0, rnd.uniform(-2, 2)]}]
        return g

    objs = [{"type": {"plane": {}}, "transform": [{"translate": [0, -3, 0]}], "material": material()},
            # nested glass: exercises the containers walk (n1/n2)
            {"type": {"sphere": {}}, "transform": [{"scale": [1.5, 1.5, 1.5]}, {"translate": [0, 0.5, 0]}],
             "material": {"transparency": 0.9, "reflective": 0.4, "refractive-index": 1.5, "diffuse": 0.1}},
            {"type": {"sphere": {}}, "transform": [{"scale": [0.7, 0.7, 0.7]}, {"translate": [0.2, 0.5, 0]}],
             "material": {"transparency": 0.9, "reflective": 0.2, "refractive-index": 1.0003, "diffuse": 0.1}}]
    objs += [leaf(4.0) for _ in range(rnd.randint(3, 8))]
    objs += [group(0, rnd.randint(6, 14)) for _ in range(rnd.randint(1, 3))]
    if seed > 22:   # later seeds add csg: top-level, and inside a group (seeds 1..22 keep their scenes)
        objs += [csg(0) for _ in range(rnd.randint(1, 3))]
        g = group(1, rnd.randint(2, 4))
        g["type"]["group"].append(csg(1))
        objs.append(g)
    lights = [{"point-light": {"position": [rnd.uniform(-8, 8), rnd.uniform(4, 10), rnd.uniform(-10, -4)],
                               "intensity": [rnd.uniform(0.3, 1.0)] * 3}} for _ in range(rnd.randint(1, 3))]
    return json.dumps({"camera": {"width": 96, "height": 64, "field-of-view": 1.0, "from": [rnd.uniform(-3, 3), 3, -9],
                                  "to": [0, 0.5, 0], "up": [0, 1, 0]}, "lights": lights, "objects": objs})


# 410: a cone inside a group reports an entry outside the group's box (found by tools/fuzz_more.py)
@pytest.mark.parametrize("seed", list(range(1, 31)) + [410])
def test_random_scenes(rtc, seed):
    hs = rtc.HostScene(_random_scene(seed))
    cam = hs.camera()
    gpu = rtc.GpuScene(hs.desc)
    got = gpu.render(cam, 5)
    want, counters = ob.OracleScene(hs.desc).render(cam, 5)
    delta = np.abs(got - want)
    print(f"seed {seed}: leaves {hs.desc.n_leaves} nodes {hs.desc.n_nodes} max|delta| {delta.max():.2e}")
    assert delta.max() < TOL, (seed, delta.max(), np.unravel_index(np.argmax(delta), delta.shape))
    st = gpu.stats()
    assert st["secondary"] == counters["secondary"] and st["shadow_calls"] == counters["shadow"] and st["overflow"] == 0


def _random_flat_scene(seed, simple, max_objects=60):
    """Random world WITHOUT groups: many top-level leaves with random transforms, glass inside glass, every pattern kind.
    simple: spheres, planes and cubes only (rtc_render_kernel_simple); else every leaf kind (rtc_render_kernel_flat)."""
    import json
    import random
    rnd = random.Random(1000 + seed)
    base = json.loads(_random_scene(40 + seed))          # materials / patterns / lights / camera of the grouped generator
    kinds = ["sphere", "cube", "plane"] if simple else ["sphere", "cube", "plane", "cylinder", "cone", "triangle"]

    def leaves(o):
        t = o["type"]
        if "group" in t:
            for c in t["group"]:
                yield from leaves(c)
        elif "csg" in t:
            yield from leaves(t["csg"]["left"])
            yield from leaves(t["csg"]["right"])
        else:
            yield o
    objs = []
    for o in base["objects"]:
        for leaf in leaves(o):
            kind = next(iter(leaf["type"]))
            if kind not in kinds:
                leaf = dict(leaf, type={rnd.choice(["sphere", "cube"]): {}})
            leaf = dict(leaf)
            leaf["transform"] = list(leaf.get("transform", [])) + [{"translate": [rnd.uniform(-3, 3), rnd.uniform(0, 2), rnd.uniform(-3, 3)]}]
            objs.append(leaf)
    objs = objs[:max_objects]
    objs.append({"type": {"plane": {}}, "transform": [{"rotate-x": 1.5707963}, {"translate": [0, 0, 12]}],
                 "material": {"reflective": 0.3, "pattern": {"type": {"checkers": [{"type": {"solid": [0.1, 0.1, 0.1]}}, {"type": {"solid": [0.9, 0.9, 0.9]}}]}}}})
    base["objects"] = objs
    return json.dumps(base)


@pytest.mark.parametrize("seed,simple", [(s, True) for s in range(1, 9)] + [(s, False) for s in range(1, 9)])
def test_random_scenes_without_groups(rtc, seed, simple):
    hs = rtc.HostScene(_random_flat_scene(seed, simple))
    assert hs.desc.n_nodes == 0
    cam = hs.camera()
    gpu = rtc.GpuScene(hs.desc)
    want, counters = ob.OracleScene(hs.desc).render(cam, 5)
    for launch in range(2):                                 # heuristic schedule, then the device-packed one
        got = gpu.render(cam, 5)
        delta = np.abs(got - want)
        assert delta.max() < TOL, (seed, simple, launch, delta.max(), np.unravel_index(np.argmax(delta), delta.shape))
        st = gpu.stats()
        assert st["secondary"] == counters["secondary"] and st["shadow_calls"] == counters["shadow"] and st["overflow"] == 0


def test_csg_lists_longer_than_the_first_allocation(rtc):
    """Csg.filterIntersections works on a list of any length (csg.zig:51-95).  A lane's list starts with 32 slots; a ray
    through 20 nested spheres and a cube needs 42.  The synchronous entry points grow the lists and render again - the
    image is the oracle's -; an asynchronous render on a fresh handle reports the overflow in its counters."""
    import json
    torch = pytest.importorskip("torch")
    spheres = [{"type": {"sphere": {}}, "transform": [{"scale": [0.2 + 0.05 * i] * 3}]} for i in range(20)]
    scene = {"camera": {"width": 32, "height": 32, "field-of-view": 0.6, "from": [0, 0, -6], "to": [0, 0, 0], "up": [0, 1, 0]},
             "lights": [{"point-light": {"position": [-5, 5, -5], "intensity": [1, 1, 1]}}],
             "objects": [{"type": {"csg": {"operation": "union", "left": {"type": {"group": spheres}},
                                           "right": {"type": {"cube": {}}}}}}]}
    hs = rtc.HostScene(json.dumps(scene))
    cam = hs.camera()
    want, counters = ob.OracleScene(hs.desc).render(cam, 5)
    gpu = rtc.GpuScene(hs.desc)
    buf = torch.zeros((32, 32, 3), dtype=torch.float64, device="cuda")
    gpu.render_device(cam, buf.data_ptr(), 5, None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert gpu.stats()["overflow"] > 0                      # 32 slots were not enough, and the launch says so
    got = gpu.render(cam, 5)                                # grows the lists, renders again
    st = gpu.stats()
    assert np.abs(got - want).max() < TOL
    assert [st["overflow"], st["secondary"], st["shadow_calls"]] == [0, counters["secondary"], counters["shadow"]]
    gpu.render_device(cam, buf.data_ptr(), 5, None, torch.cuda.current_stream().cuda_stream)   # the handle keeps them
    torch.cuda.synchronize()
    assert gpu.stats()["overflow"] == 0 and np.abs(buf.cpu().numpy() - want).max() < TOL
    assert np.array_equal(gpu.render_rgba8(cam, 5), rtc.canvas_rgba8(got))


def test_multi_gpu_host_output_written_by_every_rank(rtc):
    """rtc_multi_render / _render_rgba8 into a REGISTERED host canvas (rtc_canvas_register: pinned and mapped): every rank
    writes its own tiles straight to their places in the caller's canvas over its own host link (rtc_scatter_tile_list_*),
    no gather, nothing through GPU 0.  Virtual ranks, one and three frame slots: the bytes are those of the gathered path
    (an unregistered canvas) - exactly, the split is the same - and the oracle's image; the re-deal after the first frame
    and a second image size go through the same path."""
    hs = rtc.HostScene.from_file("cover.json")
    osc = ob.OracleScene(hs.desc)
    for frames in (1, 3):
        multi = rtc.MultiGpu(hs.desc, 4, True, frames)
        for (w, h) in ((400, 230), (130, 70)):
            cam = hs.camera(w, h)
            want = osc.render(cam, 5)[0]
            direct = np.full((h, w, 3), -1.0)
            direct8 = np.zeros((h, w, 4), dtype=np.uint8)
            rtc.canvas_register(direct)
            rtc.canvas_register(direct8)
            try:
                for frame in range(3):
                    direct[:] = -1.0
                    multi.render(cam, 5, direct)
                    st = multi.stats()
                    gathered = multi.render(cam, 5)                 # (a pageable canvas: the gather to GPU 0 and one copy)
                    assert np.abs(direct - want).max() < TOL and st["overflow"] == 0
                    assert np.abs(direct - gathered).max() < REPEAT_TOL
                    multi.render_rgba8(cam, 5, direct8)
                    assert np.array_equal(direct8, rtc.canvas_rgba8(gathered)) or \
                        np.abs(direct8.astype(int) - rtc.canvas_rgba8(gathered).astype(int)).max() <= 1   # (a channel on a rounding edge)
                    assert np.array_equal(direct8, multi.render_rgba8(cam, 5)) or \
                        np.abs(direct8.astype(int) - multi.render_rgba8(cam, 5).astype(int)).max() <= 1
            finally:
                rtc.canvas_unregister(direct)
                rtc.canvas_unregister(direct8)
        multi.close()


def test_multi_gpu_grows_csg_lists(rtc):
    """The csg scene of test_csg_lists_longer_than_the_first_allocation through rtc_multi (virtual ranks, two frame
    slots): a lane's list of 32 entries is too short for a ray through 20 nested spheres and a cube; the frame says how
    long it had to be, every handle that overflowed gets lists of that length (rtc_grow_csg_lists), and the synchronous
    entry points render again - the oracle's image, on every slot; the asynchronous form reports the overflow once and
    renders the frames after it with the longer lists."""
    import json
    spheres = [{"type": {"sphere": {}}, "transform": [{"scale": [0.2 + 0.05 * i] * 3}]} for i in range(20)]
    scene = {"camera": {"width": 96, "height": 96, "field-of-view": 0.6, "from": [0, 0, -6], "to": [0, 0, 0], "up": [0, 1, 0]},
             "lights": [{"point-light": {"position": [-5, 5, -5], "intensity": [1, 1, 1]}}],
             "objects": [{"type": {"csg": {"operation": "union", "left": {"type": {"group": spheres}},
                                           "right": {"type": {"cube": {}}}}}}]}
    hs = rtc.HostScene(json.dumps(scene))
    cam = hs.camera()
    want, counters = ob.OracleScene(hs.desc).render(cam, 5)
    multi = rtc.MultiGpu(hs.desc, 2, True, 2)
    for frame in range(4):                                       # (both slots, before and after the re-deal)
        got = multi.render(cam, 5)
        st = multi.stats()
        assert np.abs(got - want).max() < TOL, frame
        assert [st["overflow"], st["secondary"], st["shadow_calls"]] == [0, counters["secondary"], counters["shadow"]]
    assert np.array_equal(multi.render_rgba8(cam, 5), rtc.canvas_rgba8(multi.render(cam, 5)))
    multi.close()
    # asynchronous: the first frame's overflow is reported (once) when its slot is finished; the lists are longer then
    multi = rtc.MultiGpu(hs.desc, 2, True, 1)
    multi.render_device(cam, 5)
    with pytest.raises(rtc.RtcError) as e:
        multi.synchronize()
    assert "Overflow" in str(e.value) or "overflow" in str(e.value)
    for _ in range(3):                                           # (a handle may need a second step: it sizes by its own share)
        try:
            multi.render_device(cam, 5)
            multi.synchronize()
            break
        except rtc.RtcError:
            continue
    ptr = multi.render_device(cam, 5)
    multi.synchronize()
    got = np.empty_like(want)
    import ctypes
    assert ctypes.CDLL(None).hipMemcpy(ctypes.c_void_p(got.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(got.nbytes), 2) == 0
    assert np.abs(got - want).max() < TOL and multi.stats()["overflow"] == 0
    multi.close()


def test_three_lights_same_bits_whatever_the_kernel_and_the_schedule(rtc):
    """World.shadeHit adds the lights' contributions in light order (world.zig:89-96).  With three or more lights a sum
    split over the halves of a cooperative group would round differently from that running sum, and whether an iteration
    runs cooperatively depends on the schedule: the cooperative iteration (rtc_render_kernel_simple, small launches)
    therefore deals lights to halves only when there are at most two.  Here: three and five lights on an opaque world
    (every pixel's tree stays in one lane, so rtc.h promises bitwise reproducibility): the same bits from repeated
    renders, from the two-wave cooperative kernel and from the three-wave kernel, and the oracle's image."""
    import json
    for n_lights in (3, 5, 2):
        lights = [{"point-light": {"position": [-6 + 3 * i, 5 + i, -6 - i], "intensity": [0.5, 0.4 + 0.05 * i, 0.3]}} for i in range(n_lights)]
        objects = [{"type": {"plane": {}}, "material": {"pattern": {"type": {"checkers": [{"type": {"solid": [1, 1, 1]}}, {"type": {"solid": [0.2, 0.2, 0.3]}}]}}, "specular": 0.4}},
                   {"type": {"sphere": {}}, "transform": [{"translate": [0, 1, 0]}], "material": {"shininess": 30, "reflective": 0.3}},
                   {"type": {"cube": {}}, "transform": [{"scale": [0.5, 0.5, 0.5]}, {"translate": [2, 0.5, 0.5]}], "material": {"reflective": 0.5, "diffuse": 0.6}}]
        scene = {"camera": {"width": 72, "height": 40, "field-of-view": 0.9, "from": [0, 2.5, -7], "to": [0, 0.7, 0], "up": [0, 1, 0]},
                 "lights": lights, "objects": objects}
        hs = rtc.HostScene(json.dumps(scene))
        cam = hs.camera()
        want = ob.OracleScene(hs.desc).render(cam, 5)[0]
        images = []
        for simple3 in (None, 0):
            if simple3 is not None:
                rtc.set_option("simple3_min_chunks", simple3)
            try:
                gpu = rtc.GpuScene(hs.desc)
                for frame in range(3):                            # estimate-scheduled, measured, steady
                    images.append((gpu.render(cam, 5), gpu.last_kernel_name()))
            finally:
                rtc.set_option("simple3_min_chunks", -1)
        kernels = {k for _, k in images}
        assert kernels == set(_simple_names(hs)), kernels
        for img, k in images:
            assert np.abs(img - want).max() < TOL
            assert np.array_equal(img, images[0][0]), (n_lights, k)


def _simple_names(hs):
    """The two-wave and three-wave kernel of a simple world (top-level spheres, planes and cubes only): a world with more
    cubes than spheres runs the forms whose root loop rejects by world boxes (`_b`; rtc_capi.hip, box_cull)."""
    import ctypes
    d = hs.desc
    kinds = [d.leaf_kind[d.roots[i]] for i in range(d.n_roots) if not (d.roots[i] & 0x80000000)]
    b = "_b" if kinds.count(2) > kinds.count(0) else ""
    return "rtc_render_kernel_simple" + b, "rtc_render_kernel_simple3" + b


def _one_trial_only(names, three):
    """The frames a handle's trial ran on its three-wave kernel are ONE run: the frame that changes kernels packs its
    schedule into buffers that were sized for either kernel, so nothing is re-allocated, no frame is re-estimated and the
    trial does not start over (sized for the kernel in use only, the first trial of every handle was aborted
    and run again - a second run of three-wave frames a few frames after the first)."""
    idx = [i for i, n in enumerate(names) if n == three]
    assert idx and idx == list(range(idx[0], idx[-1] + 1)), names
    assert idx[0] == 4, names          # the estimate-scheduled frame, three timed two-wave frames, then the switch
    assert len(idx) >= 4, names        # the switch frame and three timed three-wave frames


def test_general_kernel_at_three_waves_and_its_trial(rtc):
    """rtc_render_kernel3 (the general kernel at three waves per SIMD: smaller LDS tables, two LDS entries of the walk's
    stack) forced by option: the oracle's image and counters on meshes, nested groups, cones on a group's always-visited
    list.  Then the measured choice: twelve frames of a static view large enough for the trial (frames alternate between
    the two kernels while a handle times them) - every frame is the same image to the last bits, whatever kernel ran."""
    torch = pytest.importorskip("torch")
    rtc.set_option("waves3", 1)
    try:
        for scene, w, h in (("teapot.json", 192, 108), ("dragons.json", 192, 108), ("groups.json", 150, 50), ("nefertiti.json", 90, 150)):
            got, want, stats, counters = _both(rtc, scene, w, h, 5)
            assert np.abs(got - want).max() < TOL, scene
            assert [stats["overflow"], stats["secondary"], stats["shadow_calls"]] == [0, counters["secondary"], counters["shadow"]]
        hs = rtc.HostScene.from_file("teapot.json")
        gpu = rtc.GpuScene(hs.desc)
        gpu.render(hs.camera(64, 64), 5)
        assert gpu.last_kernel_name() == "rtc_render_kernel3"
    finally:
        rtc.set_option("waves3", -1)
    hs = rtc.HostScene.from_file("teapot.json")
    cam = hs.camera(1024, 576)
    want = ob.OracleScene(hs.desc).render(cam, 5, row_step=24)[0]
    rows = np.arange(0, 576, 24)
    gpu = rtc.GpuScene(hs.desc)
    buf = torch.zeros((576, 1024, 3), dtype=torch.float64, device="cuda")
    kernels, first, names = set(), None, []
    for frame in range(14):
        gpu.render_device(cam, buf.data_ptr(), 5, None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        img = buf.cpu().numpy()
        kernels.add(gpu.last_kernel_name())
        names.append(gpu.last_kernel_name())
        assert np.abs(img[rows] - want[rows]).max() < TOL and gpu.stats()["overflow"] == 0, frame
        first = img if first is None else first
        assert np.abs(img - first).max() < REPEAT_TOL, frame
    assert kernels == {"rtc_render_kernel", "rtc_render_kernel3"}, kernels   # (the trial ran both)
    _one_trial_only(names, "rtc_render_kernel3")
    # a clone made in the middle of another handle's trial inherits that handle's DECISION, never the kernel under trial
    gpu2 = rtc.GpuScene(hs.desc)
    for frame in range(6):   # (estimate, three two-wave samples, the switch, one three-wave sample: mid-trial)
        gpu2.render_device(cam, buf.data_ptr(), 5, None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    assert gpu2.last_kernel_name() == "rtc_render_kernel3"
    clone = gpu2.clone()
    clone.render_device(cam, buf.data_ptr(), 5, None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert clone.last_kernel_name() == "rtc_render_kernel", clone.last_kernel_name()   # (nothing decided yet: two waves)
    gpu2.render_device(cam, buf.data_ptr(), 5, None, torch.cuda.current_stream().cuda_stream)   # (frames in flight now: no trial, the decision)
    torch.cuda.synchronize()
    assert gpu2.last_kernel_name() == "rtc_render_kernel", gpu2.last_kernel_name()
    assert np.abs(buf.cpu().numpy()[rows] - want[rows]).max() < TOL


def test_simple_world_trial_between_its_two_kernels(rtc):
    """A simple world's launch of one to four chunks per wave (here 960x540) is rendered by whichever of
    rtc_render_kernel_simple / _simple3 the handle's own trial found faster: fourteen frames of a static view run both,
    and every frame is the same image to the last bits - the oracle's - whichever kernel ran."""
    torch = pytest.importorskip("torch")
    hs = rtc.HostScene.from_file("cover.json")
    w, h = 960, 540
    cam = hs.camera(w, h)
    want = ob.OracleScene(hs.desc).render(cam, 5, row_step=27)[0]
    rows = np.arange(0, h, 27)
    gpu = rtc.GpuScene(hs.desc)
    buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    kernels, first, names = set(), None, []
    for frame in range(14):
        gpu.render_device(cam, buf.data_ptr(), 5, None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        img = buf.cpu().numpy()
        kernels.add(gpu.last_kernel_name())
        names.append(gpu.last_kernel_name())
        assert np.abs(img[rows] - want[rows]).max() < TOL and gpu.stats()["overflow"] == 0, frame
        first = img if first is None else first
        assert np.abs(img - first).max() < REPEAT_TOL, frame
    assert kernels == set(_simple_names(hs)), kernels   # (the trial ran both)
    _one_trial_only(names, _simple_names(hs)[1])


def test_host_output_in_bands(rtc):
    """rtc_render cuts a large frame into horizontal bands (the lower ones on clones of the handle), each copied to the
    caller while the next renders: forced to three and four bands on small images - a rectangle that starts off the chunk
    grid, a csg scene whose lists must grow in every band - the image, the summed counters and the RGBA8 clamp are those of
    the whole frame."""
    import json
    rtc.set_option("host_bands", 3)
    try:
        hs = rtc.HostScene.from_file("reflection_and_refraction.json")
        cam = hs.camera(320, 250)
        want, counters = ob.OracleScene(hs.desc).render(cam, 5)
        gpu = rtc.GpuScene(hs.desc)
        for frame in range(3):
            got = gpu.render(cam, 5)
            st = gpu.stats()
            assert np.abs(got - want).max() < TOL
            assert [st["primary"], st["secondary"], st["shadow_calls"], st["overflow"]] == [320 * 250, counters["secondary"], counters["shadow"], 0]
        x0, y0, w, h = 13, 21, 290, 215                      # (bands of a rectangle: whole chunk rows counted from ITS top)
        assert np.abs(gpu.render(cam, 5, (x0, y0, w, h)) - want[y0:y0 + h, x0:x0 + w]).max() < TOL
        assert gpu.stats()["primary"] == w * h
        buf_stats = gpu.stats()
        import torch
        dev = torch.zeros((250, 320, 3), dtype=torch.float64, device="cuda")
        gpu.render_device(cam, dev.data_ptr(), 5, None, torch.cuda.current_stream().cuda_stream)   # an ordinary launch after a banded one
        torch.cuda.synchronize()
        assert gpu.stats()["primary"] == 320 * 250 and gpu.stats() != buf_stats
        gpu.close()
        rtc.set_option("host_bands", 4)
        spheres = [{"type": {"sphere": {}}, "transform": [{"scale": [0.2 + 0.05 * i] * 3}]} for i in range(20)]
        scene = {"camera": {"width": 64, "height": 288, "field-of-view": 1.4, "from": [0, 0, -6], "to": [0, 0, 0], "up": [0, 1, 0]},
                 "lights": [{"point-light": {"position": [-5, 5, -5], "intensity": [1, 1, 1]}}],
                 "objects": [{"type": {"csg": {"operation": "union", "left": {"type": {"group": spheres}},
                                               "right": {"type": {"cube": {}}}}}}]}
        hs = rtc.HostScene(json.dumps(scene))
        cam = hs.camera()
        want, counters = ob.OracleScene(hs.desc).render(cam, 5)
        gpu = rtc.GpuScene(hs.desc)
        got = gpu.render(cam, 5)                             # (the middle bands need 42 list entries: every handle grows)
        st = gpu.stats()
        assert np.abs(got - want).max() < TOL
        assert [st["overflow"], st["secondary"], st["shadow_calls"]] == [0, counters["secondary"], counters["shadow"]]
        gpu.close()
    finally:
        rtc.set_option("host_bands", 0)


def test_csg_of_forty_nested_operations(rtc):
    """A csg whose left operand is a csg ... forty deep (union / difference / intersection in turn, spheres and cubes
    marching along x): one unit of forty csg nodes - the (inl, inr) toggles of every node are one bit each of a 64-bit
    mask - and lists of up to 80 entries; sixty-five nodes are refused."""
    import json

    def chain(n):
        node = {"type": {"sphere": {}}, "transform": [{"scale": [0.5, 0.5, 0.5]}]}
        for i in range(n):
            leaf = {"type": {"cube" if i % 2 else "sphere": {}},
                    "transform": [{"scale": [0.35, 0.35 + 0.01 * i, 0.35]}, {"translate": [0.12 * (i + 1), 0.02 * (i % 5), 0]}]}
            node = {"type": {"csg": {"operation": ["union", "difference", "intersection", "union"][i % 4] if i % 7 else "union",
                                     "left": node, "right": leaf}}}
        return node
    scene = {"camera": {"width": 64, "height": 40, "field-of-view": 0.9, "from": [2.5, 1.5, -7], "to": [2.5, 0, 0], "up": [0, 1, 0]},
             "lights": [{"point-light": {"position": [-5, 6, -8], "intensity": [1, 1, 1]}}],
             "objects": [chain(40), {"type": {"plane": {}}, "transform": [{"translate": [0, -1, 0]}]}]}
    hs = rtc.HostScene(json.dumps(scene))
    cam = hs.camera()
    gpu = rtc.GpuScene(hs.desc)
    got = gpu.render(cam, 5)
    want, counters = ob.OracleScene(hs.desc).render(cam, 5)
    st = gpu.stats()
    assert np.abs(got - want).max() < TOL
    assert [st["overflow"], st["shadow_calls"]] == [0, counters["shadow"]]
    scene["objects"] = [chain(65)]
    with pytest.raises(rtc.RtcError) as e:
        rtc.GpuScene(rtc.HostScene(json.dumps(scene)).desc)
    assert e.value.name in ("Unsupported", "StackOverflow")   # (the tree's depth is checked first: refused either way)


def test_mixing_patterns_nested_in_one_another(rtc):
    """gradient.zig:19-33 and blend.zig:16-27 take arbitrary patterns as children: a blend of gradients, a gradient whose
    ends are a blend and a radial gradient over perturbed stripes, three levels deep - walked on the device with an
    explicit stack (pattern_tree, the *_ext kernels); nine levels are refused, loudly."""
    import json
    solid = lambda r, g, b: {"type": {"solid": [r, g, b]}}
    grad = lambda a, b, t=None: {"type": {"gradient": [a, b]}, **({"transform": t} if t else {})}
    blend = lambda a, b: {"type": {"blend": [a, b]}}
    radial = lambda a, b: {"type": {"radial-gradient": [a, b]}, "transform": [{"scale": [0.4, 0.4, 0.4]}]}
    stripes = {"type": {"stripes": [solid(0.9, 0.1, 0.1), solid(0.1, 0.1, 0.9)]}, "transform": [{"scale": [0.2, 0.2, 0.2]}, {"rotate-y": 0.7}]}
    perturbed = {"type": {"perturb": stripes}}
    floor = blend(grad(solid(1, 0, 0), solid(0, 1, 0), [{"rotate-y": 1.2}]), grad(solid(0, 0, 1), stripes))
    ball = grad(blend(solid(1, 1, 0), radial(solid(0, 1, 1), perturbed)), radial(grad(solid(1, 0, 1), solid(0.2, 0.2, 0.2)), solid(1, 1, 1)),
                [{"scale": [0.5, 0.5, 0.5]}])
    scene = {"camera": {"width": 96, "height": 64, "field-of-view": 1.0, "from": [0, 2.2, -5], "to": [0, 0.6, 0], "up": [0, 1, 0]},
             "lights": [{"point-light": {"position": [-4, 6, -6], "intensity": [1, 1, 1]}}],
             "objects": [{"type": {"plane": {}}, "material": {"pattern": floor, "specular": 0, "reflective": 0.2}},
                         {"type": {"sphere": {}}, "transform": [{"translate": [0, 1, 0]}], "material": {"pattern": ball, "diffuse": 0.8}},
                         {"type": {"cube": {}}, "transform": [{"scale": [0.5, 0.5, 0.5]}, {"translate": [2, 0.5, 0.5]}],
                          "material": {"pattern": blend(floor, ball), "transparency": 0.3, "refractive-index": 1.2}}]}
    hs = rtc.HostScene(json.dumps(scene))
    cam = hs.camera()
    gpu = rtc.GpuScene(hs.desc)
    got = gpu.render(cam, 5)
    assert gpu.last_kernel_name().endswith("_ext")
    want, counters = ob.OracleScene(hs.desc).render(cam, 5)
    st = gpu.stats()
    assert np.abs(got - want).max() < TOL
    assert [st["overflow"], st["secondary"], st["shadow_calls"]] == [0, counters["secondary"], counters["shadow"]]
    assert len(np.unique(np.round(want.reshape(-1, 3), 3), axis=0)) > 200        # (the patterns do show)
    deep = solid(1, 1, 1)
    for level in range(9):
        deep = blend(deep, solid(0, 0, 0)) if level % 2 else grad(deep, solid(0, 0, 0))
    scene["objects"] = [{"type": {"plane": {}}, "material": {"pattern": deep}}]
    with pytest.raises(rtc.RtcError) as e:
        rtc.GpuScene(rtc.HostScene(json.dumps(scene)).desc)
    assert e.value.name == "Unsupported"


def test_interactive_camera_loop(rtc):
    """The "preheated" mode of lib.zig:135-190: ONE scene handle on the GPU, the camera orbits and dollies between
    frames (rotateCamera / moveCamera); every frame matches the oracle and the RGBA8 framebuffer is its clamp."""
    import oracle_binding as ob
    hs = rtc.HostScene.from_file("reflection_and_refraction.json")
    gpu = rtc.GpuScene(hs.desc)
    osc = ob.OracleScene(hs.desc)
    frames = []
    for step in range(4):
        cam = hs.camera(96, 54)
        got = gpu.render(cam, 5)
        want, _ = osc.render(cam, 5)
        assert np.abs(got - want).max() < TOL, step
        frames.append(got)
        rgba = rtc.canvas_rgba8(got)
        assert rgba.shape == (54, 96, 4) and (rgba[..., 3] == 255).all()
        assert np.array_equal(gpu.render_rgba8(cam, 5), rgba)     # the same clamp on the device (rtc_render_rgba8)
        assert np.array_equal(gpu.render_rgba8(cam, 5, (10, 5, 33, 20)), rgba[5:25, 10:43])
        assert np.array_equal(rgba[..., :3], np.clip(np.floor(got * 255.0 + 0.5), 0, 255).astype(np.uint8))   # color.zig:61-71 (@round: half away from 0)
        if step % 2 == 0:
            hs.rotate_camera(0.35)
        else:
            hs.move_camera(0.15)
    assert np.abs(frames[0] - frames[1]).max() > 1e-3 and np.abs(frames[1] - frames[2]).max() > 1e-3


def test_moving_view_measured_every_nth_frame(rtc):
    """Option "measure_every": a view that moves in small steps is measured (and its schedule re-packed) every n-th frame
    only, a jump at once; the frames in between run on a schedule up to n - 1 frames old.  Results never depend on the
    schedule: every frame of an orbit equals the frame a fresh handle renders of that view (to the last bits: the pixels
    of cover whose ray trees are shared between lanes are summed with atomics), whatever n."""
    hs = rtc.HostScene.from_file("cover.json")
    W, H = 320, 200                                   # 1000 chunks: scheduled launches
    cams = []
    for step in range(9):
        cams.append(hs.camera(W, H))
        hs.rotate_camera(1.0 if step == 5 else 0.01)  # small steps, one jump
    want = [rtc.GpuScene(hs.desc).render(c, 5) for c in cams]
    try:
        for every in (4, 1):
            rtc.set_option("measure_every", every)
            gpu = rtc.GpuScene(hs.desc)
            for i, c in enumerate(cams):
                got = gpu.render(c, 5)
                assert np.abs(got - want[i]).max() < REPEAT_TOL, (every, i)
                assert gpu.stats()["overflow"] == 0
    finally:
        rtc.set_option("measure_every", 1)


def test_bench_multi_rank_path_rehearsal():
    """bench.py's N > 1 code path (tile split, gather, rtc_assemble_tiles_device, stats reduction, the JSON line) with
    two ranks on the ONE GPU of this box over gloo (--rehearse); real RCCL runs over N GPUs are the driver's."""
    import json
    import subprocess
    import sys
    repo = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", _os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--width", "480", "--height", "270", "--rehearse", "--check"]
    out = subprocess.run(cmd, cwd=repo, capture_output=True, text=True, timeout=200)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "check ok" in out.stderr
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout            # ONE JSON line on stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0 and "roofline" in line
    assert line["config"]["frames_in_flight"] == 3   # (a split frame's default: three handles, streams and tile buffers per rank)
    assert line["config"]["rays_per_frame"]["primary"] == 480 * 270
    ranks = line["ranks"]                         # what a scaling curve is read from: every rank's share
    assert len(ranks["render_ms"]) == 2 and min(ranks["render_ms"]) > 0 and sum(ranks["tiles"]) == 8 * 5
    assert ranks["render_ms_max"] >= ranks["render_ms_mean"] and ranks["gather_unpermute_ms_rank0"] > 0
    # the same with the shares clamped to RGBA8 by the rank that rendered them (4 B/pixel through the gather)
    out = subprocess.run(cmd[:-1] + ["--output", "rgba8", "--check"], cwd=repo, capture_output=True, text=True, timeout=200)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "check ok: assembled RGBA8 framebuffer" in out.stderr
    assert json.loads([l for l in out.stdout.splitlines() if l.strip()][0])["config"]["output"].startswith("RGBA8")


def test_bench_frames_in_flight():
    """bench.py --inflight 2 on one GPU: two handles, streams and canvases, frames dealt round-robin; the last frame is
    checked against a plain render."""
    import json
    import subprocess
    import sys
    repo = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    cmd = [sys.executable, _os.path.join(repo, "bench.py"), "--steps", "5", "--warmup", "1", "--settle-frames", "4", "--width", "320",
           "--height", "180", "--inflight", "2", "--no-cpu-baseline", "--no-extras", "--check"]
    out = subprocess.run(cmd, cwd=repo, capture_output=True, text=True, timeout=200)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "check ok" in out.stderr
    line = json.loads([l for l in out.stdout.splitlines() if l.strip()][0])
    assert line["config"]["frames_in_flight"] == 2 and line["n_gpus"] == 1 and line["value"] > 0


def _json_scene(objects, lights, w=24, h=16):
    import json
    return json.dumps({"camera": {"width": w, "height": h, "field-of-view": 1.0, "from": [0, 1.5, -5], "to": [0, 1, 0], "up": [0, 1, 0]},
                       "lights": lights, "objects": objects})


def _parity(rtc, scene_json, depth=5, cam_size=None):
    import oracle_binding as ob
    hs = rtc.HostScene(scene_json)
    cam = hs.camera(*cam_size) if cam_size else hs.camera()
    gpu = rtc.GpuScene(hs.desc)
    got = gpu.render(cam, depth)
    want, counters = ob.OracleScene(hs.desc).render(cam, depth)
    assert got.shape == want.shape and np.isfinite(got).all()
    assert np.abs(got - want).max() < TOL
    st = gpu.stats()
    assert [st["primary"], st["secondary"], st["shadow_calls"]] == [counters["primary"], counters["secondary"], counters["shadow"]]
    return got


LIGHT = {"point-light": {"position": [-5, 8, -6], "intensity": [1, 1, 1]}}
BALL = {"type": {"sphere": {}}, "transform": [{"translate": [0, 1, 0]}],
        "material": {"reflective": 0.5, "transparency": 0.5, "refractive-index": 1.4}}
FLOOR = {"type": {"plane": {}}, "material": {"reflective": 0.3}}


def test_edge_empty_world_and_no_lights(rtc):
    """World.objects empty -> every pixel black (world.zig:119); no lights -> shadeHit sums nothing (world.zig:89)."""
    assert np.all(_parity(rtc, _json_scene([], [LIGHT])) == 0.0)
    assert np.all(_parity(rtc, _json_scene([], [])) == 0.0)
    _parity(rtc, _json_scene([FLOOR, BALL], []))     # reflections / refractions of nothing but black: still traced


@pytest.mark.parametrize("size", [(1, 1), (1, 7), (9, 1), (8, 8), (9, 9), (63, 65), (130, 3)])
def test_edge_ragged_image_sizes(rtc, size):
    """Canvas sizes around the 8x8 chunk and 64-lane wave boundaries, incl. a single pixel and single rows / columns."""
    _parity(rtc, _json_scene([FLOOR, BALL], [LIGHT]), cam_size=size)


def test_edge_depth_limits(rtc):
    """remaining = 0 spawns nothing (world.zig:158,172); 16 is the deepest the per-lane stacks allow; 17 is refused."""
    scene = _json_scene([FLOOR, BALL, {"type": {"sphere": {}}, "transform": [{"translate": [1.8, 1, 0.5]}],
                                       "material": {"reflective": 0.9, "transparency": 0.9, "refractive-index": 1.1}}], [LIGHT])
    _parity(rtc, scene, depth=0)
    _parity(rtc, scene, depth=1)
    _parity(rtc, scene, depth=16, cam_size=(12, 8))
    hs = rtc.HostScene(scene)
    with pytest.raises(rtc.RtcError) as e:
        rtc.GpuScene(hs.desc).render(hs.camera(), 17)
    assert e.value.name == "InvalidArgument"


def test_edge_many_lights_and_materials(rtc):
    """More lights / materials / patterns than the LDS staging area holds: the table-in-memory kernel variant."""
    lights = [{"point-light": {"position": [6 * np.cos(k), 6 + (k % 3), 6 * np.sin(k) - 2], "intensity": [0.06, 0.05, 0.07]}}
              for k in range(20)]
    objs = [FLOOR]
    for k in range(70):
        objs.append({"type": {"sphere": {}}, "transform": [{"scale": [0.2, 0.2, 0.2]}, {"translate": [(k % 10) * 0.5 - 2.2, 0.2 + (k // 10) * 0.45, 0]}],
                     "material": {"pattern": {"type": {"stripes": [{"type": {"solid": [k / 70, 0.3, 0.6]}}, {"type": {"solid": [0.9, k / 70, 0.1]}}]},
                                              "transform": [{"scale": [0.1 + 0.01 * k] * 3}]},
                                  "shininess": 10 + k, "reflective": 0.1 if k % 7 == 0 else 0.0}})
    _parity(rtc, _json_scene(objs, lights, 40, 24))


def test_shared_divisor_quotients_are_exact(tmp_path):
    """cube_slab() divides a cube axis's two numerators by the same direction component with ONE refined reciprocal
    (csrc/rtc_kernels.hip: refined_rcp / quotient).  tests/hip/shared_divisor_check.hip runs that sequence beside the
    plain division on 2^28 operand pairs (random over the whole range cube_slab admits, and edge cases around the
    1e-5 threshold and numerators of a few ulps): the quotients must be bit-identical."""
    import subprocess
    repo = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    exe = str(tmp_path / "shared_divisor_check")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-o", exe,
                    _os.path.join(repo, "tests", "hip", "shared_divisor_check.hip")], check=True, capture_output=True, timeout=600)
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and " 0 mismatches" in run.stdout, run.stdout + run.stderr
