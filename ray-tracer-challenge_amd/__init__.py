"""ray-tracer-challenge_amd — MI355X-native render path for SinclaM/ray-tracer-challenge.

Python is only the harness language here (tests, bench, multi-GPU driver): this
module is a ctypes binding of the two product libraries

  lib/librtc_hip.so   HIP kernels behind the C ABI of include/rtc.h
                      (replaces Camera.render, reference src/raytracer/camera.zig:80-125)
  lib/librtc_host.so  C++ host side: scene JSON / OBJ loaders, Camera/World/Canvas
                      mirror (reference src/parsing/*.zig, src/raytracer/canvas.zig)

There is no CPU implementation of the render path in this package: `GpuScene`
needs the HIP library and a GPU, and raises `RtcError` otherwise.  The package
directory name contains a hyphen, so import it with
`importlib.import_module("ray-tracer-challenge_amd")`.
"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (RTC_LIB_DIR: a diagnostic build of the three libraries somewhere else - tools/variants.py; the product is lib/)
LIB_DIR = os.environ.get("RTC_LIB_DIR", os.path.join(_HERE, "lib"))
REPO_ROOT = os.path.dirname(_HERE)
# Where HostScene.from_file looks for a scene given by name, and for the OBJ / PNG files a scene names.  The harness
# (tests, bench.py) uses the copies of the reference's scene and data files kept as fixtures under tests/golden/ (the
# reference tree does not exist on the GPU box); another host points these at its own directories.
SCENE_DIR = os.environ.get("RTC_SCENE_DIR", os.path.join(REPO_ROOT, "tests", "golden", "scenes"))
DATA_DIR = os.environ.get("RTC_DATA_DIR", os.path.join(REPO_ROOT, "tests", "golden", "data"))

RTC_CHILD_NODE_BIT = 0x80000000
RTC_MAT_STRIDE = 7
REFERENCE_DEPTH = 5  # camera.zig:118

_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)


class RtcError(RuntimeError):
    """An rtc_status != RTC_OK; `.name` is the Zig-style error name."""

    def __init__(self, name, message):
        super().__init__(message or name)
        self.name = name


class SceneDesc(C.Structure):
    """struct rtc_scene_desc (include/rtc.h)."""

    _fields_ = [
        ("abi_version", C.c_uint32),
        ("n_xforms", C.c_uint32), ("xf_inv", _dp), ("xf_inv_t", _dp),
        ("n_leaves", C.c_uint32), ("leaf_kind", _u8p), ("leaf_xform", _u32p), ("leaf_material", _u32p),
        ("leaf_shadow", _u8p), ("leaf_id", _u32p), ("leaf_geom", _u32p),
        ("n_cyls", C.c_uint32), ("cyl_min", _dp), ("cyl_max", _dp), ("cyl_closed", _u8p),
        ("n_tris", C.c_uint32), ("tri_p1", _dp), ("tri_e1", _dp), ("tri_e2", _dp),
        ("tri_n1", _dp), ("tri_n2", _dp), ("tri_n3", _dp),
        ("n_materials", C.c_uint32), ("mat_params", _dp), ("mat_pattern", _u32p),
        ("n_patterns", C.c_uint32), ("pat_kind", _u8p), ("pat_inv", _dp), ("pat_rgb", _dp),
        ("pat_a", _u32p), ("pat_b", _u32p),
        ("n_nodes", C.c_uint32), ("node_min", _dp), ("node_max", _dp), ("node_first", _u32p), ("node_count", _u32p),
        ("node_op", _u8p),
        ("n_children", C.c_uint32), ("children", _u32p),
        ("n_roots", C.c_uint32), ("roots", _u32p),
        ("n_lights", C.c_uint32), ("light_pos", _dp), ("light_rgb", _dp),
        ("n_texmaps", C.c_uint32), ("tex_mapping", _u8p), ("tex_uv", _u32p),
        ("n_uvs", C.c_uint32), ("uv_kind", _u8p), ("uv_size", _dp), ("uv_sub", _u32p), ("uv_image", _u32p),
        ("uv_interp", _u8p),
        ("n_images", C.c_uint32), ("img_width", _u32p), ("img_height", _u32p), ("img_offset", C.POINTER(C.c_uint64)),
        ("img_rgb", C.POINTER(C.c_float)),
    ]


class Camera(C.Structure):
    """struct rtc_camera (include/rtc.h)."""

    _fields_ = [("hsize", C.c_uint32), ("vsize", C.c_uint32),
                ("half_width", C.c_double), ("half_height", C.c_double), ("pixel_size", C.c_double),
                ("inv_view", C.c_double * 16)]


class Stats(C.Structure):
    """struct rtc_stats (include/rtc.h)."""

    _fields_ = [("primary", C.c_uint64), ("secondary", C.c_uint64), ("shadow_calls", C.c_uint64),
                ("shadow_traced", C.c_uint64), ("overflow", C.c_uint64)]


# (include/rtc.h: what a host binds ...)
RTC_SYMBOLS = ["rtc_scene_create", "rtc_scene_clone", "rtc_scene_destroy", "rtc_render", "rtc_render_rgba8", "rtc_render_device",
               "rtc_render_tiles_device", "rtc_assemble_tiles_device", "rtc_render_tile_list_device", "rtc_get_tile_costs",
               "rtc_assign_tiles", "rtc_assemble_tile_list_device", "rtc_assemble_tile_list_rgba8_device", "rtc_scatter_tile_list_device",
               "rtc_scatter_tile_list_rgba8_device", "rtc_scene_synchronize", "rtc_get_stats", "rtc_last_error", "rtc_status_name",
               "rtc_grow_csg_lists", "rtc_canvas_register", "rtc_canvas_unregister", "rtc_rgba8_device"]
# (... and include/rtc_diag.h: diagnostics and tuning, for the tests, bench.py and tools/)
RTC_DIAG_SYMBOLS = ["rtc_set_option", "rtc_last_kernel_name", "rtc_get_schedule", "rtc_get_chunk_times", "rtc_diag_build_tables", "rtc_diag_root_boxes"]
HOST_SYMBOLS = ["rtch_last_error", "rtch_scene_load", "rtch_scene_free", "rtch_scene_desc", "rtch_scene_camera",
                "rtch_camera_rotate", "rtch_camera_move", "rtch_camera_make", "rtch_canvas_ppm", "rtch_canvas_rgba8", "rtch_scene_render", "rtch_set_loader_threads"]

MULTI_SYMBOLS = ["rtc_multi_create", "rtc_multi_destroy", "rtc_multi_render", "rtc_multi_render_rgba8", "rtc_multi_render_device", "rtc_multi_render_rgba8_device",
                 "rtc_multi_synchronize", "rtc_multi_stream", "rtc_multi_get_stats", "rtc_multi_balance", "rtc_multi_last_error"]
RTC_MULTI_VIRTUAL = 1

_hip = None
_host = None
_multi = None


def _one_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64 (same SONAME as ROCm's, other file).
    A process that loads librtc_hip.so first (bound to /opt/rocm's copy) and torch later ends up with TWO
    HIP/HSA runtimes driving one GPU, and stream handles passed between them belong to the wrong one.  When
    torch is installed its copy is mapped first, so that the loader resolves our DT_NEEDED to it and one
    runtime serves both (what happens anyway when torch is imported before this package)."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    for name in ("libamdhip64.so",):
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", name)
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def hip_lib():
    """librtc_hip.so; raises if the library has not been built (no fallback)."""
    global _hip
    if _hip is None:
        path = os.path.join(LIB_DIR, "librtc_hip.so")
        if not os.path.exists(path):
            raise RtcError("LibraryMissing", f"{path} not built: run `make` or __graft_entry__.build()")
        _one_hip_runtime()
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        lib.rtc_last_error.restype = C.c_char_p
        lib.rtc_status_name.restype = C.c_char_p
        lib.rtc_status_name.argtypes = [C.c_int]
        lib.rtc_scene_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
        lib.rtc_scene_clone.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        lib.rtc_scene_destroy.argtypes = [C.c_void_p]
        lib.rtc_scene_destroy.restype = None
        lib.rtc_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32] + [C.c_uint32] * 4 + [C.c_void_p]
        lib.rtc_render_rgba8.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32] + [C.c_uint32] * 4 + [C.c_void_p]
        lib.rtc_render_device.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32] + [C.c_uint32] * 4 + [C.c_void_p, C.c_void_p]
        lib.rtc_render_tiles_device.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32] + [C.c_uint32] * 5 + [C.c_void_p, C.c_void_p]
        lib.rtc_assemble_tiles_device.argtypes = [C.c_void_p] + [C.c_uint32] * 6 + [C.c_void_p, C.c_void_p]
        lib.rtc_render_tile_list_device.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, _u32p, C.c_uint32,
                                                    C.c_void_p, C.c_void_p]
        lib.rtc_get_tile_costs.argtypes = [C.c_void_p, _dp, C.c_uint32]
        lib.rtc_assign_tiles.argtypes = [_dp, C.c_uint32, C.c_uint32, _u32p, _u32p]
        lib.rtc_assemble_tile_list_device.argtypes = [C.c_void_p, C.c_void_p] + [C.c_uint32] * 4 + [C.c_void_p, C.c_void_p]
        lib.rtc_assemble_tile_list_rgba8_device.argtypes = [C.c_void_p, C.c_void_p] + [C.c_uint32] * 4 + [C.c_void_p, C.c_void_p]
        lib.rtc_scene_synchronize.argtypes = [C.c_void_p]
        lib.rtc_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        lib.rtc_grow_csg_lists.argtypes = [C.c_void_p]
        lib.rtc_last_kernel_name.argtypes = [C.c_void_p]
        lib.rtc_last_kernel_name.restype = C.c_char_p
        lib.rtc_get_schedule.argtypes = [C.c_void_p, _u32p, C.c_size_t, _u32p]
        lib.rtc_get_chunk_times.argtypes = [C.c_void_p, C.POINTER(Camera), _u32p, _u32p, C.c_size_t, _u32p]
        lib.rtc_canvas_register.argtypes = [C.c_void_p, C.c_size_t]
        lib.rtc_canvas_unregister.argtypes = [C.c_void_p]
        lib.rtc_rgba8_device.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        lib.rtc_set_option.argtypes = [C.c_char_p, C.c_double]
        lib.rtc_diag_build_tables.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        _hip = lib
    return _hip


def host_lib():
    """librtc_host.so (links librtc_hip.so)."""
    global _host
    if _host is None:
        hip_lib()
        path = os.path.join(LIB_DIR, "librtc_host.so")
        if not os.path.exists(path):
            raise RtcError("LibraryMissing", f"{path} not built: run `make` or __graft_entry__.build()")
        lib = C.CDLL(path)
        lib.rtch_last_error.restype = C.c_char_p
        lib.rtch_scene_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        lib.rtch_scene_free.argtypes = [C.c_void_p]
        lib.rtch_scene_free.restype = None
        lib.rtch_set_loader_threads.argtypes = [C.c_uint32]
        lib.rtch_set_loader_threads.restype = None
        lib.rtch_scene_desc.argtypes = [C.c_void_p]
        lib.rtch_scene_desc.restype = C.POINTER(SceneDesc)
        lib.rtch_scene_camera.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(Camera)]
        lib.rtch_camera_rotate.argtypes = [C.c_void_p, C.c_double]
        lib.rtch_camera_move.argtypes = [C.c_void_p, C.c_double]
        lib.rtch_camera_make.argtypes = [C.c_uint32, C.c_uint32, C.c_double, _dp, _dp, _dp, C.POINTER(Camera)]
        lib.rtch_canvas_ppm.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]
        lib.rtch_canvas_ppm.restype = C.c_size_t
        lib.rtch_canvas_rgba8.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.rtch_canvas_rgba8.restype = None
        lib.rtch_scene_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        _host = lib
    return _host


def multi_lib():
    """librtc_multi.so (links librtc_hip.so and RCCL): the single-process multi-GPU render of include/rtc_multi.h."""
    global _multi
    if _multi is None:
        hip_lib()
        path = os.path.join(LIB_DIR, "librtc_multi.so")
        if not os.path.exists(path):
            raise RtcError("LibraryMissing", f"{path} not built: run `make` or __graft_entry__.build()")
        lib = C.CDLL(path)
        lib.rtc_multi_last_error.restype = C.c_char_p
        lib.rtc_multi_create.argtypes = [C.POINTER(SceneDesc), C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        lib.rtc_multi_destroy.argtypes = [C.c_void_p]
        lib.rtc_multi_destroy.restype = None
        lib.rtc_multi_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_void_p]
        lib.rtc_multi_render_rgba8.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_void_p]
        lib.rtc_multi_render_device.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.POINTER(C.c_void_p)]
        lib.rtc_multi_render_rgba8_device.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.POINTER(C.c_void_p)]
        lib.rtc_multi_synchronize.argtypes = [C.c_void_p]
        lib.rtc_multi_stream.argtypes = [C.c_void_p]
        lib.rtc_multi_stream.restype = C.c_void_p
        lib.rtc_multi_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        lib.rtc_multi_balance.argtypes = [C.c_void_p, _u32p, _dp]
        _multi = lib
    return _multi


class MultiGpu:
    """rtc_multi: one process, n GPUs, one gather per frame (include/rtc_multi.h)."""

    def __init__(self, desc, n_gpus, virtual=False, frames=1):
        """frames: frame slots (RTC_MULTI_FRAMES): render_device hands the frames to them in turn."""
        self.n = n_gpus
        self._m = C.c_void_p()
        flags = (RTC_MULTI_VIRTUAL if virtual else 0) | ((frames & 15) << 8)
        self._check(multi_lib().rtc_multi_create(C.byref(desc), n_gpus, flags, C.byref(self._m)))

    @staticmethod
    def _check(status):
        if status != 0:
            raise RtcError(hip_lib().rtc_status_name(status).decode(), multi_lib().rtc_multi_last_error().decode())

    def render(self, cam, max_depth=REFERENCE_DEPTH, out=None):
        if out is None:
            out = np.empty((cam.vsize, cam.hsize, 3), dtype=np.float64)
        self._check(multi_lib().rtc_multi_render(self._m, C.byref(cam), max_depth, out.ctypes.data))
        return out

    def render_rgba8(self, cam, max_depth=REFERENCE_DEPTH, out=None):
        """The RGBA8 framebuffer of lib.zig:146-153, clamped on GPU 0; [vsize][hsize][4] u8 (host)."""
        if out is None:
            out = np.empty((cam.vsize, cam.hsize, 4), dtype=np.uint8)
        self._check(multi_lib().rtc_multi_render_rgba8(self._m, C.byref(cam), max_depth, out.ctypes.data))
        return out

    def render_device(self, cam, max_depth=REFERENCE_DEPTH):
        """Enqueues the frame and returns the device pointer (GPU 0) of its [vsize][hsize][3] f64 canvas; nothing is
        copied or waited for (synchronize(), or work enqueued on stream(), orders behind it)."""
        ptr = C.c_void_p()
        self._check(multi_lib().rtc_multi_render_device(self._m, C.byref(cam), max_depth, C.byref(ptr)))
        return ptr.value

    def render_rgba8_device(self, cam, max_depth=REFERENCE_DEPTH):
        """render_device for the RGBA8 framebuffer: the ranks clamp their tiles, 4 bytes per pixel are gathered."""
        ptr = C.c_void_p()
        self._check(multi_lib().rtc_multi_render_rgba8_device(self._m, C.byref(cam), max_depth, C.byref(ptr)))
        return ptr.value

    def synchronize(self):
        self._check(multi_lib().rtc_multi_synchronize(self._m))

    def stream(self):
        return multi_lib().rtc_multi_stream(self._m)

    def stats(self):
        st = Stats()
        self._check(multi_lib().rtc_multi_get_stats(self._m, C.byref(st)))
        return {k: getattr(st, k) for k, _ in Stats._fields_}

    def balance(self):
        tiles = np.zeros(self.n, dtype=np.uint32)
        ratio = C.c_double()
        self._check(multi_lib().rtc_multi_balance(self._m, tiles.ctypes.data_as(_u32p), C.byref(ratio)))
        return tiles, ratio.value

    def close(self):
        if self._m:
            multi_lib().rtc_multi_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _check_host(status):
    if status != 0:
        msg = host_lib().rtch_last_error().decode()
        raise RtcError(msg.split(":")[0], msg)


def _check_hip(status):
    if status != 0:
        lib = hip_lib()
        raise RtcError(lib.rtc_status_name(status).decode(), lib.rtc_last_error().decode())


class HostScene:
    """parseScene (reference src/parsing/scene.zig:612-661) + flattening, on the host."""

    def __init__(self, scene_json, data_dir=DATA_DIR):
        if isinstance(scene_json, str):
            scene_json = scene_json.encode()
        self._h = C.c_void_p()
        _check_host(host_lib().rtch_scene_load(scene_json, data_dir.encode(), C.byref(self._h)))
        self.desc = host_lib().rtch_scene_desc(self._h).contents
        self.desc._owner = self   # the tables behind `desc` are this object's: HostScene(...).desc alone must keep them alive

    @classmethod
    def from_file(cls, name, data_dir=DATA_DIR):
        path = name if os.path.exists(name) else os.path.join(SCENE_DIR, name)
        with open(path, "rb") as f:
            return cls(f.read(), data_dir)

    def camera(self, width=0, height=0):
        cam = Camera()
        _check_host(host_lib().rtch_scene_camera(self._h, width, height, C.byref(cam)))
        return cam

    def rotate_camera(self, angle):
        """Renderer.rotateCamera (lib.zig:166-178): orbit the camera around its target, about `up`."""
        _check_host(host_lib().rtch_camera_rotate(self._h, C.c_double(angle)))

    def move_camera(self, distance):
        """Renderer.moveCamera (lib.zig:180-190): move the camera along its line of sight by distance * |to - from|."""
        _check_host(host_lib().rtch_camera_move(self._h, C.c_double(distance)))

    def array(self, field, count, width=1, dtype=None):
        """numpy view of one table of the flat description (writable: tests patch tables)."""
        ptr = getattr(self.desc, field)
        n = count * width
        if n == 0:
            return np.zeros((0,), dtype=dtype)
        arr = np.ctypeslib.as_array(ptr, shape=(n,))
        return arr.reshape(count, width) if width > 1 else arr

    def close(self):
        if self._h:
            host_lib().rtch_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_camera(hsize, vsize, fov, frm, to, up):
    """Camera.new + setTransform(viewTransform(from,to,up)) (camera.zig:33-61, matrix.zig:54-67)."""
    cam = Camera()
    arr = lambda v: (C.c_double * 3)(*v)
    _check_host(host_lib().rtch_camera_make(hsize, vsize, fov, arr(frm), arr(to), arr(up), C.byref(cam)))
    return cam


class GpuScene:
    """rtc_scene: the flat scene resident in HBM; many renders per upload."""

    def __init__(self, desc, _clone_of=None):
        self._s = C.c_void_p()
        if _clone_of is not None:
            _check_hip(hip_lib().rtc_scene_clone(_clone_of._s, C.byref(self._s)))
        else:
            _check_hip(hip_lib().rtc_scene_create(C.byref(desc), C.byref(self._s)))

    def clone(self):
        """rtc_scene_clone: a handle of its own (stream, schedule, counters) on the same device copy of the scene - one
        per frame in flight."""
        return GpuScene(None, _clone_of=self)

    def render(self, cam, max_depth=REFERENCE_DEPTH, tile=None):
        """Camera.render for the whole image or tile=(x0,y0,w,h); returns [h][w][3] f64 (host)."""
        x0, y0, w, h = tile if tile else (0, 0, cam.hsize, cam.vsize)
        out = np.empty((h, w, 3), dtype=np.float64)
        _check_hip(hip_lib().rtc_render(self._s, C.byref(cam), max_depth, x0, y0, w, h, out.ctypes.data))
        return out

    def render_into(self, cam, out, max_depth=REFERENCE_DEPTH):
        """Camera.render into a caller-owned [h][w][3] f64 array (an interactive host reuses its canvas, and pins it
        once with canvas_register so that the copy runs at link speed)."""
        assert out.dtype == np.float64 and out.flags["C_CONTIGUOUS"] and out.shape == (cam.vsize, cam.hsize, 3)
        _check_hip(hip_lib().rtc_render(self._s, C.byref(cam), max_depth, 0, 0, cam.hsize, cam.vsize, out.ctypes.data))
        return out

    def render_rgba8(self, cam, max_depth=REFERENCE_DEPTH, tile=None, out=None):
        """The RGBA8 framebuffer of lib.zig:146-153, clamped on the device; returns [h][w][4] u8 (host)."""
        x0, y0, w, h = tile if tile else (0, 0, cam.hsize, cam.vsize)
        if out is None:
            out = np.empty((h, w, 4), dtype=np.uint8)
        _check_hip(hip_lib().rtc_render_rgba8(self._s, C.byref(cam), max_depth, x0, y0, w, h, out.ctypes.data))
        return out

    def render_device(self, cam, d_out_ptr, max_depth=REFERENCE_DEPTH, tile=None, stream=None):
        x0, y0, w, h = tile if tile else (0, 0, cam.hsize, cam.vsize)
        _check_hip(hip_lib().rtc_render_device(self._s, C.byref(cam), max_depth, x0, y0, w, h, d_out_ptr, stream))

    def render_tiles_device(self, cam, d_out_ptr, tile_w, tile_h, first_tile, tile_stride, n_my_tiles,
                            max_depth=REFERENCE_DEPTH, stream=None):
        _check_hip(hip_lib().rtc_render_tiles_device(self._s, C.byref(cam), max_depth, tile_w, tile_h, first_tile,
                                                     tile_stride, n_my_tiles, d_out_ptr, stream))

    def render_tile_list_device(self, cam, d_out_ptr, tile_w, tile_h, tiles, max_depth=REFERENCE_DEPTH, stream=None):
        """A rank's share of a cost-balanced split: region k of the buffer is tile tiles[k]."""
        tiles = np.ascontiguousarray(tiles, dtype=np.uint32)
        _check_hip(hip_lib().rtc_render_tile_list_device(self._s, C.byref(cam), max_depth, tile_w, tile_h,
                                                         tiles.ctypes.data_as(_u32p), len(tiles), d_out_ptr, stream))

    def tile_costs(self, n_regions):
        """Measured cost of every region (tile) of the last measuring tile-mode render on this handle."""
        out = np.empty(n_regions, dtype=np.float64)
        _check_hip(hip_lib().rtc_get_tile_costs(self._s, out.ctypes.data_as(_dp), n_regions))
        return out

    def synchronize(self):
        _check_hip(hip_lib().rtc_scene_synchronize(self._s))

    def stats(self):
        st = Stats()
        _check_hip(hip_lib().rtc_get_stats(self._s, C.byref(st)))
        return {k: getattr(st, k) for k, _ in Stats._fields_}

    def last_kernel_name(self):
        """The render kernel the last launch on this handle ran (the name rocprofv3 shows)."""
        return hip_lib().rtc_last_kernel_name(self._s).decode()

    def schedule(self):
        """Diagnostic: the packets the next launch of the current pixel map would run, [n][16] u32 (rtc_get_schedule)."""
        n = C.c_uint32()
        st = hip_lib().rtc_get_schedule(self._s, None, 0, C.byref(n))
        if n.value == 0:
            _check_hip(st)
            return np.zeros((0, 16), dtype=np.uint32)
        out = np.empty((n.value, 16), dtype=np.uint32)
        _check_hip(hip_lib().rtc_get_schedule(self._s, out.ctypes.data_as(_u32p), out.size, C.byref(n)))
        return out[:n.value]

    def chunk_times(self, cam):
        """Diagnostic: (estimated, measured) ticks per 8x8 chunk of the last scheduled launch's pixel map (rtc_get_chunk_times)."""
        n = C.c_uint32()
        _check_hip(hip_lib().rtc_get_chunk_times(self._s, C.byref(cam), None, None, 0, C.byref(n)))
        est, got = np.zeros(n.value, dtype=np.uint32), np.zeros(n.value, dtype=np.uint32)
        if n.value:
            _check_hip(hip_lib().rtc_get_chunk_times(self._s, C.byref(cam), est.ctypes.data_as(_u32p), got.ctypes.data_as(_u32p), n.value, C.byref(n)))
        return est, got

    def close(self):
        if self._s:
            hip_lib().rtc_scene_destroy(self._s)
            self._s = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def canvas_register(array):
    """rtc_canvas_register: pins a caller-owned numpy canvas for the HIP runtime (copies into it then run at link
    speed).  The caller keeps the array alive and calls canvas_unregister before dropping it."""
    _check_hip(hip_lib().rtc_canvas_register(array.ctypes.data, array.nbytes))


def canvas_unregister(array):
    _check_hip(hip_lib().rtc_canvas_unregister(array.ctypes.data))


def build_tables_digest(desc):
    """rtc_diag_build_tables: (digest, ms) of the host half of rtc_scene_create for a scene description - no device needed."""
    digest, ms = C.c_uint64(0), C.c_double(0.0)
    _check_hip(hip_lib().rtc_diag_build_tables(C.byref(desc), C.byref(digest), C.byref(ms)))
    return digest.value, ms.value


def root_boxes(desc):
    """rtc_diag_root_boxes: (boxes [n][7] f32: lo, hi, line_only; world_index [n] u32; (cull_bmax, cull_par)) of a scene
    description, as rtc_scene_create builds them - no device needed."""
    lib = hip_lib()
    n = C.c_uint32(0)
    f32p = C.POINTER(C.c_float)
    lib.rtc_diag_root_boxes.argtypes = [C.c_void_p, f32p, _u32p, C.c_uint32, _u32p, f32p]
    _check_hip(lib.rtc_diag_root_boxes(C.byref(desc), None, None, 0, C.byref(n), None))
    boxes = np.zeros((n.value, 7), dtype=np.float32)
    order = np.zeros(n.value, dtype=np.uint32)
    scales = np.zeros(2, dtype=np.float32)
    _check_hip(lib.rtc_diag_root_boxes(C.byref(desc), boxes.ctypes.data_as(f32p), order.ctypes.data_as(_u32p), n.value, C.byref(n),
                                       scales.ctypes.data_as(f32p)))
    return boxes, order, (float(scales[0]), float(scales[1]))


def set_option(name, value):
    """rtc_set_option: a process-wide tuning / test option of the library (include/rtc_diag.h lists them)."""
    _check_hip(hip_lib().rtc_set_option(name.encode(), float(value)))


def set_loader_threads(threads):
    """rtch_set_loader_threads: threads HostScene's loader may build a scene's objects on (0: automatic, 1: one loop)."""
    host_lib().rtch_set_loader_threads(int(threads))


def canvas_ppm(rgb):
    """Canvas.ppm (canvas.zig:181-254) of an [h][w][3] f64 array."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    h, w, _ = rgb.shape
    n = host_lib().rtch_canvas_ppm(rgb.ctypes.data, w, h, None, 0)
    buf = C.create_string_buffer(n)
    host_lib().rtch_canvas_ppm(rgb.ctypes.data, w, h, buf, n)
    return buf.raw[:n].decode()


def canvas_rgba8(rgb):
    """RGBA8 framebuffer of lib.zig:146-153."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    h, w, _ = rgb.shape
    out = np.empty((h, w, 4), dtype=np.uint8)
    host_lib().rtch_canvas_rgba8(rgb.ctypes.data, w, h, out.ctypes.data)
    return out


# ---- multi-GPU tile partition: interleaved tiles, one gather, un-permute on rank 0
def tile_grid(hsize, vsize, tile_w, tile_h):
    return (hsize + tile_w - 1) // tile_w, (vsize + tile_h - 1) // tile_h


def tiles_of_rank(n_tiles, rank, world):
    """Tiles rank, rank+world, ... of the row-major tiling.  Returns (first_tile, stride, count, padded_count):
    `count` tiles are rendered, `padded_count` = ceil(n_tiles / world) is the size of every rank's buffer so that one
    equal-count gather suffices; slots count..padded_count-1 are never written (allocate the buffer zeroed)."""
    count = (n_tiles - rank + world - 1) // world if rank < n_tiles else 0
    padded = (n_tiles + world - 1) // world
    return rank, world, count, padded


def assign_tiles(tile_cost, world):
    """rtc_assign_tiles: (rank_of_tile, slot_of_tile) of a cost-balanced split; slot = rank * ceil(n / world) + k."""
    cost = np.ascontiguousarray(tile_cost, dtype=np.float64)
    rank_of = np.empty(len(cost), dtype=np.uint32)
    slot_of = np.empty(len(cost), dtype=np.uint32)
    _check_hip(hip_lib().rtc_assign_tiles(cost.ctypes.data_as(_dp), len(cost), world, rank_of.ctypes.data_as(_u32p),
                                          slot_of.ctypes.data_as(_u32p)))
    return rank_of, slot_of


def assemble_tile_list_device(d_gathered_ptr, d_slot_of_tile_ptr, tile_w, tile_h, hsize, vsize, d_canvas_ptr, stream):
    _check_hip(hip_lib().rtc_assemble_tile_list_device(d_gathered_ptr, d_slot_of_tile_ptr, tile_w, tile_h, hsize, vsize,
                                                       d_canvas_ptr, stream))


def assemble_tile_list_rgba8_device(d_gathered_rgba_ptr, d_slot_of_tile_ptr, tile_w, tile_h, hsize, vsize, d_rgba_ptr, stream):
    """Shares clamped to RGBA8 before the gather (rgba8_device on a rank's tile buffer) -> the [vsize][hsize] u32 framebuffer."""
    _check_hip(hip_lib().rtc_assemble_tile_list_rgba8_device(d_gathered_rgba_ptr, d_slot_of_tile_ptr, tile_w, tile_h, hsize, vsize,
                                                             d_rgba_ptr, stream))


def rgba8_device(d_canvas_ptr, n_pixels, d_rgba_ptr, stream):
    """rtc_rgba8_device: the clamp of color.zig:61-71 alone, device to device, asynchronous on `stream`."""
    _check_hip(hip_lib().rtc_rgba8_device(d_canvas_ptr, n_pixels, d_rgba_ptr, stream))


def assemble_tiles_device(d_gathered_ptr, world, padded, tile_w, tile_h, hsize, vsize, d_canvas_ptr, stream):
    """Device-side twin of assemble_tiles (rank 0, after the gather); asynchronous on `stream`."""
    _check_hip(hip_lib().rtc_assemble_tiles_device(d_gathered_ptr, world, padded, tile_w, tile_h, hsize, vsize,
                                                   d_canvas_ptr, stream))


def assemble_tiles(gathered, hsize, vsize, tile_w, tile_h, world):
    """gathered: [world][padded][tile_h][tile_w][3] -> [vsize][hsize][3] canvas (row-major, canvas.zig:132)."""
    tx, ty = tile_grid(hsize, vsize, tile_w, tile_h)
    n_tiles = tx * ty
    out = np.zeros((ty * tile_h, tx * tile_w, 3), dtype=gathered.dtype)
    for t in range(n_tiles):
        r, k = t % world, t // world
        y, x = divmod(t, tx)
        out[y * tile_h:(y + 1) * tile_h, x * tile_w:(x + 1) * tile_w] = gathered[r, k]
    return out[:vsize, :hsize]
