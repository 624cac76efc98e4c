"""The C-ABI library loads and exports every symbol include/rtc.h declares (no compute calls)."""
import ctypes as C
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtc_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported(rtc):
    names = _declared_functions(os.path.join(REPO, "include", "rtc.h"))
    assert set(names) == set(rtc.RTC_SYMBOLS), names
    lib = rtc.hip_lib()
    for n in names:
        assert getattr(lib, n) is not None


def test_diagnostics_live_in_their_own_header(rtc):
    """include/rtc.h is what a Zig host binds; rtc_set_option and the three diagnostic getters are declared in
    include/rtc_diag.h only (and exported by the same library)."""
    diag = _declared_functions(os.path.join(REPO, "include", "rtc_diag.h"))
    assert set(diag) == set(rtc.RTC_DIAG_SYMBOLS), diag
    assert not set(diag) & set(_declared_functions(os.path.join(REPO, "include", "rtc.h")))
    lib = rtc.hip_lib()
    for n in diag:
        assert getattr(lib, n) is not None


def test_host_symbols_are_exported(rtc):
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(REPO, "include", "rtc_host.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(rtch_[a-z_0-9]+)\s*\(", text)))
    assert set(names) == set(rtc.HOST_SYMBOLS), names
    lib = rtc.host_lib()
    for n in names:
        assert getattr(lib, n) is not None


def test_multi_symbols_are_exported(rtc):
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(REPO, "include", "rtc_multi.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(rtc_multi_[a-z_0-9]+)\s*\(", text)))
    assert set(names) == set(rtc.MULTI_SYMBOLS), names
    lib = rtc.multi_lib()
    for n in names:
        assert getattr(lib, n) is not None


def test_headers_are_valid_c():
    """include/*.h is a C ABI: every header compiles as strict C99 on its own (a Zig @cImport or a C host sees this)."""
    import subprocess
    for name in ("rtc.h", "rtc_diag.h", "rtc_host.h", "rtc_multi.h"):
        r = subprocess.run(["gcc", "-x", "c", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only",
                            os.path.join(REPO, "include", name)], capture_output=True, text=True)
        assert r.returncode == 0, (name, r.stderr)


def test_library_reads_no_environment(rtc):
    """Tuning and test knobs go through rtc_set_option; the product library does not read the environment (the
    RTC_PROFILE diagnostics build does, behind its #ifdef)."""
    src = os.path.join(REPO, "ray-tracer-challenge_amd", "csrc")
    for name in os.listdir(src):
        text = open(os.path.join(src, name)).read()
        text = re.sub(r"#ifdef RTC_PROFILE.*?#endif", "", text, flags=re.S)
        assert "getenv" not in text, name
    lib = rtc.hip_lib()
    assert lib.rtc_set_option(b"simple3_min_chunks", -1.0) == 0
    assert lib.rtc_set_option(b"no_such_option", 1.0) == 1
    assert b"no_such_option" in lib.rtc_last_error()


def test_status_names(rtc):
    lib = rtc.hip_lib()
    assert lib.rtc_status_name(0) == b"Ok"
    assert lib.rtc_status_name(3) == b"NotInvertible"   # matrix.zig:7
    assert lib.rtc_status_name(2) == b"OutOfMemory"


def test_struct_layout_matches_header(rtc):
    # rtc_camera: 2 u32 + 3 f64 + 16 f64
    assert C.sizeof(rtc.Camera) == 8 + 8 * 19
    # rtc_scene_desc: 13 counters padded to pointer alignment + 37 pointers; spot-check offsets
    assert rtc.SceneDesc.xf_inv.offset == 8
    assert rtc.SceneDesc.n_leaves.offset == 24
    assert rtc.SceneDesc.leaf_kind.offset == 32
    assert C.sizeof(rtc.Stats) == 40


def test_create_rejects_bad_scenes_without_gpu(rtc):
    """Validation runs on the host before any HIP call, so these work without a GPU."""
    hs = rtc.HostScene.from_file("fresnel.json")
    d = hs.desc
    lib = rtc.hip_lib()
    out = C.c_void_p()
    # wrong ABI version
    d.abi_version = 99
    assert lib.rtc_scene_create(C.byref(d), C.byref(out)) == 1
    assert b"InvalidArgument" in lib.rtc_last_error()
    d.abi_version = 3   # RTC_ABI_VERSION (2: node_op; 3: rtc_canvas_register / rtc_set_option / rtc_rgba8_device, rtc_render keeps no pointer)
    # a non-affine inverse (last row != (0,0,0,1))
    xf = hs.array("xf_inv", d.n_xforms, 16)
    saved = xf[0, 12]
    xf[0, 12] = 0.5
    assert lib.rtc_scene_create(C.byref(d), C.byref(out)) == 7
    assert b"NotAffine" in lib.rtc_last_error()
    xf[0, 12] = saved
    # out-of-range material index
    lm = hs.array("leaf_material", d.n_leaves)
    lm[0] = 1000
    assert lib.rtc_scene_create(C.byref(d), C.byref(out)) == 5
    lm[0] = 0
    # duplicate Shape.id
    ids = hs.array("leaf_id", d.n_leaves)
    saved_id = int(ids[1])
    ids[1] = ids[0]
    assert lib.rtc_scene_create(C.byref(d), C.byref(out)) == 4
    ids[1] = saved_id
    # a csg node must have exactly a left and a right child
    hs2 = rtc.HostScene.from_file("csg.json")
    d2 = hs2.desc
    cnt = hs2.array("node_count", d2.n_nodes)
    cnt[0] = 1
    assert lib.rtc_scene_create(C.byref(d2), C.byref(out)) == 1
    assert b"csg node" in lib.rtc_last_error()
    cnt[0] = 2
    ops = hs2.array("node_op", d2.n_nodes)
    ops[0] = 9
    assert lib.rtc_scene_create(C.byref(d2), C.byref(out)) == 1
    ops[0] = 3


def test_diagnostic_entry_points_refuse_null_handles(rtc):
    """rtc_get_schedule / rtc_last_kernel_name (diagnostics of include/rtc_diag.h) check their arguments before they touch the
    device: callable on a box without one."""
    import ctypes as C
    lib = rtc.hip_lib()
    n = C.c_uint32(7)
    assert lib.rtc_get_schedule(None, None, 0, C.byref(n)) != 0 and b"null" in lib.rtc_last_error()
    assert lib.rtc_get_schedule(None, None, 0, None) != 0
    assert lib.rtc_last_kernel_name(None) == b""


def test_zig_binding_matches_the_header():
    """integration/gpu.zig (the reference-side binding, shipped as source) and the excerpt of it in INTEGRATION.md declare
    the extern struct a Zig maintainer binds: its fields must be rtc.h's, in order; every function of rtc.h that the
    binding uses must exist in the header with the same number of parameters; and the Flattener must fill every field."""
    import re
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(repo, "include", "rtc.h")).read()
    i = header.index("typedef struct rtc_scene_desc")
    body = re.sub(r"/\*.*?\*/", "", header[i:header.index("} rtc_scene_desc;", i)], flags=re.S)
    c_fields = re.findall(r"(?:const\s+)?(?:uint\d+_t|double|float)\s*\*?\s*(\w+)\s*;", body)
    assert len(c_fields) > 50
    zig = open(os.path.join(repo, "integration", "gpu.zig")).read()
    for name, text in (("INTEGRATION.md", open(os.path.join(repo, "INTEGRATION.md")).read()), ("integration/gpu.zig", zig)):
        i = text.index("pub const RtcSceneDesc = extern struct {")
        zig_fields = re.findall(r"(\w+):", text[i:text.index("};", i)])
        assert c_fields == zig_fields, name
    # the camera and stats structs
    for zig_name, c_name in (("RtcCamera", "rtc_camera"), ("RtcStats", "rtc_stats")):
        i = header.index("typedef struct " + c_name)
        cb = re.sub(r"/\*.*?\*/", "", header[i:header.index("} " + c_name + ";", i)], flags=re.S)
        cf = [n for decl in re.findall(r"(?:uint\d+_t|double)\s+([^;]+);", cb) for n in re.findall(r"(\w+)(?:\[\d+\])?\s*(?:,|$)", decl)]
        i = zig.index("pub const %s = extern struct {" % zig_name)
        zf = re.findall(r"(\w+):", zig[i:zig.index("};", i)])
        assert cf == zf, (zig_name, cf, zf)
    # extern functions: declared in a header, same arity
    headers = re.sub(r"/\*.*?\*/", "", header + open(os.path.join(repo, "include", "rtc_multi.h")).read() +
                     open(os.path.join(repo, "include", "rtc_diag.h")).read(), flags=re.S)
    externs = re.findall(r"pub extern fn (\w+)\(([^)]*)\)", zig)
    assert len(externs) >= 10
    for fn, params in externs:
        m = re.search(r"\b%s\s*\(([^)]*)\)" % fn, headers)
        assert m, fn
        c_n = 0 if m.group(1).strip() in ("", "void") else m.group(1).count(",") + 1
        z_n = len(re.findall(r"\b[A-Za-z_]\w*\s*:", params))   # (`name: type` pairs; the colon of a `[*:0]` sentinel has no name before it)
        assert c_n == z_n, (fn, c_n, z_n)
    # the Flattener's desc() names every field of the struct
    i = zig.index("pub fn desc(self: *const Self) RtcSceneDesc {")
    filled = re.findall(r"\.(\w+) =", zig[i:zig.index("};", i)])
    assert filled == c_fields


_ONE_RUNTIME_PROBE = r"""
import importlib, sys
sys.path.insert(0, %(repo)r)
def load_rtc():
    rtc = importlib.import_module("ray-tracer-challenge_amd")
    rtc.hip_lib()
def load_torch():
    import torch
    torch.cuda.device_count()
order = sys.argv[1]
(load_rtc(), load_torch()) if order == "rtc_first" else (load_torch(), load_rtc())
seen = set()
for line in open("/proc/self/maps"):
    path = line.split()[-1]
    if "libamdhip64" in path or "libhsa-runtime64" in path:
        seen.add(path)
print("\n".join(sorted(seen)))
"""


def test_one_hip_runtime_whatever_the_import_order():
    """The round-1 hang ("first torch.cuda call after the library had rendered"): PyTorch bundles its own
    libamdhip64 / libhsa-runtime64 (no SONAME), librtc_hip.so asks for ROCm's libamdhip64.so.7, and a process that
    loaded this package before torch had BOTH runtimes mapped, torch's own libraries split between them.
    hip_lib() maps torch's copy first; here: fresh processes, both orders, exactly one copy of each runtime."""
    import subprocess
    import sys
    import pytest
    pytest.importorskip("torch")
    for order in ("rtc_first", "torch_first"):
        out = subprocess.run([sys.executable, "-c", _ONE_RUNTIME_PROBE % {"repo": REPO}, order],
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        libs = [l for l in out.stdout.splitlines() if l.strip()]
        hip = [l for l in libs if "libamdhip64" in l]
        hsa = [l for l in libs if "libhsa-runtime64" in l]
        assert len(hip) == 1 and len(hsa) == 1, (order, libs)


def test_create_rejects_pattern_chains_the_device_would_cut_short(rtc):
    """pattern_chain() follows at most 64 patterns: a deeper chain, or a cycle, is refused at create (host-side)."""
    import json

    def nested(n):
        p = {"type": {"solid": [1, 0, 0]}}
        for _ in range(n):
            p = {"type": {"stripes": [p, {"type": {"solid": [0, 1, 0]}}]}}
        return p

    def scene(n):
        return json.dumps({"camera": {"width": 8, "height": 8, "field-of-view": 1.0, "from": [0, 0, -5], "to": [0, 0, 0], "up": [0, 1, 0]},
                           "lights": [], "objects": [{"type": {"sphere": {}}, "material": {"pattern": nested(n)}}]})

    lib = rtc.hip_lib()
    out = C.c_void_p()
    hs = rtc.HostScene(scene(70))
    assert lib.rtc_scene_create(C.byref(hs.desc), C.byref(out)) == 4
    assert b"more than 64 nested patterns" in lib.rtc_last_error()
    hs = rtc.HostScene(scene(5))
    d = hs.desc
    kinds = hs.array("pat_kind", d.n_patterns)
    a = hs.array("pat_a", d.n_patterns)
    stripes = [i for i in range(d.n_patterns) if kinds[i] == 1]
    a[stripes[0]] = stripes[0]                      # a pattern that is its own sub-pattern
    assert lib.rtc_scene_create(C.byref(d), C.byref(out)) == 4
    assert b"cycle" in lib.rtc_last_error()


def test_assign_tiles_balances_and_fills_equal_buffers(rtc):
    """rtc_assign_tiles (host only): every tile exactly once, no rank above ceil(n / world) tiles (one equal-count gather),
    slots unique, deterministic, and on a skewed cost map far better balanced than dealing the tiles round-robin."""
    import numpy as np
    rng = np.random.default_rng(5)
    for n_tiles, world in ((510, 8), (510, 4), (17, 3), (5, 8), (64, 1)):
        cost = rng.uniform(1.0, 2.0, n_tiles)
        blob = rng.choice(n_tiles, size=max(1, n_tiles // 20), replace=False)
        cost[blob] *= 40.0                                   # a glass sphere: a few tiles cost forty times the rest
        rank_of, slot_of = rtc.assign_tiles(cost, world)
        padded = (n_tiles + world - 1) // world
        assert rank_of.max() < world and len(set(slot_of.tolist())) == n_tiles
        assert np.array_equal(slot_of // padded, rank_of) and (slot_of % padded).max() < padded
        counts = np.bincount(rank_of, minlength=world)
        assert counts.max() <= padded
        for r in range(world):                               # a rank's slots follow increasing tile order
            mine = np.flatnonzero(rank_of == r)
            assert np.array_equal(slot_of[mine] - r * padded, np.arange(len(mine)))
        again = rtc.assign_tiles(cost, world)
        assert np.array_equal(again[0], rank_of) and np.array_equal(again[1], slot_of)
        if world > 1 and n_tiles >= 8 * world:
            load = np.bincount(rank_of, weights=cost, minlength=world)
            rr = np.bincount(np.arange(n_tiles) % world, weights=cost, minlength=world)
            assert load.max() <= rr.max() + 1e-9
            assert load.max() / load.mean() < 1.10, (n_tiles, world, load)
