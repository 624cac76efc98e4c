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


def test_host_symbols_are_exported(rtc):
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(REPO, "include", "rtc_host.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(rtch_[a-z_0-9]+)\s*\(", text)))
    assert set(names) == set(rtc.HOST_SYMBOLS), names
    lib = rtc.host_lib()
    for n in names:
        assert getattr(lib, n) is not None


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
    d.abi_version = 2   # RTC_ABI_VERSION (node_op was added in 2)
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


def test_zig_binding_in_integration_md_matches_the_header():
    """INTEGRATION.md shows the extern struct a Zig maintainer would declare; its fields must be rtc.h's, in order."""
    import re
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(repo, "include", "rtc.h")).read()
    i = header.index("typedef struct rtc_scene_desc")
    body = re.sub(r"/\*.*?\*/", "", header[i:header.index("} rtc_scene_desc;", i)], flags=re.S)
    c_fields = re.findall(r"(?:const\s+)?(?:uint\d+_t|double|float)\s*\*?\s*(\w+)\s*;", body)
    md = open(os.path.join(repo, "INTEGRATION.md")).read()
    i = md.index("pub const RtcSceneDesc = extern struct {")
    zig_fields = re.findall(r"(\w+):", md[i:md.index("};", i)])
    assert len(c_fields) > 50 and c_fields == zig_fields
