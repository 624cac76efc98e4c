"""The product's scene build against an INDEPENDENT one.

Every GPU parity test feeds the oracle the flat description the PRODUCT loader made (tests/test_parity_gpu.py): a wrong
transform order, `from-definition` double inherit, divide(8) child order, OBJ normalisation or re-boxing in
host/rtc_loader.cpp / rtc_scene.cpp / rtc_flatten.cpp would move GPU and oracle together.  Here the oracle builds its
own World from the same scene JSON and OBJ bytes with its own restatement of scene.zig / obj.zig / shape.zig /
group.zig / bounding_box.zig (oracle/rtc_oracle_scene.hpp: own JSON reader, the oracle's own Matrix, no product code),
and the two results are compared table by table, BIT FOR BIT, in the canonical depth-first order of World.objects:
leaf count and order, kinds, Shape.ids, every inverse matrix and its transpose, cylinder / cone limits, triangle p1 / e1
/ e2 / normals, casts_shadow, materials and whole pattern trees (incl. texture maps and image pixels), every Group /
Csg box, child lists, operations, World.objects, lights, camera.  CPU only; runs in `-m "not gpu"`."""
import ctypes as C
import importlib
import json
import os
import struct
import zlib

import numpy as np
import pytest

import oracle_binding as ob

rtc = importlib.import_module("ray-tracer-challenge_amd")
BIT = rtc.RTC_CHILD_NODE_BIT

GOLDEN = sorted(f[:-5] for f in os.listdir(rtc.SCENE_DIR) if f.endswith(".json"))
IMAGES = ("earthmap1k.png",)


def _arr(hs, field, count, width=1):
    return np.array(hs.array(field, count, width))  # a copy: plain numpy from here on


def _product_tables(hs, cam):
    """The product's rtc_scene_desc re-ordered canonically: leaves and nodes numbered by the depth-first walk of roots[]."""
    d = hs.desc
    kind, lx, lm = _arr(hs, "leaf_kind", d.n_leaves), _arr(hs, "leaf_xform", d.n_leaves), _arr(hs, "leaf_material", d.n_leaves)
    shadow, lid, geom = _arr(hs, "leaf_shadow", d.n_leaves), _arr(hs, "leaf_id", d.n_leaves), _arr(hs, "leaf_geom", d.n_leaves)
    inv, inv_t = _arr(hs, "xf_inv", d.n_xforms, 16), _arr(hs, "xf_inv_t", d.n_xforms, 16)
    first, count = _arr(hs, "node_first", d.n_nodes), _arr(hs, "node_count", d.n_nodes)
    children, roots = _arr(hs, "children", d.n_children), _arr(hs, "roots", d.n_roots)
    leaves, nodes, node_children = [], [], []

    def visit(ref):
        ref = int(ref)
        if ref & BIT:
            n = ref & ~BIT
            ordinal = len(nodes)
            nodes.append(n)
            node_children.append(None)
            node_children[ordinal] = [visit(children[first[n] + i]) for i in range(count[n])]
            return ordinal | BIT
        leaves.append(ref)
        return len(leaves) - 1

    canon_roots = [visit(r) for r in roots]
    L, N = np.array(leaves, dtype=np.int64), np.array(nodes, dtype=np.int64)
    t = {"kind": kind[L], "id": lid[L].astype(np.uint64), "shadow": shadow[L], "inv": inv[lx[L]], "inv_t": inv_t[lx[L]],
         "leaf_material": lm[L], "geom": geom[L]}
    t["box"] = np.concatenate([_arr(hs, "node_min", d.n_nodes, 3)[N], _arr(hs, "node_max", d.n_nodes, 3)[N]], axis=1) if len(N) else np.zeros((0, 6))
    t["op"] = _arr(hs, "node_op", d.n_nodes)[N] if len(N) else np.zeros(0, np.uint8)
    t["count"] = count[N] if len(N) else np.zeros(0, np.uint32)
    t["children"] = np.array([c for kids in node_children for c in kids], dtype=np.uint32)
    t["roots"] = np.array(canon_roots, dtype=np.uint32)
    t["lights"] = np.concatenate([_arr(hs, "light_pos", d.n_lights, 3), _arr(hs, "light_rgb", d.n_lights, 3)], axis=1) if d.n_lights else np.zeros((0, 6))
    # geometry
    cyl = np.zeros((len(L), 3))
    tri = np.zeros((len(L), 18))
    cmin, cmax, cclosed = _arr(hs, "cyl_min", d.n_cyls), _arr(hs, "cyl_max", d.n_cyls), _arr(hs, "cyl_closed", d.n_cyls)
    T = [_arr(hs, f, d.n_tris, 3) for f in ("tri_p1", "tri_e1", "tri_e2", "tri_n1", "tri_n2", "tri_n3")]
    is_cyl = (t["kind"] == 3) | (t["kind"] == 6)
    if is_cyl.any():
        g = t["geom"][is_cyl]
        cyl[is_cyl] = np.stack([cmin[g], cmax[g], cclosed[g].astype(np.float64)], axis=1)
    is_tri = (t["kind"] == 4) | (t["kind"] == 5)
    if is_tri.any():
        g = t["geom"][is_tri]
        tri[is_tri] = np.concatenate([a[g] for a in T], axis=1)
    t["cyl"], t["tri"], t["is_cyl"], t["is_tri"] = cyl, tri, is_cyl, is_tri
    # materials, serialised like oracle_capi.cpp's serialiseMaterial
    pk, pinv, prgb = _arr(hs, "pat_kind", d.n_patterns), _arr(hs, "pat_inv", d.n_patterns, 16), _arr(hs, "pat_rgb", d.n_patterns, 3)
    pa, pb = _arr(hs, "pat_a", d.n_patterns), _arr(hs, "pat_b", d.n_patterns)
    tex_mapping, tex_uv = _arr(hs, "tex_mapping", d.n_texmaps), _arr(hs, "tex_uv", d.n_texmaps, 6)
    uv_kind, uv_size, uv_sub = _arr(hs, "uv_kind", d.n_uvs), _arr(hs, "uv_size", d.n_uvs, 2), _arr(hs, "uv_sub", d.n_uvs, 5)
    uv_image, uv_interp = _arr(hs, "uv_image", d.n_uvs), _arr(hs, "uv_interp", d.n_uvs)
    img_w, img_h, img_off = _arr(hs, "img_width", d.n_images), _arr(hs, "img_height", d.n_images), _arr(hs, "img_offset", d.n_images)
    crc = {}

    def image_crc(i):
        if i not in crc:
            n = int(img_w[i]) * int(img_h[i]) * 3
            px = np.ctypeslib.as_array(d.img_rgb, shape=(int(img_off[i]) * 3 + n,))[int(img_off[i]) * 3:]
            crc[i] = zlib.crc32(px.astype(np.float64).tobytes()) & 0xFFFFFFFF
        return crc[i]

    def ser(i):
        k = int(pk[i])
        out = bytes([k]) + pinv[i].tobytes()
        if k == 0:
            return out + prgb[i].tobytes()
        if k == 9:
            return out
        if k == 7:
            return out + prgb[i].tobytes() + ser(pa[i])
        if k == 8:
            tm = int(pa[i])
            out += bytes([int(tex_mapping[tm])])
            for f in range(6 if tex_mapping[tm] == 3 else 1):
                u = int(tex_uv[tm][f])
                out += bytes([int(uv_kind[u])])
                if uv_kind[u] == 0:
                    out += b"".join(ser(uv_sub[u][j]) for j in range(5))
                elif uv_kind[u] == 1:
                    out += uv_size[u].tobytes() + ser(uv_sub[u][0]) + ser(uv_sub[u][1])
                elif uv_kind[u] == 2:
                    im = int(uv_image[u])
                    out += struct.pack("<IIBI", int(img_w[im]), int(img_h[im]), int(uv_interp[u] != 0), image_crc(im))
            return out
        return out + ser(pa[i]) + ser(pb[i])

    mp, mpat = _arr(hs, "mat_params", d.n_materials, rtc.RTC_MAT_STRIDE), _arr(hs, "mat_pattern", d.n_materials)
    t["materials"] = [mp[i].tobytes() + ser(mpat[i]) for i in range(d.n_materials)]
    t["camera"] = cam
    return t


def _same_bits(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.dtype.kind == "f":
        bad = a.view(np.uint64) != b.view(np.uint64)
    else:
        bad = a != b
    assert not bad.any(), f"{what}: {int(bad.sum())} entries differ, first at {np.argwhere(bad)[0].tolist()}: {a[bad][0]!r} vs {b[bad][0]!r}"


def _compare(scene_json, data_dir, width=0, height=0, images=()):
    hs = rtc.HostScene(scene_json, data_dir)
    ours = _product_tables(hs, hs.camera(width, height))
    built = ob.BuiltScene(scene_json, data_dir, width, height, images)
    ref = built.tables()
    assert (built.n_leaves, built.n_nodes, built.n_children, built.n_roots, built.n_lights) == \
        (hs.desc.n_leaves, hs.desc.n_nodes, hs.desc.n_children, hs.desc.n_roots, hs.desc.n_lights)
    for key in ("kind", "shadow", "inv", "inv_t", "box", "op", "count", "children", "roots", "lights"):
        _same_bits(ours[key], ref[key], key)
    _same_bits(ours["cyl"][ours["is_cyl"]], ref["cyl"][ours["is_cyl"]], "cylinder / cone limits")
    flat = ours["kind"] == 4
    _same_bits(ours["tri"][flat][:, :12], ref["tri"][flat][:, :12], "triangle p1 / e1 / e2 / normal")
    smooth = ours["kind"] == 5
    _same_bits(ours["tri"][smooth], ref["tri"][smooth], "smooth triangle p1 / e1 / e2 / n1 / n2 / n3")
    # Shape.id: what the path uses is identity (the containers walk, csg's includes()); the reference's ids are those of
    # one process-wide counter that bounding boxes bump too - the oracle build restates every bump, so the product's ids
    # must be the same numbers relative to the first Shape of the parse, or at least induce the same identity classes
    ids_ours, ids_ref = ours["id"].astype(np.int64), ref["id"].astype(np.int64)
    assert len(np.unique(ids_ref)) == len(ids_ref), "the reference restatement gives every leaf its own id"
    assert len(np.unique(ids_ours)) == len(ids_ours), "product leaves share a Shape.id"
    exact_ids = bool(len(ids_ours) == 0 or ((ids_ours - ids_ours.min()) == (ids_ref - ids_ref.min())).all())
    # materials: every leaf's material and pattern tree, byte for byte
    for m in np.unique(ours["leaf_material"]):
        leaves = np.nonzero(ours["leaf_material"] == m)[0]
        for rm in np.unique(ref["material"][leaves]):
            assert ours["materials"][m] == ref["materials"][rm], f"material {m} (leaf {leaves[0]}) differs from the reference build's"
    cam, rcam = ours["camera"], built.camera()
    assert (cam.hsize, cam.vsize) == (rcam.hsize, rcam.vsize)
    _same_bits(np.array([cam.half_width, cam.half_height, cam.pixel_size] + list(cam.inv_view)),
               np.array([rcam.half_width, rcam.half_height, rcam.pixel_size] + list(rcam.inv_view)), "camera")
    return hs, built, exact_ids


@pytest.mark.parametrize("name", GOLDEN)
def test_product_loader_equals_the_independent_build(name):
    """All 13 scene files of the reference (and the 4 of this repo), every table bit for bit."""
    with open(os.path.join(rtc.SCENE_DIR, name + ".json")) as f:
        js = f.read()
    hs, built, exact_ids = _compare(js, rtc.DATA_DIR + os.sep, images=IMAGES)
    print(f"{name}: {built.n_leaves} leaves, {built.n_nodes} nodes, {built.n_materials} materials, ids exact: {exact_ids}, "
          f"OBJ lines ignored {built.lines_ignored}")


@pytest.mark.parametrize("name,w,h", [("cover", 1920, 1080), ("cover", 160, 200), ("dragons", 3840, 2160)])
def test_camera_with_the_size_overridden(name, w, h):
    with open(os.path.join(rtc.SCENE_DIR, name + ".json")) as f:
        js = f.read()
    _compare(js, rtc.DATA_DIR + os.sep, w, h, images=IMAGES)


CAMERA = {"width": 40, "height": 30, "field-of-view": 0.9, "from": [1, 2, -6], "to": [0, 0.5, 0], "up": [0, 1, 0]}
LIGHTS = [{"point-light": {"position": [-5, 8, -7], "intensity": [1, 0.9, 0.8]}}]
GLASS = {"transparency": 0.9, "refractive-index": 1.5, "reflective": 0.3, "diffuse": 0.1}
RED = {"pattern": {"type": {"solid": [1, 0.1, 0.1]}}, "shininess": 50}
STRIPED = {"pattern": {"type": {"stripes": [{"type": {"solid": [1, 1, 1]}}, {"type": {"checkers": [{"type": {"solid": [0, 0, 1]}}, {"type": {"solid": [0, 1, 0]}, "transform": [{"scale": [0.3, 0.3, 0.3]}]}]}, "transform": [{"rotate-y": 0.4}]}]},
                       "transform": [{"scale": [0.25, 1, 1]}, {"rotate-z": 0.7}, {"shear": {"xy": 0.5, "zx": -0.25}}]}, "ambient": 0.2}

OBJ_TEXT = """# a comment line (ignored)
v -1 1 0
v -1.0000 0.5000 0.0000
v 1 0 0
v 1 1 0
v 0 2 0

vn 0 0 1
vn 0.707 0 -0.707
vn 1 2 3
g First
f 1 2 3
f 1/0/3 2/102/1 3/14/2
g Second Group
f 1//1 3//2 4//3 5//1
f 1 2
s 1
f 1 2 3 4 5
"""


def _cases():
    ring = [{"type": {"sphere": {}}, "transform": [{"scale": [0.3, 0.3, 0.3]}, {"translate": [2 * np.cos(a), 0.3, 2 * np.sin(a)]}]}
            for a in np.linspace(0, 6, 11)]
    leaf_def = {"name": "ball", "value": {"type": {"sphere": {}}, "transform": [{"scale": [0.5, 0.5, 0.5]}, {"translate": [0, 1, 0]}], "material": RED}}
    group_def = {"name": "pair", "value": {"type": {"group": [{"type": {"from-definition": "ball"}, "transform": [{"translate": [-1, 0, 0]}]},
                                                               {"type": {"cube": {}}, "transform": [{"scale": [0.4, 0.4, 0.4]}, {"translate": [1, 0.4, 0]}], "casts-shadow": False}]},
                                           "transform": [{"rotate-y": 0.3}], "material": STRIPED}}
    nested_def = {"name": "nest", "value": {"type": {"group": [{"type": {"from-definition": "pair"}, "transform": [{"scale": [0.5, 0.5, 0.5]}]},
                                                               {"type": {"group": ring}, "transform": [{"translate": [0, 0.2, 0]}]}]},
                                            "transform": [{"translate": [0, 0, 1]}]}}
    obj_def = {"name": "mesh", "value": {"type": {"from-obj": {"file": "little.obj", "normalize": False}}, "transform": [{"scale": [0.5, 0.5, 0.5]}], "material": RED}}
    yield "leaf definitions, re-inherited", {"shape-definitions": [leaf_def], "camera": CAMERA, "lights": LIGHTS, "objects": [
        {"type": {"from-definition": "ball"}},
        {"type": {"from-definition": "ball"}, "transform": [{"translate": [2, 0, 0]}, {"rotate-x": 0.2}], "material": GLASS, "casts-shadow": False},
        {"type": {"plane": {}}, "material": STRIPED}]}
    yield "group definitions inside group definitions", {"shape-definitions": [leaf_def, group_def, nested_def], "camera": CAMERA, "lights": LIGHTS, "objects": [
        {"type": {"from-definition": "nest"}, "transform": [{"rotate-z": 0.1}, {"translate": [0, 0.5, 0]}], "material": GLASS},
        {"type": {"from-definition": "pair"}, "casts-shadow": False},
        {"type": {"group": [{"type": {"from-definition": "nest"}}, {"type": {"cone": {"min": -1, "max": 0.5, "closed": True}}}]}, "transform": [{"scale": [1, 2, 1]}]}]}
    yield "nested groups that divide at several levels", {"camera": CAMERA, "lights": LIGHTS, "objects": [
        {"type": {"group": [{"type": {"group": ring}, "transform": [{"translate": [0, k, 0]}]} for k in range(9)] + ring}, "transform": [{"rotate-y": 1.0}], "material": RED},
        {"type": {"cylinder": {"min": 0, "max": 2}}, "transform": [{"translate": [4, 0, 0]}]}]}
    yield "OBJ: named groups, fans, normals, ignored lines, normalize on and off, from a definition", {"shape-definitions": [obj_def], "camera": CAMERA, "lights": LIGHTS, "objects": [
        {"type": {"from-obj": {"file": "little.obj"}}, "transform": [{"translate": [0, 1, 0]}], "material": STRIPED, "casts-shadow": False},
        {"type": {"from-obj": {"file": "little.obj", "normalize": False}}},
        {"type": {"from-definition": "mesh"}, "transform": [{"translate": [-2, 0, 0]}], "material": GLASS},
        {"type": {"group": [{"type": {"from-definition": "mesh"}}, {"type": {"triangle": {"p1": [0, 0, 0], "p2": [1, 0, 0], "p3": [0, 1, 0.5]}}}]}, "transform": [{"rotate-x": -0.5}]}]}
    yield "csg of groups and definitions", {"shape-definitions": [leaf_def, group_def], "camera": CAMERA, "lights": LIGHTS, "objects": [
        {"type": {"csg": {"operation": "difference", "left": {"type": {"from-definition": "pair"}},
                          "right": {"type": {"csg": {"operation": "intersection", "left": {"type": {"sphere": {}}, "transform": [{"translate": [0.5, 0.5, 0]}]},
                                                     "right": {"type": {"cube": {}}, "material": GLASS}}}}}},
         "transform": [{"translate": [0, 1, 0]}, {"rotate-y": 0.5}], "material": RED}]}
    pat = lambda t: {"pattern": {"type": t, "transform": [{"scale": [0.5, 0.5, 0.5]}]}}
    solid = lambda r, g, b: {"type": {"solid": [r, g, b]}}
    yield "every pattern kind, nested", {"camera": CAMERA, "lights": LIGHTS, "objects": [
        {"type": {"sphere": {}}, "material": pat({"blend": [{"type": {"gradient": [solid(1, 0, 0), solid(0, 0, 1)]}}, {"type": {"rings": [solid(1, 1, 1), solid(0, 0, 0)]}}]})},
        {"type": {"cube": {}}, "transform": [{"translate": [3, 0, 0]}], "material": pat({"perturb": {"type": {"radial-gradient": [solid(0, 1, 0), solid(1, 1, 0)]}}})},
        {"type": {"plane": {}}, "material": pat({"texture-map": {"planar": {"uv-pattern": {"checkers": {"width": 2, "height": 3, "patterns": [solid(1, 1, 1), solid(0.2, 0.2, 0.2)]}}}}})},
        {"type": {"cylinder": {"min": 0, "max": 1, "closed": True}}, "transform": [{"translate": [-3, 0, 0]}],
         "material": pat({"texture-map": {"cylindrical": {"uv-pattern": {"align-check": {"central": solid(1, 1, 1), "upper-left": solid(1, 0, 0), "upper-right": solid(1, 1, 0), "bottom-left": solid(0, 1, 0), "bottom-right": solid(0, 1, 1)}}}}})}]}


@pytest.mark.parametrize("case", list(_cases()), ids=lambda c: c[0])
def test_loader_quirks_equal_the_independent_build(case, tmp_path):
    """`from-definition` (leaf and group definitions, re-inherited material / transform / casts-shadow, definitions inside
    definitions), groups that divide at several levels, OBJ quirks (named groups, fan triangulation, a//n and a/t/n
    faces, ignored lines, normalize on / off), csg of groups, every pattern kind: tables bit for bit, and then the two
    Worlds rendered by the oracle - the one built from the product's description, the one the oracle built itself -
    must give the same image and the same ray counts."""
    (tmp_path / "little.obj").write_text(OBJ_TEXT)
    js = json.dumps(case[1])
    hs, built, exact_ids = _compare(js, str(tmp_path) + os.sep)
    cam = hs.camera()
    want, counters = ob.OracleScene(hs.desc).render(cam, 5)
    got, counters2 = built.render(5)
    assert np.array_equal(want, got) and counters == counters2
    print(case[0], ": leaves", built.n_leaves, "nodes", built.n_nodes, "ids exact:", exact_ids, "lines ignored", built.lines_ignored)


def test_obj_lines_ignored_matches_the_reference_test():
    """obj.zig:288-304 ('Ignoring unrecognized lines': five lines of gibberish are five ignored lines) on the oracle's parser."""
    gibberish = "There was a young lady named Bright\nwho traveled much faster than light.\nShe set out one day\nin a relative way,\nand came back the previous night.\n"
    scene = {"camera": CAMERA, "lights": [], "objects": [{"type": {"from-obj": {"file": "g.obj", "normalize": False}}}]}
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "g.obj"), "w") as f:
            f.write(gibberish)
        built = ob.BuiltScene(json.dumps(scene), d + os.sep)
        assert built.lines_ignored == 5 and built.n_leaves == 0


def test_oracle_built_dragons_renders_like_the_product_description():
    """dragons.json 96x54 on the CPU: the World the oracle built itself against the World rebuilt from the product's
    description - same image, same counters (the GPU twin of this test is in test_parity_gpu.py)."""
    with open(os.path.join(rtc.SCENE_DIR, "dragons.json")) as f:
        js = f.read()
    hs = rtc.HostScene(js, rtc.DATA_DIR + os.sep)
    built = ob.BuiltScene(js, rtc.DATA_DIR + os.sep, 96, 54)
    got, c1 = built.render(5)
    want, c2 = ob.OracleScene(hs.desc).render(hs.camera(96, 54), 5)
    assert np.array_equal(got, want) and c1 == c2
