"""CPU: the oracle reproduces the committed golden vectors; loader results are stable."""
import glob
import os
import sys

import numpy as np
import pytest

import oracle_binding as ob

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden import CASES, name_of  # noqa: E402

FAST = [c for c in CASES if c[0] not in ("dragons.json",)]  # dragons takes ~6 s to load: one case below


@pytest.mark.parametrize("scene,w,h,depth", FAST)
def test_oracle_matches_golden(rtc, scene, w, h, depth):
    g = np.load(os.path.join(HERE, "golden", name_of(scene, w, h, depth)))
    hs = rtc.HostScene.from_file(scene)
    img, counters = ob.OracleScene(hs.desc).render(hs.camera(w, h), depth)
    # same compiler flags, same libm: the oracle is deterministic to the bit on one machine; allow
    # 1e-12 for a different libm pow() on another host
    assert np.abs(img - g["image"]).max() < 1e-12
    assert [counters[k] for k in ob.COUNTER_NAMES] == g["counters"].tolist()


def test_dragons_loads_and_matches_golden(rtc):
    scene, w, h, depth = "dragons.json", 96, 54, 5
    g = np.load(os.path.join(HERE, "golden", name_of(scene, w, h, depth)))
    hs = rtc.HostScene.from_file(scene)
    d = hs.desc
    # 6 x dragon.obj (23 490 smooth triangles each) + 6 cylinders + 5 cubes (SURVEY feature matrix)
    assert d.n_leaves == 6 * 23490 + 6 + 5
    assert d.n_roots == 6 and d.n_lights == 4
    kinds = hs.array("leaf_kind", d.n_leaves)
    assert (kinds == 5).sum() == 6 * 23490 and (kinds == 3).sum() == 6 and (kinds == 2).sum() == 5
    # all triangles of one instance share one transform: 6 + 6 + 5 distinct matrices
    assert d.n_xforms == 17
    img, counters = ob.OracleScene(d).render(hs.camera(w, h), depth)
    assert np.abs(img - g["image"]).max() < 1e-12
    assert [counters[k] for k in ob.COUNTER_NAMES] == g["counters"].tolist()


def test_teapot_structure(rtc):
    hs = rtc.HostScene.from_file("teapot.json")
    d = hs.desc
    assert d.n_leaves == 6320 + 1 and d.n_roots == 2 and d.n_xforms == 2
    kinds = hs.array("leaf_kind", d.n_leaves)
    assert (kinds == 4).sum() == 6320  # flat triangles: teapot.obj has no vn records
    # divide(8): every group that was split holds fewer than 8 direct leaf children or only straddlers
    counts = hs.array("node_count", d.n_nodes)
    assert counts.max() < 6320 and d.n_nodes > 100


def test_all_golden_files_have_a_case():
    files = {os.path.basename(p) for p in glob.glob(os.path.join(HERE, "golden", "*.npz"))}
    assert files == {name_of(*c) for c in CASES}


def test_canvas_output_formats(rtc):
    """canvas.zig:258-303 KATs through the host C API (PPM P3 with 70-column wrap, RGBA8 clamp)."""
    img = np.zeros((3, 5, 3))
    img[0, 0, 0] = 1.5
    img[1, 2, 1] = 0.5
    img[2, 4] = [-0.5, 0.0, 1.0]
    assert rtc.canvas_ppm(img) == ("P3\n5 3\n255\n255 0 0 0 0 0 0 0 0 0 0 0 0 0 0\n0 0 0 0 0 0 0 128 0 0 0 0 0 0 0\n"
                                   "0 0 0 0 0 0 0 0 0 0 0 0 0 0 255\n")
    rgba = rtc.canvas_rgba8(img)
    assert rgba[0, 0].tolist() == [255, 0, 0, 255] and rgba[1, 2].tolist() == [0, 128, 0, 255]
    assert rgba[2, 4].tolist() == [0, 0, 255, 255]


def test_loader_error_names(rtc):
    with pytest.raises(rtc.RtcError) as e:
        rtc.HostScene('{"camera":{"width":2,"height":2,"field-of-view":1,"from":[0,0,-5],"to":[0,0,0],"up":[0,1,0]},'
                      '"lights":[],"objects":[{"type":{"from-definition":"nope"}}]}')
    assert e.value.name == "UnknownDefinition"   # scene.zig:212
    with pytest.raises(rtc.RtcError) as e:
        rtc.HostScene('{"camera":{"width":2,"height":2,"field-of-view":1,"from":[0,0,-5],"to":[0,0,0],"up":[0,1,0]},'
                      '"lights":[],"objects":[{"type":{"cube":{}},"transform":[{"scale":[0,0,0]}]}]}')
    assert e.value.name == "NotInvertible"       # matrix.zig:7
    with pytest.raises(rtc.RtcError) as e:
        rtc.HostScene("{ not json")
    assert e.value.name == "SyntaxError"


def test_camera_override_matches_camera_new(rtc):
    """Width/height overrides re-run Camera.new (camera.zig:33-52), including the aspect < 1 branch."""
    hs = rtc.HostScene.from_file("cover.json")
    c0 = hs.camera()
    assert (c0.hsize, c0.vsize) == (1280, 1280)
    c1 = hs.camera(1920, 1080)
    import math
    half_view = math.tan(0.785 / 2.0)
    assert c1.half_width == half_view and c1.half_height == half_view / (1920 / 1080)
    assert c1.pixel_size == (half_view * 2.0) / 1920
    c2 = hs.camera(1080, 1920)
    assert c2.half_height == half_view and c2.half_width == half_view * (1080 / 1920)
    assert list(c1.inv_view) == list(c0.inv_view)
