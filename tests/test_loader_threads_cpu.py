"""parseScene builds the entries of "objects" on several threads (host/rtc_loader.cpp; scene.zig:650-655 is one loop) and
numbers them afterwards with the ids the one loop would have drawn.  The World - and the rtc_scene_desc flattened from it -
must be the same to the bit whatever the thread count: every table, and the Shape ids relative to the parse's first."""
import importlib
import json
import os

import numpy as np
import pytest

import test_oracle_scene_cpu as t

rtc = importlib.import_module("ray-tracer-challenge_amd")


@pytest.fixture(autouse=True)
def _automatic_again():
    yield
    rtc.set_loader_threads(0)


def _tables(js, threads):
    rtc.set_loader_threads(threads)
    hs = rtc.HostScene(js, rtc.DATA_DIR + os.sep)
    return hs, t._product_tables(hs, hs.camera(0, 0))


@pytest.mark.parametrize("name", t.GOLDEN)
def test_same_description_on_one_thread_and_on_many(name):
    with open(os.path.join(rtc.SCENE_DIR, name + ".json")) as f:
        js = f.read()
    hs1, one = _tables(js, 1)
    for threads in (3, 16):
        hsn, many = _tables(js, threads)
        d1, dn = hs1.desc, hsn.desc
        for field in ("n_xforms", "n_leaves", "n_cyls", "n_tris", "n_materials", "n_patterns", "n_nodes", "n_children", "n_roots", "n_lights"):
            assert getattr(d1, field) == getattr(dn, field), field
        for key in ("kind", "shadow", "inv", "inv_t", "leaf_material", "box", "op", "count", "children", "roots", "lights", "cyl", "tri"):
            t._same_bits(one[key], many[key], f"{name} {key} ({threads} threads)")
        assert one["materials"] == many["materials"]
        ids1, idsn = one["id"].astype(np.int64), many["id"].astype(np.int64)
        assert len(ids1) == 0 or ((ids1 - ids1.min()) == (idsn - idsn.min())).all(), "Shape ids relative to the parse's first"
        # (... and the raw arrays, not re-ordered: the flattening walks the same World)
        for field, count, width in (("leaf_kind", d1.n_leaves, 1), ("leaf_xform", d1.n_leaves, 1), ("leaf_geom", d1.n_leaves, 1),
                                    ("xf_inv", d1.n_xforms, 16), ("children", d1.n_children, 1), ("node_first", d1.n_nodes, 1)):
            t._same_bits(t._arr(hs1, field, count, width), t._arr(hsn, field, count, width), field)


def test_the_first_failing_object_in_file_order_is_the_error():
    scene = {"camera": t.CAMERA, "lights": t.LIGHTS,
             "objects": [{"type": {"sphere": {}}},
                         {"type": {"from-definition": "no-such-thing"}},                          # UnknownDefinition
                         {"type": {"cube": {}}, "transform": [{"scale": [0, 1, 1]}]},            # NotInvertible
                         {"type": {"sphere": {}}}] * 3}
    for threads in (1, 4):
        rtc.set_loader_threads(threads)
        with pytest.raises(rtc.RtcError) as e:
            rtc.HostScene(json.dumps(scene), rtc.DATA_DIR + os.sep)
        assert "UnknownDefinition" in str(e.value), (threads, str(e.value))
