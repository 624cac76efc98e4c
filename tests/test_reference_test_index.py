"""Every `test "..."` block of the reference (140, indexed in tests/golden/reference_tests.json by
tests/golden/make_reference_test_index.py: file, first line, last line, name - no source text) is held to at least one
known-answer vector that cites a line INSIDE the block, and no vector cites a line that is in no block unless it cites the
implementation above the file's first test.  Round 5's audit found 31 blocks without such a vector: 20 were vectors citing a
line or two off (two of them inside a neighbouring test), 11 were tests that had been passed over as trivial (constructors,
three-line bounds(), the P3 reader)."""
import collections
import json
import os
import re

import test_oracle_kats as kats_module

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INDEX = json.load(open(os.path.join(REPO, "tests", "golden", "reference_tests.json")))


def _citations():
    out = []
    for binary, tag in ((kats_module.ORACLE_KAT, "oracle"), (kats_module.HOST_KAT, "host")):
        rc, lines, _ = kats_module._run(binary)
        assert rc == 0
        for l in lines:
            m = re.match(r"([\w\.]+\.zig):(\d+)", l[1])
            if m:
                out.append((m.group(1), int(m.group(2)), l[2], l[3], tag))
    return out


def test_index_is_the_reference_s_140_tests():
    assert len(INDEX) == 140
    per_file = collections.Counter(r["file"] for r in INDEX)
    assert per_file["world.zig"] == 10 and per_file["texture_map.zig"] == 14 and per_file["triangle.zig"] == 13 and per_file["canvas.zig"] == 7
    for r in INDEX:
        assert r["first_line"] < r["last_line"] and r["name"]


def test_every_reference_test_has_a_vector_inside_it():
    cites = _citations()
    missing = [r for r in INDEX
               if not any(c[0] == r["file"] and r["first_line"] <= c[1] <= r["last_line"] and c[3] == "PASS" for c in cites)]
    assert not missing, missing
    # the path's own tests (world.zig, camera.zig, shape.zig, the shapes) are held by the ORACLE's vectors, not only by the host library's
    for r in INDEX:
        if r["file"] in ("world.zig", "camera.zig", "sphere.zig", "plane.zig", "cube.zig", "cylinder.zig", "cone.zig", "csg.zig", "material.zig", "ray.zig"):
            assert any(c[0] == r["file"] and r["first_line"] <= c[1] <= r["last_line"] and c[4] == "oracle" for c in cites), r


def test_no_vector_cites_a_line_between_tests():
    first_test = {}
    for r in INDEX:
        first_test[r["file"]] = min(first_test.get(r["file"], 1 << 30), r["first_line"])
    stray = [c for c in _citations() if c[0] in first_test and c[1] >= first_test[c[0]]
             and not any(c[0] == r["file"] and r["first_line"] <= c[1] <= r["last_line"] for r in INDEX)]
    assert not stray, stray
