"""Pins the CPU oracle — and the product host library's build-time helpers — to the reference's
own known-answer tests.  Each KAT line of the two C++ KAT binaries becomes one case."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_KAT = os.path.join(REPO, "oracle", "build", "oracle_kat")
HOST_KAT = os.path.join(REPO, "ray-tracer-challenge_amd", "lib", "rtc_host_kat")


def _run(binary):
    if not os.path.exists(binary):
        subprocess.run(["make", "-C", REPO, "all"], check=True, stdout=subprocess.DEVNULL)
    p = subprocess.run([binary], capture_output=True, text=True)
    lines = [l.split(" ", 4) for l in p.stdout.splitlines() if l.startswith("KAT ")]
    summary = [l for l in p.stdout.splitlines() if l.startswith("KAT-SUMMARY")]
    return p.returncode, lines, summary


def test_oracle_kats_all_pass():
    rc, lines, summary = _run(ORACLE_KAT)
    failed = [l for l in lines if l[3] != "PASS"]
    assert not failed, failed
    assert rc == 0 and summary and "failed=0" in summary[0]
    # every area of the SURVEY §4 table is represented
    areas = {l[1].split(":")[0] for l in lines}
    for area in ["tuple.zig", "matrix.zig", "ray.zig", "sphere.zig", "plane.zig", "cube.zig", "cylinder.zig", "cone.zig", "blend.zig", "csg.zig",
                 "triangle.zig", "group.zig", "bounding_box.zig", "shape.zig", "material.zig", "pattern.zig",
                 "checkers.zig", "stripes.zig", "world.zig", "camera.zig"]:
        assert area in areas, area
    assert len(lines) >= 490
    # the oracle's OWN scene build (rtc_oracle_scene.hpp) is held to the reference's vectors too, not only to the product
    # loader: partition / makeSubgroup / divide x 2 (group.zig:246-381), the four splits (bounding_box.zig:365-423), the
    # OBJ parser's cases (obj.zig:288-544), parseScene (scene.zig:664-774)
    scene = {(l[1], l[2]) for l in lines if l[2].startswith("scene_")}
    for case in [("group.zig:270", "scene_partition"), ("group.zig:291", "scene_make_subgroup"), ("group.zig:313", "scene_divide_1"),
                 ("group.zig:363", "scene_divide_3"), ("bounding_box.zig:365", "scene_split_cube"),
                 ("bounding_box.zig:380", "scene_split_x_wide"), ("bounding_box.zig:395", "scene_split_y_wide"),
                 ("bounding_box.zig:410", "scene_split_z_wide"), ("obj.zig:306", "scene_ignored_lines"), ("obj.zig:326", "scene_vertices"),
                 ("obj.zig:361", "scene_faces"), ("obj.zig:395", "scene_fan_triangulation"), ("obj.zig:434", "scene_named_groups"),
                 ("obj.zig:471", "scene_obj_to_group"), ("obj.zig:498", "scene_normals"), ("obj.zig:532", "scene_faces_with_normals"),
                 ("scene.zig:724", "scene_camera"), ("scene.zig:740", "scene_pattern_transform"), ("scene.zig:770", "scene_light")]:
        assert case in scene, case


def test_host_kats_all_pass():
    rc, lines, summary = _run(HOST_KAT)
    failed = [l for l in lines if l[3] != "PASS"]
    assert not failed, failed
    assert rc == 0 and summary and "failed=0" in summary[0]
    areas = {l[1].split(":")[0] for l in lines}
    for area in ["matrix.zig", "bounding_box.zig", "group.zig", "scene.zig", "obj.zig", "canvas.zig", "camera.zig", "cone.zig"]:
        assert area in areas, area


@pytest.mark.parametrize("where,name", [
    ("camera.zig:186", "render_center_pixel"),   # end-to-end default world, (0.38066, 0.47583, 0.2855)
    ("world.zig:890", "shade_schlick"),          # reflection + refraction + schlick
    ("shape.zig:551", "n1_n2_3"),                # containers walk
])
def test_headline_kats_present(where, name):
    _, lines, _ = _run(ORACLE_KAT)
    assert any(l[1] == where and l[2] == name and l[3] == "PASS" for l in lines)
