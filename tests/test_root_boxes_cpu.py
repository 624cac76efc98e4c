"""Phase 1 of the render kernels' root loop rejects a World.objects entry by an FP32 test of the ray against the entry's
world box (`roots_kept_box` in csrc/rtc_kernels.hip).  The test may only ever REMOVE work: whenever the
reference's exact arithmetic yields an entry the visitor could use, the box test must keep the entry's root.  Here - on the
CPU, no GPU - the kernel's arithmetic is replayed in float32 on the boxes rtc_scene_create builds (rtc_diag_root_boxes)
and held against a float64 restatement of Sphere / Cube.localIntersect (sphere.zig:24-46, cube.zig:24-79, with the cube's
"parallel" rule: a direction component below 1e-5 in object space is ignored), for random rays and for rays aimed at what
a box that is tight on whole faces gets wrong first: along faces, through edges and corners, nearly parallel to an axis,
from points on a face, and at cubes stretched until the parallel rule applies to ordinary rays."""
import json

import numpy as np
import pytest

F = np.float32
INF = float("inf")


def _object_rays(desc, leaf, o, d):
    """Ray.transform (ray.zig:30-32) with the leaf's inverse, in the reference's order of operations (float64)."""
    m = np.ctypeslib.as_array(desc.xf_inv, shape=(desc.n_xforms * 16,))[16 * desc.leaf_xform[leaf]:16 * desc.leaf_xform[leaf] + 16]
    oo = np.stack([((m[4 * r] * o[:, 0] + m[4 * r + 1] * o[:, 1]) + m[4 * r + 2] * o[:, 2]) + m[4 * r + 3] for r in range(3)], axis=1)
    dd = np.stack([(m[4 * r] * d[:, 0] + m[4 * r + 1] * d[:, 1]) + m[4 * r + 2] * d[:, 2] for r in range(3)], axis=1)
    return oo, dd


def _reference_entries(kind, oo, dd):
    """(has_entries, t_first, t_second) of Sphere / Cube.localIntersect for every ray (object space, float64)."""
    with np.errstate(all="ignore"):
        if kind == 0:   # sphere.zig:24-46
            a = (dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]) + dd[:, 2] * dd[:, 2]
            b = 2.0 * ((oo[:, 0] * dd[:, 0] + oo[:, 1] * dd[:, 1]) + oo[:, 2] * dd[:, 2])
            c = ((oo[:, 0] * oo[:, 0] + oo[:, 1] * oo[:, 1]) + oo[:, 2] * oo[:, 2]) - 1.0
            disc = b * b - 4.0 * a * c
            sq = np.sqrt(np.where(disc >= 0, disc, 0.0))
            return disc >= 0.0, (-b - sq) / (2.0 * a), (-b + sq) / (2.0 * a)
        tmin, tmax = np.full(len(oo), -INF), np.full(len(oo), INF)   # cube.zig:24-79
        for k in range(3):
            lo_n, hi_n = -1.0 - oo[:, k], 1.0 - oo[:, k]
            big = np.abs(dd[:, k]) >= 1e-5
            t0 = np.where(big, lo_n / np.where(big, dd[:, k], 1.0), lo_n * INF)
            t1 = np.where(big, hi_n / np.where(big, dd[:, k], 1.0), hi_n * INF)
            swap = t0 > t1
            t0, t1 = np.where(swap, t1, t0), np.where(swap, t0, t1)
            tmin, tmax = np.fmax(tmin, t0), np.fmin(tmax, t1)   # Zig @max / @min: the non-NaN operand
        return ~(tmin > tmax), tmin, tmax


def _kernel_keeps(box, scales, o, d, t_lo, t_hi):
    """ray_box + roots_kept_box of csrc/rtc_kernels.hip for one root, in float32 (an FMA: the product and the sum in float64,
    rounded once; v_rcp_f32: the correctly rounded reciprocal - each within an ulp of the hardware, a hundredth of the margin)."""
    bmax, par = F(scales[0]), F(scales[1])
    o32, d32 = o.astype(F), d.astype(F)
    d32 = np.copysign(np.maximum(np.abs(d32), F(1e-30)), d32)
    with np.errstate(all="ignore"):
        inv = (F(1.0) / d32).astype(F)
        reach = (np.max(np.abs(o32), axis=1) + bmax).astype(F)
        slack = (F(1e-6) * reach + par * (F(4.0) * reach)).astype(F)
        m = (slack[:, None] * np.abs(inv)).astype(F)
        c = (-o32 * inv).astype(F)
        cn, cf = (c - m).astype(F), (c + m).astype(F)
        lo, hi = box[0:3].astype(F), box[3:6].astype(F)
        near = np.where(d32 < 0, hi[None, :], lo[None, :])
        far = np.where(d32 < 0, lo[None, :], hi[None, :])
        tn_k = (near.astype(np.float64) * inv.astype(np.float64) + cn.astype(np.float64)).astype(F)
        tf_k = (far.astype(np.float64) * inv.astype(np.float64) + cf.astype(np.float64)).astype(F)
        tn = np.fmax(np.fmax(tn_k[:, 0], tn_k[:, 1]), tn_k[:, 2])
        tf = np.fmin(np.fmin(tf_k[:, 0], tf_k[:, 1]), tf_k[:, 2])
        culled = np.fmax(tn, F(t_lo)) > np.fmin(tf, F(t_hi))
        if box[6] != 0.0:
            culled = tn > tf
    return ~culled


def _rays(rng, desc, boxes, n):
    """Origins and (about) unit directions: random ones, and rays aimed at the finite boxes' faces, edges and corners."""
    finite = [b for b in boxes if abs(b[0]) < 1e37]
    o = rng.uniform(-12, 12, size=(n, 3))
    d = rng.normal(size=(n, 3))
    k = n // 2
    if finite:
        pick = rng.integers(0, len(finite), size=k)
        corners = np.array([[b[0 + 3 * rng.integers(0, 2)], b[1 + 3 * rng.integers(0, 2)], b[2 + 3 * rng.integers(0, 2)]] for b in
                            (finite[i] for i in pick)], dtype=np.float64)
        mids = np.array([[0.5 * (b[0] + b[3]), 0.5 * (b[1] + b[4]), 0.5 * (b[2] + b[5])] for b in (finite[i] for i in pick)])
        blend = rng.choice([0.0, 0.0, 1.0, 0.5], size=(k, 3))            # a corner, a point of an edge or a face, the middle
        target = corners * blend + mids * (1.0 - blend) + rng.choice([0.0, 1e-7, -1e-7, 1e-4, -1e-4], size=(k, 3))
        d[:k] = target - o[:k]
        on_face = rng.random(k) < 0.25                                     # a quarter of them START on the box (shadow rays do)
        o[:k][on_face] = target[on_face]
        d[:k][on_face] = rng.normal(size=(int(on_face.sum()), 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # axis-parallel and nearly parallel directions
    axis = rng.random(n) < 0.3
    comp = rng.integers(0, 3, size=n)
    tiny = rng.choice([0.0, 1e-12, 1e-7, 3e-6, 9.9e-6, 1.1e-5, 1e-4], size=n) * rng.choice([-1.0, 1.0], size=n)
    d[axis, comp[axis]] = tiny[axis]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d


def _scenes(rtc):
    import test_parity_gpu as t
    for name in ("cover.json", "cubes.json", "reflection_and_refraction.json", "fresnel.json", "skybox_demo.json"):
        yield name, rtc.HostScene.from_file(name)
    yield "grazing cubes", rtc.HostScene(t._grazing_cubes_scene("simple"))
    for seed in range(12):
        yield "random simple %d" % seed, rtc.HostScene(t._random_flat_scene(seed, True, 7))


def test_box_test_never_rejects_a_root_the_reference_needs(rtc):
    rng = np.random.default_rng(20260105)
    checked = kept_entries = 0
    for name, hs in _scenes(rtc):
        desc = hs.desc
        boxes, order, scales = rtc.root_boxes(desc)
        o, d = _rays(rng, desc, boxes, 6000)
        distance = rng.uniform(0.3, 40.0, size=len(o))
        limits = {"closest": (-1.0002e-4, INF),
                  "behind": (-INF, 1.0002e-4)}
        for pos in range(desc.n_roots):
            root = desc.roots[int(order[pos])]
            if root & 0x80000000:
                continue
            kind = desc.leaf_kind[root]
            if kind not in (0, 2):     # spheres and cubes (a plane has no bound; the other kinds' entries lie on their surfaces)
                continue
            has, t1, t2 = _reference_entries(kind, *_object_rays(desc, root, o, d))
            need = {"closest": has & ((t1 >= 0) | (t2 >= 0)),
                    "behind": has & ((t1 < 0) | (t2 < 0)),
                    "shadow": has & (((t1 >= 0) & (t1 < distance)) | ((t2 >= 0) & (t2 < distance)))}
            for visitor, needed in need.items():
                if visitor == "shadow":   # far_limit(): (distance x 1.0001 + 1.0001e-4) x 1.0002, in float32, a limit per ray
                    hi = ((distance.astype(F) * F(1.0001) + F(1.0001e-4)) * F(1.0002)).astype(F)
                    keeps = np.array([_kernel_keeps(boxes[pos], scales, o[i:i + 1], d[i:i + 1], -1.0002e-4, hi[i])[0] for i in np.flatnonzero(needed)])
                    bad = np.flatnonzero(needed)[~keeps] if len(keeps) else []
                else:
                    keeps = _kernel_keeps(boxes[pos], scales, o, d, *limits[visitor])
                    bad = np.flatnonzero(needed & ~keeps)
                assert len(bad) == 0, (name, "table position", pos, visitor, "ray", o[bad[0]].tolist(), d[bad[0]].tolist(),
                                       "entries", float(t1[bad[0]]), float(t2[bad[0]]), "box", boxes[pos].tolist())
                checked += int(needed.sum())
            kept_entries += 1
    assert kept_entries >= 60 and checked > 50000   # (the test did look at something)


def test_box_test_rejects_what_it_can(rtc):
    """... and it does remove work: on cover.json most (ray, cube) pairs of rays that miss a cube are rejected."""
    rng = np.random.default_rng(7)
    hs = rtc.HostScene.from_file("cover.json")
    desc = hs.desc
    boxes, order, scales = rtc.root_boxes(desc)
    o = np.tile(np.array([[-6.0, 6.0, -10.0]]), (4000, 1))
    d = rng.normal(size=(4000, 3)) * 0.25 + np.array([12.0, -6.0, 16.0]) / 20.88
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    missed = rejected = 0
    for pos in range(desc.n_roots):
        root = desc.roots[int(order[pos])]
        if desc.leaf_kind[root] != 2:
            continue
        has, t1, t2 = _reference_entries(2, *_object_rays(desc, root, o, d))
        keeps = _kernel_keeps(boxes[pos], scales, o, d, -1.0002e-4, INF)
        missed += int((~has).sum())
        rejected += int((~has & ~keeps).sum())
    assert missed > 10000 and rejected > 0.97 * missed, (missed, rejected)
