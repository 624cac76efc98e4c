"""Two tests of the render kernels stop before their divisions where the entries "provably cannot matter" (round 5:
a plane's one quotient, `leaf_of_kind` in trace(); the six of a cube the lights are inside of, `segment_stays_inside_cube`,
csrc/rtc_kernels.hip).  "Provably" is an argument about IEEE double arithmetic written next to the code; here the same
conditions are evaluated in numpy float64 - whose +, x, / are the hardware's - on millions of operands, most of them placed
where the argument is thinnest (quotients a few ulps either side of the limit, origins 1e-9 from a face, directions either
side of the reference's 1e-5 "parallel" rule, values that underflow), and every case in which the kernel would skip is held
to what the reference's own arithmetic (plane.zig:25-36, cube.zig:24-79) yields: an entry the visitor ignores.  CPU only."""
import numpy as np

INF = float("inf")


def _ulps(x, k):
    """x moved by k units in the last place (k an integer array)."""
    return (x.view(np.int64) + k).view(np.float64)


def _plane_cases(rng, n):
    limit = np.where(rng.random(n) < 0.1, INF, 10.0 ** rng.uniform(-4, 4, n))
    d = rng.normal(size=n) * 10.0 ** rng.uniform(-6, 2, n)
    d[rng.random(n) < 0.05] *= 1e9          # beyond the 1e10 guard now and then
    t0 = np.where(rng.random(n) < 0.5, np.where(np.isfinite(limit), limit, 1.0), 10.0 ** rng.uniform(-12, 6, n))
    t0 = _ulps(np.abs(t0), rng.integers(-6, 7, n)) * np.where(rng.random(n) < 0.5, 1.0, -1.0)
    o = -t0 * d                               # the plane is met at (about) t0 ...
    o = _ulps(o + (o == 0) * 1e-300, rng.integers(-4, 5, n))
    tiny = rng.random(n) < 0.1                # ... or so near the origin that the quotient underflows
    o[tiny] = rng.normal(size=int(tiny.sum())) * 10.0 ** rng.uniform(-320, -280, int(tiny.sum()))
    return o, d, limit


def test_a_plane_s_quotient_is_skipped_only_where_its_entry_is_ignored():
    rng = np.random.default_rng(20260105)
    skipped = unguarded = 0
    for _ in range(8):
        o, d, limit = _plane_cases(rng, 500_000)
        with np.errstate(all="ignore"):
            ao, ad = np.abs(o), np.abs(d)
            tested = ad > 1e-5                                   # plane.zig:27: otherwise no entry at all
            negative = ((o > 0.0) == (d > 0.0)) & (ao >= 1e-290) & (ad <= 1e10)
            beyond = ao > (limit * ad) * (1.0 + 1e-12)
            skip = tested & (negative | beyond)
            t = -o / d                                           # the reference's entry
        # ClosestVisitor: `et >= 0 && (et < t || (et == t && ...))`; ShadowVisitor: `et >= 0 && et < distance`
        matters = (t >= 0.0) & (t <= limit)
        wrong = skip & matters
        assert not wrong.any(), (o[wrong][:3], d[wrong][:3], limit[wrong][:3], t[wrong][:3])
        skipped += int(skip.sum())
        # (the guards are not vacuous: a quotient that underflows to -0 passes `t >= 0`, and only `ao >= 1e-290` keeps it)
        unguarded += int((tested & ((o > 0.0) == (d > 0.0)) & matters).sum())
    assert skipped > 1_000_000 and unguarded > 0


def _reference_cube(o, d):
    """cube.zig:24-79 for the unit cube, vectorised: (tmin, tmax)."""
    tmin, tmax = np.full(len(o), -INF), np.full(len(o), INF)
    with np.errstate(all="ignore"):
        for k in range(3):
            lo_n, hi_n = -1.0 - o[:, k], 1.0 - o[:, k]
            big = np.abs(d[:, k]) >= 1e-5
            safe = np.where(big, d[:, k], 1.0)
            t0 = np.where(big, lo_n / safe, lo_n * INF)
            t1 = np.where(big, hi_n / safe, hi_n * INF)
            swap = t0 > t1
            t0, t1 = np.where(swap, t1, t0), np.where(swap, t0, t1)
            tmin, tmax = np.fmax(tmin, t0), np.fmin(tmax, t1)
    return tmin, tmax


def test_a_room_s_test_is_skipped_only_where_both_entries_are_ignored():
    rng = np.random.default_rng(7)
    skipped = 0
    for _ in range(6):
        n = 400_000
        o = rng.uniform(-1.0, 1.0, (n, 3))
        near = rng.random((n, 3)) < 0.3                           # on a wall: within 1e-9 +- a little of a face, either side
        o[near] = np.sign(o[near]) * (1.0 - 10.0 ** rng.uniform(-12, -6, int(near.sum())))
        o[rng.random((n, 3)) < 0.02] *= 1.0 + 1e-9                # just outside now and then
        d = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-3, 1, (n, 1))
        small = rng.random((n, 3)) < 0.2                          # either side of the reference's parallel rule
        d[small] = rng.choice([-1.0, 1.0], int(small.sum())) * 10.0 ** rng.uniform(-5.3, -4.7, int(small.sum()))
        d[rng.random((n, 3)) < 0.02] = 0.0
        tmin, tmax = _reference_cube(o, d)
        limit = np.where(rng.random(n) < 0.6, _ulps(np.where(np.isfinite(tmax) & (tmax > 0), tmax, 1.0), rng.integers(-8, 9, n)),
                         10.0 ** rng.uniform(-3, 3, n))           # the light where the ray leaves the cube, a few ulps either side
        with np.errstate(all="ignore"):
            inside = (np.abs(o) <= 1.0 - 1e-9).all(axis=1)
            ad = np.abs(d)
            ahead = np.where(d > 0.0, 1.0 - o, 1.0 + o)
            clear = (~(ad >= 1e-5) | (ahead > (limit[:, None] * ad) * (1.0 + 1e-12))).all(axis=1)
        skip = inside & clear
        # ShadowVisitor::entry: `et >= 0 && et < distance` for either of the cube's entries, reported if !(tmin > tmax)
        reported = ~(tmin > tmax)
        shadowing = reported & (((tmin >= 0.0) & (tmin < limit)) | ((tmax >= 0.0) & (tmax < limit)))
        wrong = skip & shadowing
        assert not wrong.any(), (o[wrong][:2], d[wrong][:2], limit[wrong][:2], tmin[wrong][:2], tmax[wrong][:2])
        skipped += int(skip.sum())
    assert skipped > 200_000
