"""CPU: nep_inflate_static (Neptune::setStaticObst, neptune.cpp:639-664) through the C ABI against the oracle's restatement
(orc_inflate_static) bit for bit, and against Qhull (scipy.spatial.ConvexHull: same extreme points, same cyclic order)."""
import numpy as np
import pytest

from neptune_amd import abi, scene
from neptune_amd._lib import lib


def _footprints(rng, n):
    out = []
    for k in range(n):
        kind = k % 5
        c = rng.uniform(-20, 20, 2)
        if kind == 0:                                   # the reference's 0.5 m squares (neptune_ros.cpp:212-250)
            v = c + 0.25 * np.array([[1, 1], [1, -1], [-1, -1], [-1, 1.0]])
        elif kind == 1:                                 # a random convex polygon
            ang = np.sort(rng.uniform(0, 2 * np.pi, int(rng.integers(3, 9))))
            v = c + np.stack([np.cos(ang), np.sin(ang)], 1) * rng.uniform(0.2, 3.0)
        elif kind == 2:                                 # an unordered point cloud (the hull does not care)
            v = c + rng.normal(size=(int(rng.integers(3, 12)), 2))
        elif kind == 3:                                 # collinear points, one of them repeated
            t = rng.integers(-8, 9, 4) * 0.25           # (exactly representable: collinear in exact arithmetic too, as CGAL's predicates see it)
            v = np.round(c) + np.stack([t, 2 * t], 1); v[3] = v[0]
        else:                                           # a single point
            v = c[None, :]
        out.append(np.ascontiguousarray(v))
    return out


def test_inflate_static_equals_the_oracle_and_qhull(oracle):
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(5)
    fps = _footprints(rng, 60)
    for r in (0.6, 0.4):          # the reference's drone_radius; and one whose safe_dist (1.0) keeps the collinear footprints exactly collinear
        got = scene.inflate_statics(fps, r)
        sd = 2 * r + 0.2
        for v, g in zip(fps, got):
            want = oracle.inflate_static(v, sd)
            assert g.shape == want.shape and g.tobytes() == want.tobytes()
            assert tuple(g[0]) == min(map(tuple, g))                                 # starts at the lexicographically smallest point
            if r == 0.6:
                continue
            pts = np.concatenate([v + sd * np.array(s) for s in ((1, 1), (1, -1), (-1, -1), (-1, 1))])
            q = pts[ConvexHull(pts).vertices]                                        # counter-clockwise
            k = min(range(len(q)), key=lambda i: tuple(q[i]))
            q = np.roll(q, -k, axis=0)
            assert len(q) == len(g) and np.array_equal(q, g)


def test_inflate_static_argument_checks():
    L = lib()
    off = np.array([0, 20], dtype=np.int32)
    ang = np.linspace(0, 2 * np.pi, 20, endpoint=False)
    xy = np.ascontiguousarray(np.stack([np.cos(ang), np.sin(ang)], 1) * 5.0)          # 20 extreme points: more than NEP_HULL_MAX_V
    out_off = np.zeros(2, dtype=np.int32); out = np.zeros((80, 2))
    assert L.nep_inflate_static(1, abi.iptr(off), abi.dptr(xy), 0.6, abi.iptr(out_off), abi.dptr(out), 80) == -4     # NEP_E_CAP
    off4 = np.array([0, 4], dtype=np.int32)
    assert L.nep_inflate_static(1, abi.iptr(off4), abi.dptr(xy), 0.6, abi.iptr(out_off), abi.dptr(out), 2) == -4     # capacity
    assert L.nep_inflate_static(1, abi.iptr(off4), abi.dptr(xy), 0.6, None, abi.dptr(out), 80) == -1                 # NEP_E_ARG
    assert L.nep_inflate_static(0, None, None, 0.6, abi.iptr(out_off), None, 0) == 0
