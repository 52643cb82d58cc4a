"""Experiment (round 6): the interval hull as four quadrant arcs of the four translated copies of the control points (each copy's
own rounded coordinates, monotone chains of <= 16 points) against the generic monotone chain over all 64 inflated points
(oracle.convex_hull_2d) — on random, straight-line, stationary and axis-aligned trajectories.  CPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle
from neptune_amd.scene import A_POS_INV


def cross3(o, a, b):
    return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])


def chain(pts):
    """Andrew's one-directional chain over pts in the given order (already sorted, unique): left turns kept"""
    h = []
    for p in pts:
        while len(h) >= 2 and cross3(h[-2], h[-1], p) <= 0.0:
            h.pop()
        h.append(p)
    return h


def uniq_sorted(pts):
    pts = sorted(pts)
    out = []
    for p in pts:
        if not out or p != out[-1]:
            out.append(p)
    return out


def hull_planb(cps, dx, dy):
    xm = [x - dx for x, _ in cps]; xp = [x + dx for x, _ in cps]
    ym = [y - dy for _, y in cps]; yp = [y + dy for _, y in cps]
    mm = uniq_sorted(list(zip(xm, ym))); pm = uniq_sorted(list(zip(xp, ym)))
    pp = uniq_sorted(list(zip(xp, yp))); mp = uniq_sorted(list(zip(xm, yp)))
    out = []
    # arc 1: lower chain of mm from its start to the first vertex attaining min y
    c = chain(mm); ymin = min(p[1] for p in c)
    i1 = next(i for i, p in enumerate(c) if p[1] == ymin)
    out += c[:i1 + 1]
    # arc 2: lower chain of pm from the last vertex attaining min y to the first attaining max x
    c = chain(pm); ymin = min(p[1] for p in c); xmax = max(p[0] for p in c)
    i0 = max(i for i, p in enumerate(c) if p[1] == ymin); i1 = next(i for i, p in enumerate(c) if p[0] == xmax)
    out += c[i0:i1 + 1]
    # arc 3: upper chain of pp (descending order) from its start to the first vertex attaining max y
    c = chain(pp[::-1]); ymax = max(p[1] for p in c)
    i1 = next(i for i, p in enumerate(c) if p[1] == ymax)
    out += c[:i1 + 1]
    # arc 4: upper chain of mp from the last vertex attaining max y to the first attaining min x
    c = chain(mp[::-1]); ymax = max(p[1] for p in c); xmin = min(p[0] for p in c)
    i0 = max(i for i, p in enumerate(c) if p[1] == ymax); i1 = next(i for i, p in enumerate(c) if p[0] == xmin)
    out += c[i0:i1 + 1]
    return np.array(out)


def control_points(P, _t):
    c = [_t * _t * _t, _t * _t, _t, 1.0]
    out = []
    for k in range(4):
        v = []
        for ax in range(2):
            p = P[ax]
            v.append((((p[0] * c[0]) * A_POS_INV[0][k] + (p[1] * c[1]) * A_POS_INV[1][k]) + (p[2] * c[2]) * A_POS_INV[2][k]) + (p[3] * c[3]) * A_POS_INV[3][k])
        out.append((float(v[0]), float(v[1])))
    return out


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    rng = np.random.default_rng(1)
    dx = dy = 1.2
    bad = 0; nv_hist = {}
    kinds = ["random", "line", "rest", "axis", "line_axis", "mixed"]
    per_kind = {k: [0, 0] for k in kinds}
    for it in range(n_cases):
        kind = kinds[it % len(kinds)]
        nseg = int(rng.integers(1, 5))
        cps = []
        pos = rng.uniform(-40, 40, 2)
        if it % 7 == 0:
            pos = np.array([rng.choice([1, 2, 4, 8, 16, 32]) * rng.choice([-1, 1]) + rng.uniform(-1.3, 1.3), rng.choice([1, 2, 4, 8, 16, 32]) + rng.uniform(-1.3, 1.3)])   # around a binade boundary
        vel = rng.uniform(-2, 2, 2)
        for s in range(nseg):
            k = kind if kind != "mixed" else kinds[int(rng.integers(0, 5))]
            if k == "random":
                P = np.stack([np.array([rng.normal() * 0.05, rng.normal() * 0.1, rng.normal() * 0.5, pos[a]]) for a in range(2)])
            elif k == "line":
                P = np.stack([np.array([0.0, 0.0, vel[a], pos[a]]) for a in range(2)])
            elif k == "rest":
                P = np.stack([np.array([0.0, 0.0, 0.0, pos[a]]) for a in range(2)])
            elif k == "axis":
                P = np.stack([np.array([rng.normal() * 0.05, rng.normal() * 0.1, rng.normal() * 0.5, pos[0]]), np.array([0.0, 0.0, 0.0, pos[1]])])
            else:
                P = np.stack([np.array([0.0, 0.0, vel[0], pos[0]]), np.array([0.0, 0.0, 0.0, pos[1]])])
            _t = [0.5, 0.5, 0.0, float(rng.uniform(0, 0.5)), 0.37][int(rng.integers(0, 5))]
            cps += control_points(P, _t)
            pos = np.array([((P[a][0] * 0.125 + P[a][1] * 0.25) + P[a][2] * 0.5) + P[a][3] for a in range(2)])
        pts = []
        for x, y in cps:
            pts += [(x + dx, y + dy), (x + dx, y - dy), (x - dx, y - dy), (x - dx, y + dy)]
        ref = oracle.convex_hull_2d(np.array(pts))
        got = hull_planb(cps, dx, dy)
        per_kind[kind][0] += 1
        nv_hist[len(ref)] = nv_hist.get(len(ref), 0) + 1
        if got.shape != ref.shape or got.tobytes() != ref.tobytes():
            bad += 1; per_kind[kind][1] += 1
            if bad <= 5:
                print("MISMATCH", kind, nseg, "ref", len(ref), "got", len(got)); print(ref); print(got)
    print("cases", n_cases, "mismatches", bad, per_kind, "vertex counts", dict(sorted(nv_hist.items())))


if __name__ == "__main__":
    main()
