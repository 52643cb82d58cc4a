"""Synthetic N-agent / M-obstacle scenes for the back-end path (host side, numpy only).

There is no ROS and no front end in this build; this module plays the role of the reference's
experiment drivers for the one path we accelerate:
  * bases on a circle               — reference neptune/src/neptune_ros.cpp:138-151
  * random 0.5 m square obstacles   — reference neptune/src/neptune_ros.cpp:212-250
  * inflation of static obstacles   — reference neptune/src/neptune.cpp:639-664 (host, once)
  * random goals                    — reference neptune/scripts/benchmark_mtlp.py:225-242
  * initial guesses                 — jerk-lattice roll-outs with the primitive structure of
                                      reference neptune/src/kinodynamic_search.cpp:1053-1097
                                      (piecewise-constant jerk on {-5,-2.5,0,2.5,5}, T_span per step)
Parameter values are those of reference neptune/param/neptune_mtlp_benchmark.yaml; the world is
scaled with sqrt(N/5) to keep the shipped scene's agent density.
"""
import math
from dataclasses import dataclass, field

import numpy as np

from . import abi

# MINVO position basis inverse on [0,1] (mader_types.hpp:152-157 inverted exactly; same literals
# as csrc/nep_tables.h).  Host copy used only for scene rejection tests and tests' helpers.
A_POS_INV = np.array([
    [-0.03203276669713047, -0.09273093424558249, 0.3420572455666699, 1.1023313949144335],
    [-0.05111494245568798, -0.046272612998418894, 0.5458234872124772, 1.0979806946005568],
    [-0.07454781852812224, 0.203951949894552, 0.796048050105448, 1.0745478185281223],
    [1.0, 1.0, 0.9999999999999996, 0.9999999999999993]])


@dataclass
class Params:
    num_agents: int = 5
    n_static: int = 0
    num_pol: int = 8
    T_span: float = 0.5
    dc: float = 0.05
    weight: float = 1000.0
    v_max: float = 2.0
    a_max: float = 3.0
    j_max: float = 5.0
    drone_radius: float = 0.6
    z_min: float = -0.2
    z_max: float = 5.1
    x_min: float = -12.0
    x_max: float = 12.0
    y_min: float = -12.0
    y_max: float = 12.0
    tether_length: float = 40.0
    goal_height: float = 1.0
    enable_entangle: bool = False
    pb: np.ndarray = field(default_factory=lambda: np.zeros((0, 2)))

    @property
    def max_states(self):
        return int(math.ceil(self.num_pol * self.T_span / self.dc)) + 3


def scaled_params(num_agents, n_static, **kw):
    """World half-width 12*sqrt(N/5) m, base circle radius 10*sqrt(N/5)."""
    sc = math.sqrt(max(num_agents, 1) / 5.0)
    if num_agents <= 5:
        sc = 1.0
    p = Params(num_agents=num_agents, n_static=n_static, x_min=-12.0 * sc, x_max=12.0 * sc,
               y_min=-12.0 * sc, y_max=12.0 * sc, **kw)
    p.tether_length = 40.0 * sc
    p.pb, _ = bases_on_circle(num_agents, 10.0 * sc)
    return p


def bases_on_circle(n, radius=10.0, dist_to_agent=2.5):
    """neptune_ros.cpp:138-151 (cosf/sinf are single precision there; kept)."""
    one_slice = 3.1415927 * 2 / n
    pb = np.zeros((n, 2)); start = np.zeros((n, 2))
    for i in range(n):
        th = one_slice * i
        c = float(np.cos(np.float32(th))); s = float(np.sin(np.float32(th)))
        c2 = float(np.cos(np.float32(1.571 - th))); s2 = float(np.sin(np.float32(1.571 - th)))
        pb[i] = (-radius * c - dist_to_agent * c2, -radius * s + dist_to_agent * s2)
        start[i] = (-radius * c, -radius * s)
    return pb, start


def hull_ccw_lexmin(pts):
    """Andrew monotone chain, CCW from the lexicographically smallest point, collinear dropped
    (host restatement of cu::convexHullOfPoints2d, cgal_utils.cpp:157-174, for setup-time use)."""
    P = sorted(set(map(tuple, np.asarray(pts, dtype=np.float64).reshape(-1, 2))))
    if len(P) <= 1:
        return np.array(P, dtype=np.float64).reshape(-1, 2)

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])
    lo = []
    for p in P:
        while len(lo) >= 2 and cross(lo[-2], lo[-1], p) <= 0:
            lo.pop()
        lo.append(p)
    up = []
    for p in reversed(P):
        while len(up) >= 2 and cross(up[-2], up[-1], p) <= 0:
            up.pop()
        up.append(p)
    return np.array(lo[:-1] + up[:-1], dtype=np.float64)


def inflate_static(verts, drone_radius):
    """Neptune::setStaticObst, neptune.cpp:639-664: every vertex +-(2*drone_radius+0.2), hull — through the C ABI
    (nep_inflate_static, host-only set-up step of the library)."""
    return inflate_statics([verts], drone_radius)[0]


def inflate_statics(footprints, drone_radius):
    """nep_inflate_static for a list of (n,2) footprints -> list of inflated counter-clockwise polygons."""
    import ctypes as C
    from ._lib import check, lib
    fp = [np.ascontiguousarray(v, dtype=np.float64).reshape(-1, 2) for v in footprints]
    off = np.zeros(len(fp) + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(v) for v in fp])
    xy = np.ascontiguousarray(np.concatenate(fp)) if len(fp) and off[-1] else np.zeros((0, 2))
    out_off = np.zeros(len(fp) + 1, dtype=np.int32); cap = max(4 * int(off[-1]), 1); out = np.zeros((cap, 2))
    check(lib().nep_inflate_static(len(fp), abi.iptr(off), abi.dptr(xy), float(drone_radius), abi.iptr(out_off), abi.dptr(out), cap))
    return [out[out_off[j]:out_off[j + 1]].copy() for j in range(len(fp))]


def random_static_obstacles(p, rng, voxel=0.2):
    """neptune_ros.cpp:212-250 with the base-distance test applied to every base."""
    pts = []
    tries = 0
    while len(pts) < p.n_static and tries < 100000:
        tries += 1
        x = rng.uniform(p.x_min, p.x_max); y = rng.uniform(p.y_min, p.y_max)
        ok = True
        for q in pts:
            d = math.hypot(x - q[0], y - q[1])
            if d < 3 * voxel or (d > 2 * p.drone_radius and d < voxel * 2.83 + 8 * p.drone_radius):
                ok = False
                break
        if ok and len(p.pb) and np.min(np.hypot(p.pb[:, 0] - x, p.pb[:, 1] - y)) < 6.0:
            ok = False
        if ok:
            pts.append((x, y))
    raw = [np.array([[x + .25, y + .25], [x + .25, y - .25], [x - .25, y - .25], [x - .25, y + .25]])
           for x, y in pts]
    return raw, inflate_statics(raw, p.drone_radius)


LATTICE = np.array([-5.0, -2.5, 0.0, 2.5, 5.0])


def _rollout_axis(p0, v0, a0, goal, T, K, v_max, a_max):
    """Greedy per-axis jerk-lattice roll-out: segment coefficients [a b c d] with a=j/6, b=a0/2,
    c=v0, d=p0 (kinodynamic_search.cpp:1096-1097)."""
    co = np.zeros((K, 4))
    for i in range(K):
        best, bj = None, 0.0
        remaining = K - i
        for j in LATTICE:
            a1 = a0 + j * T
            v1 = v0 + a0 * T + 0.5 * j * T * T
            pp1 = p0 + v0 * T + 0.5 * a0 * T * T + j * T ** 3 / 6
            # desired speed: brake so that the goal is reached at rest
            dist = goal - pp1
            v_des = np.clip(1.2 * dist, -0.8 * v_max, 0.8 * v_max)
            if remaining <= 3:
                v_des *= (remaining - 1) / 3.0
            cost = (v1 - v_des) ** 2 + 0.15 * a1 * a1
            if abs(a1) > 0.85 * a_max:
                cost += 100 * (abs(a1) - 0.85 * a_max) ** 2 + 10
            if abs(v1) > 0.9 * v_max:
                cost += 100 * (abs(v1) - 0.9 * v_max) ** 2 + 10
            if best is None or cost < best:
                best, bj = cost, j
        co[i] = (bj / 6.0, a0 / 2.0, v0, p0)
        p0, v0, a0 = (p0 + v0 * T + 0.5 * a0 * T * T + bj * T ** 3 / 6,
                      v0 + a0 * T + 0.5 * bj * T * T, a0 + bj * T)
    return co


def rollout(p0, v0, a0, goal, par, K):
    co = np.zeros((3, K, 4))
    for ax in range(2):
        co[ax] = _rollout_axis(p0[ax], v0[ax], a0[ax], goal[ax], par.T_span, K, par.v_max, par.a_max)
    co[2, :, 3] = p0[2]  # constant height (the reference feeds a z B-spline profile, neptune.cpp:1427)
    return co


def pos_ctrl_pts(coeff_axis, T):
    """[K][4] coefficients -> [K][4] MINVO position control points (solver_gurobi_poly.cpp:240)."""
    M = A_POS_INV * np.array([T ** 3, T ** 2, T, 1.0])[:, None]
    return coeff_axis @ M


def _aabb_sep(lo1, hi1, lo2, hi2):
    return (hi1[0] < lo2[0]) or (hi2[0] < lo1[0]) or (hi1[1] < lo2[1]) or (hi2[1] < lo1[1])


# MINVO velocity basis inverse (mader_types.hpp:159-163 inverted; same literals as csrc/nep_tables.h)
A_VEL_INV = np.array([
    [-0.07735026918962577, 0.16666666666666635, 1.077350269189625],
    [-0.07735026918962577, 0.49999999999999967, 1.077350269189625],
    [1.0000000000000002, 1.0000000000000009, 1.0000000000000016]])


def separable(A, B, margin=1e-6):
    """Is there a line with the convex hulls of point sets A and B at least `margin` apart on either side
    (= the separator LP of separator_glpk.cpp:248-373 is feasible)?  Separating-axis test: for two convex
    polygons one of the edge normals of either separates them if anything does; degenerate hulls (a hovering
    agent's coincident control points, collinear points) add the axes through pairs of points."""
    A = np.asarray(A, dtype=np.float64).reshape(-1, 2); B = np.asarray(B, dtype=np.float64).reshape(-1, 2)
    HA = hull_ccw_lexmin(A); HB = hull_ccw_lexmin(B)
    axes = []
    for H in (HA, HB):
        if len(H) >= 2:
            e = np.roll(H, -1, 0) - H
            axes.append(np.stack([-e[:, 1], e[:, 0]], 1))
    if len(HA) < 3 or len(HB) < 3:
        d = (HA[:, None, :] - HB[None, :, :]).reshape(-1, 2)
        axes.append(d); axes.append(np.stack([-d[:, 1], d[:, 0]], 1))
    ax = np.concatenate(axes)
    nrm = np.hypot(ax[:, 0], ax[:, 1])
    ax = ax[nrm > 0] / nrm[nrm > 0, None]
    if not len(ax):
        return False
    pa = ax @ HA.T; pb = ax @ HB.T
    gap = np.maximum(pa.min(1) - pb.max(1), pb.min(1) - pa.max(1))
    return bool(gap.max() > margin)


def interval_ctrl_pts(times, coeff_xy, t0, t1, T_span):
    """MINVO control points of the committed segments overlapping [t0, t1] (Neptune::vertexesOfInterval2d,
    neptune.cpp:379-429): host restatement for scene set-up (which guesses are mutually separable)."""
    n = len(times) - 1
    first = int(np.searchsorted(times, t0, side="left")) - 1
    last = int(np.searchsorted(times, t1, side="right")) - 1
    first = min(max(first, 0), n - 1); last = min(max(last, 0), n - 1)
    pts = []
    for s in range(first, last + 1):
        if s != last:
            _t = times[s + 1] - times[s]
        elif t1 > times[s + 1]:
            _t = times[s + 1] - times[s]
        else:
            _t = t1 - times[s]
        _t = min(max(_t, 0.0), T_span)
        M = A_POS_INV * np.array([_t ** 3, _t ** 2, _t, 1.0])[:, None]
        pts.append(np.stack([coeff_xy[0, s] @ M, coeff_xy[1, s] @ M], 1))
    return np.concatenate(pts)


def interval_hull(times, coeff_xy, t0, t1, T_span, delta):
    """inflated interval hull (neptune.cpp:436-446 + cu::convexHullOfPoints2d), counter-clockwise"""
    q = interval_ctrl_pts(times, coeff_xy, t0, t1, T_span)
    c = np.array([[delta, delta], [delta, -delta], [-delta, -delta], [-delta, delta]])
    return hull_ccw_lexmin((q[:, None, :] + c[None, :, :]).reshape(-1, 2))


def make_scene(num_agents, n_static, seed, K=8, warm_fraction=0.5, par=None, t_jitter=0.0, separation="hull"):
    """Returns dict(par, statics_raw, statics, starts, goals, guesses [N] (GUESS_DTYPE),
    committed [N] (TRAJ_REC_DTYPE)).

    separation: which guesses are accepted ("rejecting guesses ... whose hulls are not separable
    from every obstacle hull, so every LP is feasible, as after the real front end"):
      "hull"  every separator LP the back end will pose in round 0 is feasible — each segment's control polygon
              against every inflated static and against the interval hull of every other agent's guess, both ways
              (separating-axis test on the actual hulls);
      "aabb"  the round-1 generator: bounding boxes of the control polygons against inflated, windowed boxes —
              sufficient for the above but much stricter (agents end up farther apart; kept because the committed
              golden QP cases of tests/golden/qp_cases.npz were drawn from it)."""
    if separation not in ("hull", "aabb"):
        raise ValueError("separation must be 'hull' or 'aabb'")
    rng = np.random.default_rng(seed)
    p = par if par is not None else scaled_params(num_agents, n_static)
    N = p.num_agents
    sc = (p.x_max - p.x_min) / 24.0
    _, starts = bases_on_circle(N, 10.0 * sc if N > 5 else 10.0)
    raw, statics = random_static_obstacles(p, rng)
    st_lo = [s.min(0) for s in statics]; st_hi = [s.max(0) for s in statics]
    T = p.T_span
    infl = 2 * p.drone_radius  # bbox/2 + drone_radius with bbox = 2*drone_radius (neptune_ros.cpp:444-446)
    guesses = np.zeros(N, dtype=abi.GUESS_DTYPE)
    committed = np.zeros(N, dtype=abi.TRAJ_REC_DTYPE)
    goals = np.zeros((N, 3))
    prev_lo = np.zeros((N, K, 2)); prev_hi = np.zeros((N, K, 2)); prev_n = 0   # windowed control-point boxes of accepted agents
    st_lo_a = np.array(st_lo).reshape(-1, 2); st_hi_a = np.array(st_hi).reshape(-1, 2)
    close_range = 4.0
    prev_hulls, prev_cps = [], []    # separation == "hull": interval hulls / control polygons of the accepted guesses
    t_scene = 0.0                    # (hulls are taken on the common grid; with t_jitter the start times differ by < T and the test is approximate)
    for i in range(N):
        accepted = None
        for attempt in range(300):
            g = np.array([rng.uniform(p.x_min + 4.0, p.x_max - 4.0),
                          rng.uniform(p.y_min + 4.0, p.y_max - 4.0), p.goal_height])
            if np.hypot(*(g[:2] - p.pb[i])) > p.tether_length:
                continue
            if i and np.min(np.hypot(goals[:i, 0] - g[0], goals[:i, 1] - g[1])) < close_range:
                continue
            p0 = np.array([starts[i][0], starts[i][1], p.goal_height]); v0 = np.zeros(3); a0 = np.zeros(3)
            co = rollout(p0, v0, a0, g, p, K)
            if K >= 2 and rng.uniform() < warm_fraction:  # mid-flight replan: restart from a later knot
                s = int(rng.integers(1, min(4, K)))
                c = co[:, s - 1, :]
                p0 = c[:, 0] * T ** 3 + c[:, 1] * T ** 2 + c[:, 2] * T + c[:, 3]
                v0 = 3 * c[:, 0] * T ** 2 + 2 * c[:, 1] * T + c[:, 2]
                a0 = 6 * c[:, 0] * T + 2 * c[:, 1]
                co = rollout(p0, v0, a0, g, p, K)
            cx = pos_ctrl_pts(co[0], T); cy = pos_ctrl_pts(co[1], T)
            lo = np.stack([cx.min(1), cy.min(1)], 1); hi = np.stack([cx.max(1), cy.max(1)], 1)
            if lo[:, 0].min() < p.x_min + 0.2 or hi[:, 0].max() > p.x_max - 0.2 or \
               lo[:, 1].min() < p.y_min + 0.2 or hi[:, 1].max() > p.y_max - 0.2:
                continue
            # windowed boxes of this guess: interval k of a trajectory covers its segments k-1..k+1
            # (neptune.cpp:379-389), for it as an obstacle and for the conservative test below
            wlo = np.stack([lo[max(k - 1, 0):min(k + 1, K - 1) + 1].min(0) for k in range(K)])
            whi = np.stack([hi[max(k - 1, 0):min(k + 1, K - 1) + 1].max(0) for k in range(K)])
            my_hulls = None
            if separation == "hull":   # base squares (solver_gurobi_poly.cpp:521-553: 0.7 m half-width): LPs against them must be feasible too
                br = 0.7
                sepb = ((hi[:, None, 0] < p.pb[None, :, 0] - br - 0.05) | (p.pb[None, :, 0] + br + 0.05 < lo[:, None, 0]) |
                        (hi[:, None, 1] < p.pb[None, :, 1] - br - 0.05) | (p.pb[None, :, 1] + br + 0.05 < lo[:, None, 1]))
                ok = True
                for k, j in zip(*np.nonzero(~sepb)):
                    sq = p.pb[j] + br * np.array([[1.0, 1.0], [1.0, -1.0], [-1.0, -1.0], [-1.0, 1.0]])
                    if not separable(sq, np.stack([cx[k], cy[k]], 1), 1e-3):
                        ok = False
                        break
                if not ok:
                    continue
            if len(st_lo_a):   # static obstacles: [S][2] boxes against every segment's box
                sep = ((hi[:, None, 0] < st_lo_a[None, :, 0] - 0.05) | (st_hi_a[None, :, 0] + 0.05 < lo[:, None, 0]) |
                       (hi[:, None, 1] < st_lo_a[None, :, 1] - 0.05) | (st_hi_a[None, :, 1] + 0.05 < lo[:, None, 1]))
                if not sep.all():
                    if separation == "aabb":
                        continue
                    ok = True
                    for k, j in zip(*np.nonzero(~sep)):      # boxes overlap: the actual polygons decide
                        if not separable(statics[j], np.stack([cx[k], cy[k]], 1), 1e-3):
                            ok = False
                            break
                    if not ok:
                        continue
            if prev_n:         # previously accepted agents: [n][K][2] inflated windowed boxes
                plo = prev_lo[:prev_n] - infl - 0.05; phi = prev_hi[:prev_n] + infl + 0.05
                sep = ((whi[None, :, 0] < plo[:, :, 0]) | (phi[:, :, 0] < wlo[None, :, 0]) |
                       (whi[None, :, 1] < plo[:, :, 1]) | (phi[:, :, 1] < wlo[None, :, 1]))
                if not sep.all():
                    if separation == "aabb":
                        continue
                    # the LPs of both agents for this pair of guesses: my control polygon k against the other's
                    # interval hull k, and the other's control polygon k against mine
                    tk = t_scene + np.arange(K + 1) * T
                    my_hulls = [interval_hull(tk, co[:2], tk[k], tk[k + 1], T, infl) for k in range(K)]
                    ok = True
                    for j, k in zip(*np.nonzero(~sep)):
                        if not separable(prev_hulls[j][k], np.stack([cx[k], cy[k]], 1), 1e-3) or \
                           not separable(my_hulls[k], prev_cps[j][k], 1e-3):
                            ok = False
                            break
                    if not ok:
                        continue
            accepted = (g, co, lo, hi, wlo, whi, my_hulls)
            break
        if accepted is None:  # hover in place
            g = np.array([starts[i][0], starts[i][1], p.goal_height])
            co = np.zeros((3, K, 4)); co[:, :, 3] = g[:, None]
            cx = pos_ctrl_pts(co[0], T); cy = pos_ctrl_pts(co[1], T)
            lo = np.stack([cx.min(1), cy.min(1)], 1); hi = np.stack([cx.max(1), cy.max(1)], 1)
            accepted = (g, co, lo, hi, lo, hi, None)
        g, co, lo, hi, wlo, whi, my_hulls = accepted
        goals[i] = g
        prev_lo[prev_n] = wlo; prev_hi[prev_n] = whi; prev_n += 1
        if separation == "hull":
            tk = t_scene + np.arange(K + 1) * T
            cx = pos_ctrl_pts(co[0], T); cy = pos_ctrl_pts(co[1], T)
            prev_hulls.append(my_hulls if my_hulls is not None else [interval_hull(tk, co[:2], tk[k], tk[k + 1], T, infl) for k in range(K)])
            prev_cps.append([np.stack([cx[k], cy[k]], 1) for k in range(K)])
        t_start = float(rng.uniform(0, t_jitter)) if t_jitter > 0 else 0.0
        guesses[i]["K"] = K
        guesses[i]["t_start"] = t_start
        guesses[i]["coeff"][:, :K, :] = co
        r = committed[i]
        r["id"] = i + 1; r["is_agent"] = 1; r["valid"] = 1; r["n_bend"] = 1
        r["bbox"] = 2 * p.drone_radius
        r["pos"] = co[:, 0, 3]
        r["bend"][0] = p.pb[i]
        r["pwp"]["n_seg"] = K
        r["pwp"]["times"][:K + 1] = t_start + np.arange(K + 1) * T
        r["pwp"]["coeff"][:, :K, :] = co
    return dict(par=p, statics_raw=raw, statics=statics, starts=starts, goals=goals,
                guesses=guesses, committed=committed, seed=seed)



def _make_scene_job(job):
    num_agents, n_static, seed, kw = job
    return make_scene(num_agents, n_static, seed=seed, **kw)


def make_scenes(num_agents, n_static, seeds, workers=None, **kw):
    """make_scene for every seed, over a pool of host processes (a 64-agent scene takes about a second of rejection
    sampling; the bench keeps a hundred of them in flight per GPU).  Same scenes as the serial calls: a scene depends on its
    seed only."""
    import os
    seeds = list(seeds)
    if workers is None:
        workers = min(len(seeds), max(1, (os.cpu_count() or 1) // 2), 64)
    if workers <= 1 or len(seeds) <= 1:
        return [make_scene(num_agents, n_static, seed=s, **kw) for s in seeds]
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    with ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn")) as ex:
        return list(ex.map(_make_scene_job, [(num_agents, n_static, s, kw) for s in seeds]))

def scene_statics(num_agents, n_static, seed, par=None):
    """The inflated static obstacles make_scene(num_agents, n_static, seed) produces (its first random draws),
    without building the rest of the scene."""
    p = par if par is not None else scaled_params(num_agents, n_static)
    return random_static_obstacles(p, np.random.default_rng(seed))[1]


def active_rows(p, coeff, K, line_seg, line_nd, tol=1e-6):
    """Inequality rows of the spline QP (solver_gurobi_poly.cpp:433-489) that are active (slack < tol) at the
    trajectory `coeff` [3][K][4]: (box rows, line rows)."""
    T = p.T_span
    M4 = A_POS_INV * np.array([T ** 3, T ** 2, T, 1.0])[:, None]
    V3 = A_VEL_INV * (np.array([3.0, 2.0, 1.0]) * np.array([T ** 2, T, 1.0]))[:, None]
    mins = [p.x_min, p.y_min, p.z_min]; maxs = [p.x_max, p.y_max, p.z_max]
    nb = 0
    for ax in range(3):
        q = coeff[ax, :K] @ M4; v = coeff[ax, :K, :3] @ V3; a = 6 * T * coeff[ax, :K, 0] + 2 * coeff[ax, :K, 1]
        nb += int((q > maxs[ax] - tol).sum() + (q < mins[ax] + tol).sum() + (np.abs(v) > p.v_max - tol).sum() + (np.abs(a) > p.a_max - tol).sum())
    nl = 0
    if len(line_seg):
        cx = coeff[0, :K] @ M4; cy = coeff[1, :K] @ M4
        val = line_nd[:, 0:1] * cx[line_seg] + line_nd[:, 1:2] * cy[line_seg] + line_nd[:, 2:3] - 1.0
        nl = int((val > -tol).sum())
    return nb, nl


def statics_csr(statics):
    off = np.zeros(len(statics) + 1, dtype=np.int32)
    for i, s in enumerate(statics):
        off[i + 1] = off[i] + len(s)
    xy = np.concatenate(statics).astype(np.float64) if len(statics) else np.zeros((0, 2))
    return off, np.ascontiguousarray(xy)


def synthetic_entangle(sc, seed, frac=0.1, nb_range=(2, 5)):
    """Synthetic entanglement inputs for a scene: for ~frac of the agent
    pairs one active case, 2-4 bend points per agent (bend[0] is the base, neptune_ros.cpp:453-457).
    Returns case_id [N][NEP_MAX_POL][N] (row a = the block agent a+1 hands to the back end: the
    alphas case of (segment, other agent), 0 = none; solver_gurobi_poly.cpp:624-631) and writes the
    bend points into sc['committed']."""
    rng = np.random.default_rng(seed)
    p = sc["par"]; N = p.num_agents
    com = sc["committed"]
    for j in range(N):
        nb = int(rng.integers(nb_range[0], nb_range[1]))
        com[j]["n_bend"] = nb
        com[j]["bend"][0] = p.pb[j]
        for b in range(1, nb):
            # a few decimetres off another agent's guessed path, so that the distance cull
            # (:738-745, within the control polygon's length of its first point) lets LPs through
            k = int(rng.integers(0, N)); i_k = int(rng.integers(1, 7))
            q0 = np.array([sc["guesses"][k]["coeff"][0][i_k][3], sc["guesses"][k]["coeff"][1][i_k][3]])
            ang = rng.uniform(0, 2 * np.pi); rad = rng.uniform(0.2, 0.5)
            com[j]["bend"][b] = q0 + rad * np.array([np.cos(ang), np.sin(ang)])
    case_id = np.zeros((N, abi.NEP_MAX_POL, N), dtype=np.int32)
    for a in range(N):
        for j in range(N):
            if j == a or rng.uniform() > frac:
                continue
            nb = int(com[j]["n_bend"])
            cid = int(rng.integers(1, nb + 2))           # 1 .. nb+1
            s0 = int(rng.integers(0, abi.NEP_MAX_POL))
            case_id[a, s0:, j] = cid                     # active from some knot on
    return case_id


def static_reps(statics):
    """Two representative points per static obstacle (staticObsRep_) and the longest vertex distance from
    each (staticObsLongestDist_, neptune_ros.cpp:984-1002).  The reference picks the representatives
    in its ROS setup from a line through the obstacle; here: the first vertex and the one farthest
    from it (inputs of the path, not part of it)."""
    reps, longest = [], []
    for poly in statics:
        v = np.asarray(poly, dtype=np.float64).reshape(-1, 2)
        a = v[0]
        b = v[int(np.argmax(((v - a) ** 2).sum(axis=1)))]
        reps.append([a, b])
        longest.append([float(np.sqrt(((v - a) ** 2).sum(axis=1)).max()), float(np.sqrt(((v - b) ** 2).sum(axis=1)).max())])
    return np.array(reps).reshape(-1, 2, 2), np.array(longest).reshape(-1, 2)


def real_entangle(sc, num_samples=3, cable_length=None, use_statics=True):
    """Entanglement inputs of every agent's back-end call from the actual geometry:
    the guess is swept through the other agents' sampled committed trajectories and tether polylines
    with the host library (neptune_amd.entangle), starting from an empty entangle state.  Returns
    (case_id [N][NEP_MAX_POL][N], entangled_at [N], per-agent result dicts)."""
    from . import entangle
    p = sc["par"]; N = p.num_agents
    com = sc["committed"]
    reps, longest = static_reps(sc["statics"]) if (use_statics and len(sc["statics"])) else (np.zeros((0, 2, 2)), np.zeros((0, 2)))
    cable = cable_length if cable_length is not None else p.tether_length
    case_id = np.zeros((N, abi.NEP_MAX_POL, N), dtype=np.int32)
    hit = np.zeros(N, dtype=np.int32)
    res = []
    bend = [np.array(com[j]["bend"][: int(com[j]["n_bend"])]) for j in range(N)]
    for a in range(N):
        g = sc["guesses"][a]
        t0 = float(g["t_start"])
        sampled = np.zeros((N, p.num_pol, num_samples + 1, 2))
        present = np.zeros(N, dtype=np.int32)
        for j in range(N):
            if j == a or not com[j]["valid"]:
                continue
            sampled[j] = entangle.sample_points(com[j]["pwp"], t0, t0 + p.num_pol * p.T_span, p.num_pol, num_samples)
            present[j] = 1
        chk = entangle.EntangleCheck(N, a + 1, p.num_pol, num_samples, p.T_span, cable, p.pb, reps, longest)
        chk.set_inputs(sampled, present, bend)
        r = chk.propagate_guess(chk.new_state(), g)
        case_id[a] = r["case_id"]; hit[a] = r["entangled_at"]
        res.append(r)
    return case_id, hit, res


def tether_crossing_scene(num_agents, n_static, seed):
    """A scene in which tethers actually get crossed between an agent and its base: every other agent
    hovers just beyond somebody else's guessed path (seen from its own base), with a long tether.
    Returns the scene with enable_entangle set (inputs for real_entangle / the entangle rows)."""
    import dataclasses
    rng = np.random.default_rng(seed)
    sc = make_scene(num_agents, n_static, seed=seed, separation="aabb")   # (the hover positions below were tuned on these guesses)
    p = dataclasses.replace(sc["par"], enable_entangle=True, tether_length=200.0)
    sc["par"] = p
    com, gue = sc["committed"], sc["guesses"]
    N = num_agents
    for j in range(0, N, 2):
        a = (j + 3) % N
        k = int(rng.integers(2, 6))
        q = np.array([gue[a]["coeff"][0][k][3], gue[a]["coeff"][1][k][3]])
        d = np.asarray(p.pb[j]) - q
        d /= np.linalg.norm(d)
        pos = q - d * rng.uniform(0.3, 0.9) + rng.normal(scale=0.1, size=2)
        n = int(com[j]["pwp"]["n_seg"])
        com[j]["pwp"]["coeff"][:, :, :] = 0
        com[j]["pwp"]["coeff"][0, :n, 3] = pos[0]; com[j]["pwp"]["coeff"][1, :n, 3] = pos[1]; com[j]["pwp"]["coeff"][2, :n, 3] = 1.0
        com[j]["pos"][:2] = pos
    return sc


def frontend_cfg(p, beam_width=32, num_samples=5, pad_hold=0, entangle=False, ent_samples=3):
    """The front-end settings Neptune's constructor passes (neptune.cpp:92-97) with the reference yaml values
    (a_star_samp_x 5, a_star_fraction_voxel_size 0.2, goal_radius 0.2, bias 1.1)."""
    return abi.nep_fe_cfg(p.j_max, 0.2, 1.1, 0.2, p.tether_length, num_samples, beam_width, pad_hold, 1 if entangle else 0, ent_samples, 0)


def frontend_starts(sc):
    """Point A of every agent (the state its scene guess starts from) and its goal -> [N] FE_START_DTYPE."""
    N = sc["par"].num_agents
    st = np.zeros(N, dtype=abi.FE_START_DTYPE)
    for a in range(N):
        co = np.array(sc["guesses"][a]["coeff"])
        st[a]["pos"] = co[:, 0, 3]; st[a]["vel"] = co[:, 0, 2]; st[a]["accel"] = 2 * co[:, 0, 1]
        st[a]["goal"] = sc["goals"][a]
        st[a]["t_start"] = sc["guesses"][a]["t_start"]
    return st


def crossing_scene(sc, K=8):
    """The hard variant of a scene for the closed-loop legs ("circle swap"): every agent starts at rest on the base circle
    (neptune_ros.cpp:138-151) and its goal is the antipodal point — the start of the agent opposite — so the whole fleet
    crosses the middle of the world at once, against the scene's own static obstacles.  -> (starts [N] FE_START_DTYPE,
    committed [N] TRAJ_REC_DTYPE: everybody hovering at its start)."""
    p = sc["par"]; N = p.num_agents; T = p.T_span
    st = np.zeros(N, dtype=abi.FE_START_DTYPE)
    com = np.zeros(N, dtype=abi.TRAJ_REC_DTYPE)
    for a in range(N):
        pos = np.array([sc["starts"][a][0], sc["starts"][a][1], p.goal_height])
        st[a]["pos"] = pos; st[a]["goal"] = np.array([-pos[0], -pos[1], p.goal_height]); st[a]["t_start"] = 0.0
        r = com[a]
        r["id"] = a + 1; r["is_agent"] = 1; r["valid"] = 1; r["n_bend"] = 1
        r["bbox"] = 2 * p.drone_radius
        r["pos"] = pos
        r["bend"][0] = p.pb[a]
        r["pwp"]["n_seg"] = K
        r["pwp"]["times"][:K + 1] = np.arange(K + 1) * T
        r["pwp"]["coeff"][:, :K, 3] = pos[:, None]
    return st, com


def reachable_goals(sc, margin=0.4):
    """Goals of a scene with those that fall inside an inflated static obstacle (the scene generator only
    keeps the first K segments of the way clear) pushed out through the nearest face, `margin` beyond it."""
    goals = np.array(sc["goals"], dtype=np.float64)
    for i in range(len(goals)):
        for _ in range(4):
            moved = False
            for s in sc["statics"]:
                v = np.asarray(s, dtype=np.float64)
                lo, hi = v.min(axis=0) - margin, v.max(axis=0) + margin
                g = goals[i, :2]
                if (g > lo).all() and (g < hi).all():
                    d = np.array([g[0] - lo[0], hi[0] - g[0], g[1] - lo[1], hi[1] - g[1]])
                    k = int(np.argmin(d))
                    goals[i, 0 if k < 2 else 1] = [lo[0], hi[0], lo[1], hi[1]][k]
                    moved = True
            if not moved:
                break
    return goals
