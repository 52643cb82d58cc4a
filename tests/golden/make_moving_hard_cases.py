"""Generator of tests/golden/moving_hard_cases.npz — the HARD replans of the closed loop (bench.py's `moving` leg) as stand-alone
QPs with independent feasibility labels.

Input: gpurun_out/moving_hard_raw.npz, written on a GPU box by scripts/dump_moving_cases.py (after a number of closed-loop rounds:
the replans of three rounds whose first solve took more than 14 iterations or did not end with status 0, each as its guess and
the separating lines the separator made for it; a few easy replans for control).  This script runs in the build container (numpy +
SciPy's HiGHS, nothing of the product or the oracle decides a label): for every case it builds the reference's LINEAR rows in the
reference's own 12K-variable space (make_golden.build_qp: solver_gurobi_poly.cpp:385-710, first problem and the relaxed
re-solve's :838-861) and asks HiGHS for the largest margin t with  G x + t <= h, E x = e  (t <= 1):

    t < -1e-6   the linear rows alone are infeasible, decisively (so is the problem with the terminal ball, a subset)
    t > +1e-6   strictly feasible with an interior
    in between  razor-thin: a feasible set without an interior (a start state exactly on a bound) — no label

Expected status (what PolySolverGurobi::optimize returns, solver_gurobi_poly.cpp:832-861):
    FAILED (2)   first AND relaxed problem decisively infeasible
    RELAXED (1)  first decisively infeasible, relaxed strictly feasible and no terminal-ball row (which the LP cannot judge)
    OK (0)       first strictly feasible, no terminal-ball row
    -1           no label (ball row present, or razor-thin)
Usage (container): python tests/golden/make_moving_hard_cases.py [raw.npz]"""
import os
import sys

import numpy as np
import scipy.optimize as so

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import build_qp  # noqa: E402


def margin(Q):
    """max t s.t. G x + t <= h, E x = e, t <= 1 (HiGHS); None when the LP solver gives no answer"""
    n = Q["n"]
    c = np.zeros(n + 1); c[-1] = -1.0
    A = np.hstack([Q["G"], np.ones((len(Q["h"]), 1))])
    Ae = np.hstack([Q["E"], np.zeros((len(Q["e"]), 1))])
    r = so.linprog(c, A_ub=A, b_ub=Q["h"], A_eq=Ae, b_eq=Q["e"], bounds=[(None, None)] * n + [(None, 1.0)], method="highs")
    if r.status == 2:                       # equalities alone infeasible
        return -np.inf
    return float(r.x[-1]) if r.status == 0 else None


def main():
    raw = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "moving_hard_raw.npz"))
    b = raw["bounds"]; mins = [b[0], b[2], b[4]]; maxs = [b[1], b[3], b[5]]; v_max, a_max, T, w = b[6], b[7], b[8], b[9]
    n = len(raw["status"])
    out = dict(K=[], coeff=[], t_first=[], t_relaxed=[], ball=[], expected=[], device_status=[], hard=[], iters_first=[])
    off = raw["line_off"]
    TH = 1e-6
    for k in range(n):
        g = raw["guess"][k]; K = int(g["K"])
        ci = np.array(g["coeff"])[:, :K, :]
        seg = raw["line_seg"][off[k]:off[k + 1]].astype(np.int32); nd = raw["line_nd"][off[k]:off[k + 1]]
        Q1 = build_qp(K, T, w, mins, maxs, v_max, a_max, ci, seg, nd, relaxed=False)
        Q2 = build_qp(K, T, w, mins, maxs, v_max, a_max, ci, seg, nd, relaxed=True)
        t1, t2 = margin(Q1), margin(Q2)
        ball = bool(Q1["has_qc"])
        exp = -1
        if t1 is not None and t2 is not None:
            if t1 < -TH and t2 < -TH:
                exp = 2
            elif t1 < -TH and t2 > TH and not ball:
                exp = 1
            elif t1 > TH and not ball:
                exp = 0
        out["K"].append(K); out["coeff"].append(np.array(g["coeff"])); out["t_first"].append(np.nan if t1 is None else t1)
        out["t_relaxed"].append(np.nan if t2 is None else t2); out["ball"].append(ball); out["expected"].append(exp)
        out["device_status"].append(int(raw["status"][k])); out["hard"].append(bool(raw["hard"][k])); out["iters_first"].append(int(raw["iters_first"][k]))
        print("case %3d K=%d lines=%4d t_first=%+.3e t_relaxed=%+.3e ball=%d expected=%2d device=%d%s" %
              (k, K, len(seg), out["t_first"][-1], out["t_relaxed"][-1], ball, exp, int(raw["status"][k]), "" if raw["hard"][k] else "  (control)"), flush=True)
    exp = np.array(out["expected"]); dev = np.array(out["device_status"])
    print("labels: failed %d, relaxed %d, ok %d, none %d; device status at dump time differs from a label in %d cases"
          % ((exp == 2).sum(), (exp == 1).sum(), (exp == 0).sum(), (exp < 0).sum(), int(((exp >= 0) & (exp != dev)).sum())))
    # the committed fixture holds a subset (a case is ~13 KB of lines): every replan the device did not solve at the first attempt,
    # every replan whose device status differs from its label, every fifteenth of the others that were hard, and the controls
    hard = np.array(out["hard"])
    keep = (dev != 0) | ((exp >= 0) & (exp != dev)) | ~hard
    others = np.nonzero(~keep)[0]
    keep[others[::15]] = True
    idx = np.nonzero(keep)[0]
    new_off = np.zeros(len(idx) + 1, dtype=np.int64)
    segs, nds = [], []
    for j, k in enumerate(idx):
        segs.append(raw["line_seg"][off[k]:off[k + 1]]); nds.append(raw["line_nd"][off[k]:off[k + 1]])
        new_off[j + 1] = new_off[j] + (off[k + 1] - off[k])
    sel = lambda a: np.asarray(a)[idx]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "moving_hard_cases.npz"),
                        K=sel(out["K"]).astype(np.int8), coeff=sel(out["coeff"]), line_off=new_off, line_seg=np.concatenate(segs), line_nd=np.concatenate(nds),
                        t_first=sel(out["t_first"]), t_relaxed=sel(out["t_relaxed"]), ball=sel(out["ball"]), expected=sel(exp).astype(np.int8),
                        device_status_at_dump=sel(dev).astype(np.int8), hard=sel(hard), iters_first=sel(out["iters_first"]).astype(np.int16),
                        where=raw["where"][idx], bounds=b,
                        population=np.array([n, int(hard.sum()), int((exp == 2).sum()), int((exp == 1).sum()), int((exp == 0).sum()), int((exp < 0).sum())]))
    print("kept %d of %d cases" % (len(idx), n))


if __name__ == "__main__":
    main()
