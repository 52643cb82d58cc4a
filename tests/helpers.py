"""Shared helpers of the parity tests."""
import json
import os

import numpy as np

from neptune_amd import scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_qp_cases(name="qp_cases.npz"):
    """qp_cases.npz: the round-1 set (make_golden.py); qp_cases_r2.npz: BASELINE config-4 / config-5 sizes, K = 7 and
    front-end guesses (make_golden_r2.py), each accepted on a KKT certificate."""
    d = np.load(os.path.join(ROOT, "tests", "golden", name), allow_pickle=True)
    out = []
    for i in range(int(d["n"])):
        c = {k[len("c%d_" % i):]: d[k] for k in d.files if k.startswith("c%d_" % i)}
        c["tag"] = str(c["tag"]); c["K"] = int(c["K"]); c["status"] = int(c["status"])
        c["cost"] = float(c["cost"]); c["dth"] = float(c["dth"]); c["dcost"] = float(c["dcost"])
        out.append(c)
    return out


def params_of_case(c):
    """Params for a single-agent QP case (bases far away so that no base lines appear)."""
    p = scene.Params(num_agents=1, pb=np.full((1, 2), 1e6))
    p.x_min, p.y_min, p.z_min = [float(x) for x in c["mins"]]
    p.x_max, p.y_max, p.z_max = [float(x) for x in c["maxs"]]
    p.T_span = float(c["T"]); p.weight = float(c["weight"]); p.v_max = float(c["v_max"]); p.a_max = float(c["a_max"])
    return p


def golden_theta_out(c):
    """Golden optimum with the reference's z override applied (solver_gurobi_poly.cpp:879-880)."""
    K = c["K"]; ci = c["coeff_init"]; T = float(c["T"])
    th = c["theta"].copy()
    tp = np.array([T ** 3, T ** 2, T, 1.0]); final = ci[:, K - 1, :] @ tp
    if c["status"] != 2 and np.hypot(ci[0, 0, 3] - final[0], ci[1, 0, 3] - final[1]) < 1.0:
        th[2] = ci[2]
    return th


def theta_tol(c):
    """Fixtures whose two SciPy solves could not be polished carry their disagreement."""
    return max(5e-7, 2.0 * c["dth"]) if c["dth"] > 5e-7 and c["tag"] in ("tight K8 seed11", "nostop K2") else 2e-7


def load_kat():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "minvo_kat.json")))


def ent_inputs(sc, a, num_samples=3, init=None, recs=None, t0=None):
    """The entangle inputs of agent a's (0-based) search as the reference's setUp assembles them (kinodynamic_search.cpp:
    190-257): the other agents' committed trajectories sampled on the planning grid (by the Python restatement,
    oracle/entangle_oracle.py), their tether bend points, the static representatives.  -> dict for oracle.frontend_beam_ent."""
    from neptune_amd import abi
    from oracle import entangle_oracle as eo
    p = sc["par"]; N = p.num_agents
    com = sc["committed"] if recs is None else recs
    t0 = float(sc["guesses"][a]["t_start"]) if t0 is None else t0
    sampled = np.zeros((N, p.num_pol, num_samples + 1, 2)); present = np.zeros(N, dtype=np.int32)
    bend_n = np.zeros(N, dtype=np.int32); bend_xy = np.zeros((N, abi.NEP_MAX_BEND, 2))
    for j in range(N):
        r = com[j]
        ok = bool(r["valid"]) and bool(r["is_agent"]) and int(r["pwp"]["n_seg"]) >= 1
        nb = int(r["n_bend"]) if ok else 0
        bend_n[j] = nb
        bend_xy[j, :nb] = np.array(r["bend"])[:nb]
        if j == a or not ok:
            present[j] = 1 if ok else 0          # (the own entry is skipped by id, not by presence)
            if not ok:
                continue
        pw = r["pwp"]; n = int(pw["n_seg"])
        pts = eo.sample_points_of_intervals(np.array(pw["times"])[:n + 1].tolist(), np.array(pw["coeff"])[0, :n].tolist(),
                                            np.array(pw["coeff"])[1, :n].tolist(), t0, t0 + p.num_pol * p.T_span, p.num_pol, num_samples)
        sampled[j] = np.array(pts).reshape(p.num_pol, num_samples + 1, 2)
        present[j] = 1
    reps, longest = scene.static_reps(sc["statics"]) if len(sc["statics"]) else (np.zeros((0, 2, 2)), np.zeros((0, 2)))
    return dict(num_samples=num_samples, reps=reps, longest=longest, sampled=sampled, present=present, bend_n=bend_n, bend_xy=bend_xy, init=init)


def bundle_scene(n_agents, a, t_range, P=(-20.0, -1.0), Q=(20.0, 1.0)):
    """Every other agent hovers where its tether (base -> hover point) crosses agent a's straight path P -> Q at a point of its own,
    the crossing points spread over the fraction t_range of the path: a bundle of n_agents - 1 tethers to fly through (sc["bundle"] = P, Q)."""
    from neptune_amd import abi, scene
    import dataclasses
    sc = scene.make_scene(n_agents, 0, seed=77, separation="aabb")
    p = dataclasses.replace(sc["par"], enable_entangle=True, tether_length=1000.0)
    sc["par"] = p
    P = np.array(P, dtype=np.float64); Q = np.array(Q, dtype=np.float64); K = abi.NEP_MAX_POL; T = p.T_span
    g = sc["guesses"][a]
    g["K"] = K; g["coeff"][:, :, :] = 0
    v = (Q - P) / (K * T)
    for i in range(K):
        for ax in range(2):
            g["coeff"][ax][i][2] = v[ax]; g["coeff"][ax][i][3] = P[ax] + v[ax] * i * T
    others = [j for j in range(n_agents) if j != a]
    for k, j in enumerate(others):
        t = t_range[0] + (t_range[1] - t_range[0]) * k / len(others)      # (a sampled step of the guess covers 1/24 of the path)
        M = P + t * (Q - P)
        pos = 2 * M - np.asarray(p.pb[j])
        com = sc["committed"][j]; n = int(com["pwp"]["n_seg"])
        com["pwp"]["coeff"][:, :, :] = 0
        com["pwp"]["coeff"][0, :n, 3] = pos[0]; com["pwp"]["coeff"][1, :n, 3] = pos[1]; com["pwp"]["coeff"][2, :n, 3] = 1.0
        com["pos"][:2] = pos
    sc["bundle"] = (P, Q)
    return sc


def load_moving_hard_cases():
    """tests/golden/moving_hard_cases.npz (make_moving_hard_cases.py): the hard replans of the closed loop as stand-alone QPs —
    guess, separating lines, HiGHS feasibility margins of the reference's linear rows and the status they imply
    -> (Params, list of dicts)"""
    d = np.load(os.path.join(ROOT, "tests", "golden", "moving_hard_cases.npz"))
    b = d["bounds"]
    p = scene.Params(num_agents=1, pb=np.full((1, 2), 1e6))
    p.x_min, p.x_max, p.y_min, p.y_max, p.z_min, p.z_max = [float(x) for x in b[:6]]
    p.v_max, p.a_max, p.T_span, p.weight = float(b[6]), float(b[7]), float(b[8]), float(b[9])
    off = d["line_off"]
    out = []
    for k in range(len(d["K"])):
        K = int(d["K"][k])
        out.append(dict(K=K, coeff=d["coeff"][k][:, :K, :].copy(), seg=d["line_seg"][off[k]:off[k + 1]].astype(np.int32), nd=d["line_nd"][off[k]:off[k + 1]].copy(),
                        expected=int(d["expected"][k]), t_first=float(d["t_first"][k]), t_relaxed=float(d["t_relaxed"][k]), ball=bool(d["ball"][k]),
                        device_status_at_dump=int(d["device_status_at_dump"][k]), hard=bool(d["hard"][k])))
    return p, out
