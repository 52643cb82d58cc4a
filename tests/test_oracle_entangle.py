"""CPU: the oracle's entanglement rows (solver_gurobi_poly.cpp:620-637, 715-764)."""
import numpy as np

from neptune_amd import abi, scene


def test_entangle_adds_lines_that_separate(oracle):
    sc = scene.make_scene(6, 0, seed=11)
    case_id = scene.synthetic_entangle(sc, seed=5, frac=0.6)
    p = sc["par"]
    T = p.T_span
    M4 = scene.A_POS_INV * np.array([T ** 3, T ** 2, T, 1.0])[:, None]
    total_extra = 0
    for a in range(6):
        g = sc["guesses"][a]; K = int(g["K"])
        r0 = oracle.replan(p, a + 1, sc["committed"], g, sc["statics"])
        r1 = oracle.replan(p, a + 1, sc["committed"], g, sc["statics"], case_id=case_id[a])
        assert r1["n_lp"] >= r0["n_lp"]
        total_extra += r1["n_lines"] - r0["n_lines"]
        # every line (entangle ones included) keeps the guess's control points on its B side
        ci = np.array(g["coeff"])[:, :K, :]
        cx = ci[0] @ M4; cy = ci[1] @ M4
        for s_, l in zip(r1["line_seg"], r1["line_nd"]):
            assert (l[0] * cx[s_] + l[1] * cy[s_] + l[2]).max() <= -1 + 1e-9
        # no case for agent j at segment i  =>  identical to the plain run
    assert total_extra > 0


def test_case_zero_and_own_agent_are_skipped(oracle):
    sc = scene.make_scene(3, 0, seed=3)
    scene.synthetic_entangle(sc, seed=1, frac=1.0)
    p = sc["par"]
    z = np.zeros((abi.NEP_MAX_POL, 3), dtype=np.int32)
    r0 = oracle.replan(p, 1, sc["committed"], sc["guesses"][0], sc["statics"])
    rz = oracle.replan(p, 1, sc["committed"], sc["guesses"][0], sc["statics"], case_id=z)
    assert rz["n_lp"] == r0["n_lp"] and np.array_equal(rz["line_nd"], r0["line_nd"])
    own = z.copy(); own[:, 0] = 1                     # own column is never read (:622)
    ro = oracle.replan(p, 1, sc["committed"], sc["guesses"][0], sc["statics"], case_id=own)
    assert ro["n_lp"] == r0["n_lp"]
