"""GPU tests of the boundary's failure semantics (round-2 advisor findings): what a failed or empty replan
publishes, polygon orientation at upload, the hull capacity flag, one static-obstacle set per scene."""
import numpy as np
import pytest

import helpers
from neptune_amd import abi, scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from neptune_amd import backend
    return backend


def _infeasible_guess(g, p):
    """start outside the world box: the position rows of the first control point cannot hold in either solve"""
    g = g.copy()
    co = np.array(g["coeff"])
    co[0, :, 3] += (p.x_max + 5.0) - co[0, 0, 3]
    g["coeff"] = co
    return g


def test_failed_and_empty_replans_publish_nothing(be, oracle):
    """neptune_ros.cpp:651-663: a replan that fails publishes nothing, the agent keeps flying its committed
    trajectory.  Slot 1 fails (both solves infeasible), slot 2 has an empty guess (front-end miss, K = 0):
    their commit slots carry the previous records over, everything else is the new trajectory."""
    sc = scene.make_scene(5, 3, seed=4)
    p = sc["par"]
    gue = sc["guesses"].copy()
    gue[1] = _infeasible_guess(gue[1], p)
    gue[2]["K"] = 0
    r1 = oracle.replan(p, 2, sc["committed"], gue[1], sc["statics"])
    assert r1["status"] == 2
    prev = sc["committed"].copy()
    prev["pos"][:, 2] = 7.25                     # a marker the new records cannot carry
    bb = be.BatchBackend(p, sc["statics"])
    bb.replan(bb.to_device(prev), bb.to_device(gue))
    sol = bb.solutions(); com = bb.commits()
    assert [int(s["stats"]["status"]) for s in sol][1:3] == [2, 2] and all(int(sol[a]["stats"]["status"]) != 2 for a in (0, 3, 4))
    assert int(sol[2]["K"]) == 0 and int(sol[2]["n_states"]) == 0 and not np.array(sol[2]["coeff"]).any()
    np.testing.assert_array_equal(np.array(sol[1]["coeff"]), np.array(gue[1]["coeff"]))      # output == initial guess (:856-859)
    for a in (1, 2):
        assert com[a].tobytes() == prev[a].tobytes()
    for a in (0, 3, 4):
        assert int(com[a]["valid"]) == 1 and com[a]["pos"][2] != 7.25
        np.testing.assert_array_equal(np.array(com[a]["pwp"]["coeff"])[:, :8, :], np.array(sol[a]["coeff"]))
    # without the previous records (hulls reused: d_committed = None) the slot is left as the caller passed it
    bb.d_commit.fill_(0xAB)
    bb.replan(None, bb.to_device(gue))
    com2 = bb.commits()
    for a in (1, 2):
        assert (np.frombuffer(com2[a].tobytes(), dtype=np.uint8) == 0xAB).all()
    assert int(com2[0]["valid"]) == 1
    # and the safety pass keeps the previous record of such agents whatever it decides about the others
    d_final = bb.torch.zeros_like(bb.d_commit); d_acc = bb.torch.zeros(5, dtype=bb.torch.int32, device=bb.device)
    bb.replan(bb.to_device(prev), bb.to_device(gue))
    bb.safety_commit(bb.to_device(prev), bb.d_commit, bb.to_device(gue), d_final, d_acc)
    fin = d_final.cpu().numpy().view(abi.TRAJ_REC_DTYPE)
    for a in (1, 2):
        assert fin[a].tobytes() == prev[a].tobytes()
    bb.close()


def test_clockwise_polygons_are_reoriented_and_nonconvex_refused(be, oracle):
    """The reference LP (separator_glpk.cpp:248-373) does not care about vertex order; the kernel's edge rule
    does.  Clockwise statics / hull lists give the result of their counter-clockwise form, non-convex input
    is refused."""
    from neptune_amd._lib import BackendError
    sc = scene.make_scene(5, 6, seed=9)
    p = sc["par"]
    cw = [np.ascontiguousarray(np.vstack([s[:1], s[1:][::-1]])) for s in sc["statics"]]       # same first vertex, reversed
    outs = []
    for statics in (sc["statics"], cw):
        bb = be.BatchBackend(p, statics)
        bb.set_line_cull(0.0)                   # (every line, in the reference's call order: compared with the oracle's list below)
        bb.replan(bb.to_device(sc["committed"]), bb.to_device(sc["guesses"]))
        outs.append((bb.solutions(), [bb.debug_lines(a) for a in range(5)]))
        bb.close()
    n_static_lines = 0
    for a in range(5):
        np.testing.assert_array_equal(outs[0][1][a][1], outs[1][1][a][1])
        r = oracle.replan(p, a + 1, sc["committed"], sc["guesses"][a], sc["statics"])
        r0 = oracle.replan(p, a + 1, sc["committed"], sc["guesses"][a], [])
        n_static_lines += r["n_lp"] - r0["n_lp"]
        np.testing.assert_array_equal(outs[1][1][a][1], r["line_nd"])
    assert n_static_lines > 0
    assert outs[0][0].tobytes() == outs[1][0].tobytes()
    # a clockwise unit square against points at x >= 3 (the advisor's example): a separating line with A on its >= 1 side
    sq_cw = np.array([[0.0, 0.0], [0.0, 1.0], [1.0, 1.0], [1.0, 0.0]])
    s = be.PolySolver(p.num_pol, 3, 1, p.T_span, p.pb, p.weight, 0.5, True)
    s.setStaticObstVert([sq_cw])
    with pytest.raises(BackendError):
        s.setStaticObstVert([np.array([[0.0, 0.0], [2.0, 0.0], [0.5, 0.5], [0.0, 2.0]])])    # reflex vertex
    s.close()
    with pytest.raises(BackendError):
        be.BatchBackend(p, [np.array([[0.0, 0.0], [2.0, 0.0], [0.5, 0.5], [0.0, 2.0]])] * 6)


def _short_segment_record(n_short, T=0.5):
    """a committed trajectory whose first planning interval [0, T] is covered by n_short short segments"""
    rec = np.zeros(1, dtype=abi.TRAJ_REC_DTYPE)
    r = rec[0]
    knots = list(np.linspace(0.0, 0.42, n_short + 1)) + [1.0, 1.5, 2.0, 2.5, 3.0, 3.5, 4.0]
    n = len(knots) - 1
    r["id"] = 1; r["is_agent"] = 1; r["valid"] = 1; r["n_bend"] = 1; r["bbox"] = 1.2
    r["pwp"]["n_seg"] = n
    r["pwp"]["times"][: n + 1] = knots
    rng = np.random.default_rng(3)
    co = rng.normal(size=(3, n, 4)) * np.array([0.05, 0.1, 0.5, 3.0])
    r["pwp"]["coeff"][:, :n, :] = co
    return rec


def test_hull_capacity_is_flagged_not_truncated(be, oracle):
    """The reference takes every committed segment overlapping an interval (neptune.cpp:392-449).  Four fit a wave
    (bit-exact against the oracle); a fifth is NEP_E_CAP, never a silently smaller hull."""
    from neptune_amd._lib import BackendError
    p = scene.scaled_params(2, 0)
    rec4 = _short_segment_record(3)          # interval 0 overlaps segments 0..3
    pw = abi.nep_pwp.from_buffer_copy(rec4[0]["pwp"].tobytes())
    d = np.array([0.6 + p.drone_radius, 0.6 + p.drone_radius])
    h, hu, ov = oracle.hull_of_interval(pw, 0.0, p.T_span, p.T_span, d, with_overflow=True)
    hx, hn, h0, n0 = be.hulls_batch(rec4, 0.0, p.num_pol, p.T_span, p.drone_radius)
    if not ov:
        assert hn[0, 0] == len(h)
        np.testing.assert_array_equal(hx[0, 0, :len(h)], h)
    rec6 = _short_segment_record(6)
    pw6 = abi.nep_pwp.from_buffer_copy(rec6[0]["pwp"].tobytes())
    assert oracle.hull_of_interval(pw6, 0.0, p.T_span, p.T_span, d, with_overflow=True)[2]
    with pytest.raises(BackendError, match="NEP_HULL_MAX_CP"):
        be.hulls_batch(rec6, 0.0, p.num_pol, p.T_span, p.drone_radius)
    # batched handle: the asynchronous replan cannot return it, nep_batch_check does
    sc = scene.make_scene(2, 0, seed=1)
    com = sc["committed"].copy(); com[1] = rec6[0]; com[1]["id"] = 2
    bb = be.BatchBackend(sc["par"], [])
    bb.replan(bb.to_device(sc["committed"]), bb.to_device(sc["guesses"]))
    bb.check()                                # nothing flagged
    bb.replan(bb.to_device(com), bb.to_device(sc["guesses"]))
    with pytest.raises(BackendError, match="NEP_HULL_MAX_CP"):
        bb.check()
    bb.check()                                # reported once, then cleared
    bb.close()


def test_one_static_set_per_scene(be):
    """nep_batch_set_scene_statics: two scenes with their own obstacles in one handle equal two handles."""
    scs = [scene.make_scene(6, 8, seed=s) for s in (30, 31)]
    p = scs[0]["par"]
    com = np.stack([s["committed"] for s in scs]); gue = np.stack([s["guesses"] for s in scs])
    bb = be.BatchBackend(p, scs[0]["statics"], n_scenes=2)
    bb.set_scene_statics(1, scs[1]["statics"])
    bb.replan(bb.to_device(com), bb.to_device(gue))
    both = bb.solutions().reshape(2, 6)
    lines = [[bb.debug_lines(s * 6 + a)[1] for a in range(6)] for s in range(2)]
    bb.close()
    for s in range(2):
        one = be.BatchBackend(p, scs[s]["statics"])
        one.replan(one.to_device(scs[s]["committed"]), one.to_device(scs[s]["guesses"]))
        assert one.solutions().tobytes() == both[s].tobytes()
        for a in range(6):
            np.testing.assert_array_equal(one.debug_lines(a)[1], lines[s][a])
        one.close()
    assert both[0].tobytes() != both[1].tobytes()


def test_wall_clock_time_limit(be, oracle):
    """setMaxRuntime -> TimeLimit (solver_gurobi_poly.cpp:812): a budget no solve can meet turns every replan into
    "no solution" (both solves time out: output == initial guess); the reference's 0.05 s changes nothing."""
    sc = scene.make_scene(5, 3, seed=4)
    p = sc["par"]
    bb = be.BatchBackend(p, sc["statics"])
    bb.set_line_cull(0.0)                    # (every replan iterates: under the presolve — the default — these scenes need no iteration and no time)
    d_com = bb.to_device(sc["committed"]); d_g = bb.to_device(sc["guesses"])
    bb.replan(d_com, d_g)
    ref = bb.solutions()
    bb.set_max_runtime(0.05)
    bb.replan(d_com, d_g)
    assert bb.solutions().tobytes() == ref.tobytes()
    bb.set_max_runtime(1e-8)                 # one tick of the 100 MHz wall clock
    bb.replan(d_com, d_g)
    sol = bb.solutions()
    moving = [a for a in range(5) if int(ref[a]["stats"]["iters"]) > 0]
    assert moving
    for a in moving:
        assert int(sol[a]["stats"]["status"]) == 2
        np.testing.assert_array_equal(np.array(sol[a]["coeff"]), np.array(sc["guesses"][a]["coeff"]))
    bb.set_max_runtime(0.0)
    bb.replan(d_com, d_g)
    assert bb.solutions().tobytes() == ref.tobytes()
    bb.close()


def test_front_end_launch_order_does_not_change_results(be):
    """The front end's searches are started longest-expected-first, a few whole scenes per XCD (order_xcd_kernel; the key remembers
    the slot's earlier search times): a scheduling matter — guesses and results of the ordered launches (first: XCD placement alone,
    then with the keys) equal those of the slot-order launch byte for byte, and every search reports its device time."""
    scs = [scene.make_scene(64, 20, seed=s) for s in (0, 1, 2)]
    S = 24                                                        # 1 536 searches: more than one wave of workgroups, a multiple of 8
    p = scs[0]["par"]
    bb = be.BatchBackend(p, scs[0]["statics"], n_scenes=S)
    for s in range(1, S):
        bb.set_scene_statics(s, scs[s % 3]["statics"])
    T = bb.torch
    d_com = bb.to_device(np.stack([scs[s % 3]["committed"] for s in range(S)])); d_st = bb.to_device(np.stack([scene.frontend_starts(scs[s % 3]) for s in range(S)]))
    fe = scene.frontend_cfg(p, beam_width=32, pad_hold=1)
    d_g = T.zeros(S * 64 * abi.GUESS_DTYPE.itemsize, dtype=T.uint8, device=bb.device); d_r = T.zeros(S * 64 * abi.FE_RESULT_DTYPE.itemsize, dtype=T.uint8, device=bb.device)
    outs = []
    for k in range(4):
        if k == 3:
            bb.set_launch_order(False)
        d_g.zero_(); d_r.zero_()
        bb.frontend(fe, d_com, d_st, d_g, d_r); T.cuda.synchronize()
        outs.append((d_g.cpu().numpy().tobytes(), d_r.cpu().numpy().tobytes()))
        if k < 3:
            us = bb.fe_search_us()
            assert (us > 0).all() and us.max() < 1e5
    assert outs[0] == outs[3] and outs[1] == outs[3] and outs[2] == outs[3]
    bb.close()


def test_launch_order_does_not_change_results(be):
    """nep_batch_set_launch_order: from a handle's second replan on, the QP workgroups of a batch of more than 1 024 replans are
    launched longest-expected-first (the previous replan's measured device time is the key).  A scheduling matter: the
    records of the ordered launch equal those of the slot-order launch byte for byte, and the order is a permutation that
    starts with the slots that took longest."""
    scs = [scene.make_scene(64, 20, seed=s) for s in (0, 1, 2)]
    S = 18                                                        # 1 152 replans: more than one wave of workgroups
    p = scs[0]["par"]
    com = np.stack([scs[s % 3]["committed"] for s in range(S)]); gue = np.stack([scs[s % 3]["guesses"] for s in range(S)])
    bb = be.BatchBackend(p, scs[0]["statics"], n_scenes=S)
    for s in range(1, S):
        bb.set_scene_statics(s, scs[s % 3]["statics"])
    d_com, d_gue = bb.to_device(com), bb.to_device(gue)
    bb.replan(d_com, d_gue)
    assert bb.launch_order() is None                              # first replan of the handle: slot order
    first = bb.solutions(); t_first = bb.solutions(timing=True)["stats"]["solve_us"]
    assert (t_first > 0).all()
    bb.replan(d_com, d_gue)
    order = bb.launch_order()
    assert order is not None and sorted(order.tolist()) == list(range(S * 64))
    assert t_first[order[:64]].mean() > t_first[order[-64:]].mean() + 8.0      # (8 us bins)
    assert bb.solutions().tobytes() == first.tobytes()
    bb.set_launch_order(False)
    bb.replan(d_com, d_gue)
    assert bb.launch_order() is None and bb.solutions().tobytes() == first.tobytes()
    bb.close()


def test_both_hull_kernels_give_the_oracles_hulls(be, oracle):
    """nep_batch_set_hull_kernel: one hull per wave (round 1) and eight per wave (interval i on lanes 8i..8i+7, chain stacks as
    index masks) are the same algorithm; the handle picks by batch size.  Both forced on a 64-agent scene with time-shifted
    committed trajectories: hull vertices bit for bit equal to each other and to the oracle's, and so are the replans."""
    sc = scene.make_scene(64, 20, seed=7, t_jitter=0.3)
    p = sc["par"]
    outs = []
    for mode in (1, 2):
        bb = be.BatchBackend(p, sc["statics"])
        bb.set_hull_kernel(mode)
        bb.replan(bb.to_device(sc["committed"]), bb.to_device(sc["guesses"]))
        hx, hn = bb.debug_hulls(0)
        outs.append((hx.copy(), hn.copy(), bb.solutions()))
        bb.close()
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    for j in range(64):
        for i in range(p.num_pol):
            k = outs[0][1][j, i]
            np.testing.assert_array_equal(outs[0][0][j, i, :k], outs[1][0][j, i, :k])
    assert outs[0][2].tobytes() == outs[1][2].tobytes()
    t0 = float(sc["guesses"][0]["t_start"])
    for j in (0, 17, 63):
        pw = abi.nep_pwp.from_buffer_copy(sc["committed"][j]["pwp"].tobytes())
        d = np.array([sc["committed"][j]["bbox"][0] / 2 + p.drone_radius, sc["committed"][j]["bbox"][1] / 2 + p.drone_radius])
        for i in range(p.num_pol):
            h, _hu = oracle.hull_of_interval(pw, t0 + i * p.T_span, t0 + (i + 1) * p.T_span, p.T_span, d)
            assert outs[1][1][j, i] == len(h)
            np.testing.assert_array_equal(outs[1][0][j, i, :len(h)], h)


def test_short_affine_steps_discard_the_predictor(be, oracle):
    """The interior point's safeguard on the device (qp_common.h kCorrMinStep; CPU side: tests/test_frontend_cpu.py): scene 58's
    agent 5, feasible, used to cycle to the 60-iteration cap and take the relaxed solve; scene 30's agent 53, infeasible, used
    to idle to the cap.  Front-end guesses made on the device, every replan of both scenes against the oracle: same
    statuses, same iteration counts to within one, and nobody near the cap."""
    for seed, a, status in ((58, 5, 0), (30, 53, 1)):
        sc = scene.make_scene(64, 20, seed=seed); p = sc["par"]; N = p.num_agents
        fe = scene.frontend_cfg(p, beam_width=32)
        starts = scene.frontend_starts(sc)
        bb = be.BatchBackend(p, sc["statics"])
        bb.set_line_cull(0.0)                # (the safeguard is the interior point's: every row, the problem the oracle poses; the presolved path of these two hard replans: below)
        d_com = bb.to_device(sc["committed"]); d_start = bb.to_device(starts)
        d_guess = bb.torch.zeros(N * abi.GUESS_DTYPE.itemsize, dtype=bb.torch.uint8, device=bb.device)
        bb.frontend(fe, d_com, d_start, d_guess)
        bb.replan(None, d_guess)
        sol = bb.solutions(); st = sol["stats"]
        gue = d_guess.cpu().numpy().view(abi.GUESS_DTYPE)
        assert int(st["status"][a]) == status and int(st["iters_first"][a]) <= 30
        assert int(st["iters_first"].max()) <= 40
        for b in (a, (a + 7) % N, (a + 31) % N):
            if int(gue[b]["K"]) == 0:
                continue
            res = oracle.replan(p, b + 1, sc["committed"], gue[b], sc["statics"])
            assert int(st["status"][b]) == res["status"], (seed, b)
            assert abs(int(st["iters_first"][b]) - res["iters_first"]) <= 1, (seed, b, int(st["iters_first"][b]), res["iters_first"])
            K = int(gue[b]["K"])
            # (front-end guesses: the bound of the parity sweep, DESIGN.md section 2 — these are its hard cases; observed 1.1e-5)
            np.testing.assert_allclose(np.array(sol["coeff"][b])[:, :K], np.array(res["coeff"])[:, :K], atol=1e-4, rtol=0)
        bb.close()


def test_next_starts_equals_the_restatement_bit_for_bit(be):
    """nep_batch_next_starts (point A of the next bulk-synchronous round, on the device) against oracle/plan_oracle.next_start:
    inside the trajectory, on a knot, beyond its end (at rest), an invalid record (state kept), and the goal swap of an
    agent that has arrived; sharded handle (first_local > 0) and two scenes."""
    from oracle import plan_oracle as po
    N, S = 8, 2
    scs = [scene.make_scene(N, 3, seed=70 + s) for s in range(S)]
    p = scs[0]["par"]
    first, nl = 2, 4
    bb = be.BatchBackend(p, scs[0]["statics"], first_local=first, n_local=nl, n_scenes=S)
    com = np.stack([s["committed"] for s in scs]).copy()
    com[1, 3]["valid"] = 0
    starts = np.stack([scene.frontend_starts(s) for s in scs])[:, first:first + nl].copy()
    rng = np.random.default_rng(3)
    dts = [0.0, 0.25, 0.5, 1.3, 3.999, 4.0, 7.5]
    alt = rng.uniform(-5, 5, (S, nl, 3))
    # one agent sits on its goal at the end of its trajectory: it must swap goals once t passes the end
    K = int(com[0, first]["pwp"]["n_seg"]); T = p.T_span
    c = np.array(com[0, first]["pwp"]["coeff"])[:, K - 1]
    starts[0, 0]["goal"] = c[:, 0] * T ** 3 + c[:, 1] * T ** 2 + c[:, 2] * T + c[:, 3]
    d_com = bb.to_device(com); d_st = bb.to_device(starts); d_alt = bb.torch.from_numpy(alt.copy()).to(bb.device)
    want = [[{k: (list(map(float, starts[s, a][k])) if k != "t_start" else float(starts[s, a][k])) for k in ("pos", "vel", "accel", "goal", "t_start")}
             for a in range(nl)] for s in range(S)]
    want_alt = alt.copy()
    swapped = 0
    for dt in dts:
        bb.next_starts(d_com, dt, d_st, d_alt, 0.3)
        got = d_st.cpu().numpy().view(abi.FE_START_DTYPE).reshape(S, nl); got_alt = d_alt.cpu().numpy()
        for s in range(S):
            for a in range(nl):
                r = com[s, first + a]; n = int(r["pwp"]["n_seg"])
                w, al = po.next_start(list(map(float, r["pwp"]["times"][:n + 1])), np.array(r["pwp"]["coeff"])[:, :n].tolist(), bool(r["valid"]),
                                      want[s][a], dt, list(want_alt[s, a]), 0.3)
                swapped += al != list(want_alt[s, a])
                want[s][a] = w; want_alt[s, a] = al
                for k in ("pos", "vel", "accel", "goal"):
                    assert np.array(got[s, a][k]).tobytes() == np.array(w[k], dtype=np.float64).tobytes(), (dt, s, a, k)
                assert float(got[s, a]["t_start"]) == w["t_start"]
        assert got_alt.tobytes() == want_alt.tobytes()
    assert swapped >= 1
    # without the alternate goals nothing is swapped
    g0 = d_st.cpu().numpy().view(abi.FE_START_DTYPE)["goal"].copy()
    bb.next_starts(d_com, 0.5, d_st)
    assert d_st.cpu().numpy().view(abi.FE_START_DTYPE)["goal"].tobytes() == g0.tobytes()
    bb.close()


def test_redo_pass_that_fails_publishes_nothing(be):
    """Round-3 advisor finding (qp_reg_kernel.hip, the presolve's redo list).  A replan the presolve cannot verify goes on the redo
    list; its first-pass trajectory (unverified: it crosses a parked line or left the radius the skipped LPs were proven for) used
    to be written into the commit slot as valid before the redo pass ran, and stayed there when the redo's full solve FAILED and
    there were no previous records to copy back (nep_batch_replan with d_committed = NULL; the sharded nep_batch_replan_hulls).
    Forced here: a 5 cm radius sends most replans to the redo list without an iteration (their unconstrained minimiser is
    feasible for the few near rows), and a 0.1 us TimeLimit makes every solve that has to iterate fail.  A failed redo must leave
    the commit slot exactly as the caller passed it (neptune_ros.cpp:651-663: a failed replan publishes nothing)."""
    sc = scene.make_scene(64, 20, seed=3)
    p = sc["par"]
    bb = be.BatchBackend(p, sc["statics"])
    bb.set_line_cull(0.05)
    d_com = bb.to_device(sc["committed"]); d_gue = bb.to_device(sc["guesses"])
    bb.replan(d_com, d_gue)                         # hulls + a first round (everything converges: no time limit yet)
    assert bb.redo_count() > 0
    assert (bb.solutions()["stats"]["status"] != 2).all()
    bb.set_max_runtime(1e-7)
    bb.d_commit.fill_(0xAB)
    bb.replan(None, d_gue)                          # hulls reused, no previous records: prev_commit == NULL in the kernels
    listed = set(int(s) for s in bb.redo_list())
    sol = bb.solutions(); com = bb.commits()
    failed_redo = [a for a in listed if int(sol[a]["stats"]["status"]) == 2]
    assert failed_redo, "no listed replan failed its redo solve: the case this test is for did not occur"
    for a in range(p.num_agents):
        raw = np.frombuffer(com[a].tobytes(), dtype=np.uint8)
        if int(sol[a]["stats"]["status"]) == 2:
            assert (raw == 0xAB).all(), "agent %d failed (listed for redo: %s) but its commit slot was written" % (a, a in listed)
            np.testing.assert_array_equal(np.array(sol[a]["coeff"]), np.array(sc["guesses"][a]["coeff"]))      # :856-859
        else:
            assert int(com[a]["valid"]) == 1
            np.testing.assert_array_equal(np.array(com[a]["pwp"]["coeff"])[:, :8, :], np.array(sol[a]["coeff"]))
    bb.close()


def test_lp_skipping_is_off_for_statics_that_are_not_box_edged(be):
    """Round-3 advisor finding (box_far, geom_kernels.hip): the spatial presolve skips the LP of an obstacle whose bounding box is
    far, which is sound only if the box's sides are edges of the polygon (inflated statics, interval hulls).  A diamond is not:
    its nearest edge line lies at 0.71 of the box distance.  Such a handle must solve every LP: with the presolve on, every
    line exists (near or parked) and the optimum is the unculled one; with box-edged statics of the same scene LPs ARE skipped."""
    sc = scene.make_scene(16, 12, seed=21)
    p = sc["par"]
    diamonds = []
    for s in sc["statics"]:
        c = s.mean(axis=0); r = np.abs(s - c).max()
        diamonds.append(np.array([[c[0] + r, c[1]], [c[0], c[1] + r], [c[0] - r, c[1]], [c[0], c[1] - r]]))
    n_lines = {}
    for name, statics in (("squares", sc["statics"]), ("diamonds", diamonds)):
        bb = be.BatchBackend(p, statics)
        d_com = bb.to_device(sc["committed"]); d_gue = bb.to_device(sc["guesses"])
        bb.set_line_cull(0.0)
        bb.replan(d_com, d_gue)
        full = bb.solutions().copy()
        full_lines = [len(bb.debug_lines(a)[0]) for a in range(p.num_agents)]
        bb.set_line_cull(1.0)
        bb.replan(d_com, d_gue)
        cul = bb.solutions().copy()
        n_lines[name] = (sum(full_lines), sum(len(bb.debug_lines(a)[0]) for a in range(p.num_agents)))
        np.testing.assert_array_equal(cul["stats"]["status"], full["stats"]["status"])
        assert np.abs(np.array(cul["coeff"]) - np.array(full["coeff"])).max() < 1e-7
        bb.close()
    assert n_lines["diamonds"][1] == n_lines["diamonds"][0]          # nothing skipped: every line was made
    assert n_lines["squares"][1] < n_lines["squares"][0]             # box-edged statics: far LPs never solved


def test_row_scratch_is_a_pool_with_the_redo_pass(be):
    """Round-3 review (memory worst-case sized): with the presolve's redo pass the row scratch is a pool of 1 024 areas used by that
    pass only — a replan whose near lines exceed the register slots goes there unsolved.  96 agents + 20 obstacles, 16 scenes
    (1 536 slots, ~97 lines per segment): (a) the default presolve gives the unculled optimum from a pool a fraction of the
    per-slot size; (b) with a 1 km radius every line is "near": all 1 536 replans are listed, more than the pool holds — the
    overflow is flagged by nep_batch_check (NEP_E_CAP), never silent; (c) after nep_batch_reserve_row_scratch the same launch
    solves them all."""
    from neptune_amd._lib import BackendError
    from neptune_amd import dist as ndist
    S = 16
    scs = scene.make_scenes(96, 20, range(40, 40 + S), workers=16)
    p = scs[0]["par"]
    com, gue = ndist.stack_scenes(scs)

    def handle():
        b = be.BatchBackend(p, scs[0]["statics"], n_scenes=S)
        for s_ in range(S):
            b.set_scene_statics(s_, scs[s_]["statics"])
        return b
    bf = handle(); bf.set_line_cull(0.0)
    bf.replan(bf.to_device(com), bf.to_device(gue))
    full = bf.solutions().copy(); full_bytes = bf.row_scratch_bytes(); bf.close()
    bb = handle(); bb.set_line_cull(4.0)
    assert bb.row_scratch_bytes() * 1536 == full_bytes * 1024          # 1 024 areas instead of one per slot
    d_com = bb.to_device(com); d_gue = bb.to_device(gue)
    bb.replan(d_com, d_gue)
    bb.check()
    cul = bb.solutions().copy()
    np.testing.assert_array_equal(cul["stats"]["status"], full["stats"]["status"])
    assert np.abs(np.array(cul["coeff"]) - np.array(full["coeff"])).max() < 1e-6          # (two row sets, two roundings: 1.4e-7 observed)
    bb.set_line_cull(1000.0)
    bb.debug_option("presolve_kernel", 0)          # (qp_presolve_kernel would finish the replans whose unconstrained minimiser satisfies every row — whatever their row count — and
    bb.replan(d_com, d_gue)                        #  only the others would be listed: this test is about the pool, so every replan goes the interior-point kernel's way)
    assert bb.redo_count() == S * 96
    with pytest.raises(BackendError):
        bb.check()
    st = bb.solutions()["stats"]["status"]
    assert (st == 2).sum() >= S * 96 - 1024 and (st != 2).sum() >= 1000          # the listed replans beyond the pool failed, the pool's were solved
    bb.reserve_row_scratch()
    assert bb.row_scratch_bytes() == full_bytes
    bb.replan(d_com, d_gue)
    bb.check()
    sol = bb.solutions()
    np.testing.assert_array_equal(sol["stats"]["status"], full["stats"]["status"])
    assert np.abs(np.array(sol["coeff"]) - np.array(full["coeff"])).max() < 1e-6
    bb.close()


def test_line_buckets_smaller_than_the_worst_case_flag_an_overflow(be):
    """The line buckets budget 2 N entangle lines per segment instead of the worst case's 8 N (nep_batch_set_line_capacity).  A
    segment that gets more lines than its bucket holds is flagged by nep_batch_check and nothing is written past a bucket; the
    default budget gives the worst-case sizing's result bit for bit on a scene with entangle rows."""
    import dataclasses
    from neptune_amd._lib import BackendError
    sc = scene.make_scene(24, 12, seed=33)
    case = scene.synthetic_entangle(sc, seed=733, frac=0.3)
    p = dataclasses.replace(sc["par"], enable_entangle=True)
    bb = be.BatchBackend(p, sc["statics"])
    d_com = bb.to_device(sc["committed"]); d_gue = bb.to_device(sc["guesses"])
    d_ent = bb.torch.from_numpy(np.ascontiguousarray(case).reshape(-1)).to(bb.device)
    worst = 24 + 24 + 12 + 8 * 24
    assert bb.line_bucket_bytes() == 24 * 8 * (24 + 24 + 12 + 64) * 24          # default: 2 N -> max(2 N, 64) entangle lines
    for cull in (0.0, 2.0):
        bb.set_line_cull(cull)
        bb.set_line_capacity(0)
        bb.replan(d_com, d_gue, d_ent=d_ent); bb.check()
        ref = bb.solutions().copy()
        bb.set_line_capacity(-1)
        assert bb.line_bucket_bytes() == 24 * 8 * worst * 24
        bb.replan(d_com, d_gue, d_ent=d_ent); bb.check()
        assert bb.solutions().tobytes() == ref.tobytes()
        if cull == 0.0:                                            # (with the presolve the far LPs are not even solved: few lines are made)
            bb.set_line_capacity(8)                                # far too small: ~30 lines per segment
            bb.replan(d_com, d_gue, d_ent=d_ent)
            with pytest.raises(BackendError):
                bb.check()
            bb.replan(d_com, d_gue, d_ent=d_ent)                   # (sticky until read; a second launch raises it again)
            with pytest.raises(BackendError):
                bb.check()
            # ... and it fails SAFE per replan, whether or not the caller polls nep_batch_check: a replan with a segment whose lines did
            # not all fit is not solved without them (the reference poses every line, solver_gurobi_poly.cpp:473-656) — status
            # NEP_FAILED, output = the guess, and its commit slot keeps the previous record; the others are untouched
            sol = bb.solutions(); com = bb.commits()
            segs, _ = zip(*[bb.debug_lines(a) for a in range(24)])
            per_seg = np.array([[int((np.asarray(sg) == i).sum()) for i in range(8)] for sg in segs])
            failed = sol["stats"]["status"] == abi.NEP_FAILED
            assert failed.sum() > 0 and (per_seg[~failed] <= 8).all()           # (whoever was solved had every line in its buckets)
            for a in np.nonzero(failed)[0]:
                K = int(sc["guesses"][a]["K"])
                assert np.array_equal(np.array(sol[a]["coeff"])[:, :K], np.array(sc["guesses"][a]["coeff"])[:, :K])
                assert com[a].tobytes() == sc["committed"][a].tobytes()
            for a in np.nonzero(~failed)[0]:
                assert sol[a].tobytes() == ref[a].tobytes()
    bb.close()


def test_a_hostile_environment_changes_nothing(be):
    """Round-5 review: the drop-in's numerics must not depend on the caller's environment.  The same replan (24 agents + 12 obstacles,
    entangle rows on; then a front-end search) in two child processes — a clean environment, and one with every variable the library
    used to read (rounds 2-5) set to the value that used to change its behaviour most: the two print the same digests."""
    import hashlib, os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import dataclasses, hashlib, sys
        import numpy as np
        sys.path.insert(0, %r)
        from neptune_amd import abi, scene
        from neptune_amd.backend import BatchBackend
        sc = scene.make_scene(24, 12, seed=33)
        case = scene.synthetic_entangle(sc, seed=733, frac=0.3)
        p = dataclasses.replace(sc["par"], enable_entangle=True)
        bb = BatchBackend(p, sc["statics"])
        d_ent = bb.torch.from_numpy(np.ascontiguousarray(case).reshape(-1)).to(bb.device)
        d_com = bb.to_device(sc["committed"]); d_g = bb.to_device(sc["guesses"])
        bb.replan(d_com, d_g, d_ent=d_ent)
        h = hashlib.sha256(bb.solutions().tobytes())
        for a in range(24):
            seg, nd = bb.debug_lines(a, cap=8192)
            h.update(seg.tobytes()); h.update(nd.tobytes())
        h.update(np.array([bb.line_cull(), float(bb.qp_kernel_name() == "qp_reg_kernel"), float(bb.redo_count())]).tobytes())
        d_gfe = bb.torch.zeros(24 * abi.GUESS_DTYPE.itemsize, dtype=bb.torch.uint8, device=bb.device)
        p0 = sc["par"]; b0 = BatchBackend(p0, sc["statics"])
        b0.frontend(scene.frontend_cfg(p0, beam_width=16), b0.to_device(sc["committed"]), b0.to_device(scene.frontend_starts(sc)), d_gfe, None)
        b0.replan(None, d_gfe)
        h.update(d_gfe.cpu().numpy().tobytes()); h.update(b0.solutions().tobytes())
        print("DIGEST", h.hexdigest())
    """ % helpers.ROOT)
    hostile = {"NEP_QP_KERNEL": "lds", "NEP_QP_AUTOCULL": "0", "NEP_SEP_SKIP": "0", "NEP_SEP_NO_REDO": "1", "NEP_QP_LPT": "0", "NEP_FE_LPT": "0",
               "NEP_QP_KEY_DECAY": "0", "NEP_FE_KEY_DECAY": "0", "NEP_SEP_UNPACKED": "1", "NEP_SEP_PACK": "3", "NEP_CORR_FROM": "2", "NEP_CORR_MAX": "1",
               "NEP_HULL_KERNEL": "wave", "NEP_FE_THREE": "1", "NEP_FE_XCD": "0", "NEP_POLISH_GRID": "1"}
    outs = []
    for extra in ({}, hostile):
        env = {k: v for k, v in os.environ.items() if not (k.startswith("NEP_") and k != "NEP_BACKEND_LIB")}
        env.update(extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("DIGEST")][0])
    assert outs[0] == outs[1]
