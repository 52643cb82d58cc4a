"""GPU parity tests of the rows either side of the path: gjk::collision, the front-end beam search bit for bit against
the oracle, the closed loop flown to its goals, safetyCheckAfterReplan + commit."""
import numpy as np
import pytest

import helpers
from neptune_amd import abi, scene
from gpu_util import COEF_TOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from neptune_amd import backend
    return backend


def test_gjk_batch_matches_the_oracle(be, oracle):
    """gjk::collision on the device (safety check, front end) against the restatement: identical verdicts
    on control polygons scattered around real interval hulls and inflated statics."""
    sc = scene.make_scene(8, 6, seed=5)
    p = sc["par"]
    hx, hn = oracle.hulls_of_scene(p, 1, sc["committed"], 0.0, sc["statics"])
    rng = np.random.default_rng(0)
    polys, quads = [], []
    shapes = [hx[j, i, :hn[j, i]] for j in range(1, 8) for i in range(8) if hn[j, i] > 0] + [np.asarray(s) for s in sc["statics"]]
    for V in shapes:
        c = V.mean(axis=0)
        for _ in range(150):
            polys.append(V)
            quads.append(c + rng.normal(scale=1.5, size=2) + rng.normal(scale=0.6, size=(4, 2)).cumsum(axis=0))
    got = be.gjk_batch(polys, np.array(quads))
    want = np.array([oracle.gjk_collision(P, Q) for P, Q in zip(polys, quads)])
    assert 0.2 < want.mean() < 0.9
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("n_agents,n_static,seed,W", [(8, 6, 5, 32), (8, 10, 9, 64), (5, 0, 3, 1), (16, 8, 4, 16)])
def test_frontend_beam_matches_the_oracle_bit_for_bit(be, oracle, n_agents, n_static, seed, W):
    """the front-end kernel against the deterministic beam rule of the oracle — the
    guesses (lattice primitives) must be identical, then the back end runs on the device-made guesses."""
    sc = scene.make_scene(n_agents, n_static, seed=seed)
    p = sc["par"]; N = p.num_agents
    fe = scene.frontend_cfg(p, beam_width=W)
    starts = scene.frontend_starts(sc)
    bb = be.BatchBackend(p, sc["statics"])
    d_com = bb.to_device(sc["committed"])
    d_start = bb.to_device(starts)
    d_guess = bb.torch.zeros(N * abi.GUESS_DTYPE.itemsize, dtype=bb.torch.uint8, device=bb.device)
    d_res = bb.torch.zeros(N * abi.FE_RESULT_DTYPE.itemsize, dtype=bb.torch.uint8, device=bb.device)
    bb.frontend(fe, d_com, d_start, d_guess, d_res)
    bb.torch.cuda.synchronize()
    got_g = d_guess.cpu().numpy().view(abi.GUESS_DTYPE)
    got_r = d_res.cpu().numpy().view(abi.FE_RESULT_DTYPE)
    n_ok = 0
    for a in range(N):
        hx, hn = oracle.hulls_of_scene(p, a + 1, sc["committed"], float(starts[a]["t_start"]), sc["statics"])
        g, r = oracle.frontend_beam(p, fe, a + 1, starts[a], hx, hn, sc["statics"])
        for f in abi.FE_RESULT_DTYPE.names:
            assert got_r[a][f] == r[f], (a, f, got_r[a][f], r[f])
        assert int(got_g[a]["K"]) == int(g["K"]) and got_g[a]["t_start"] == g["t_start"]
        np.testing.assert_array_equal(got_g[a]["coeff"], g["coeff"])
        n_ok += int(g["K"]) > 0
    assert n_ok >= N - 1
    # pad_hold: short guesses extended with segments holding their end point — same on both sides
    fe_pad = scene.frontend_cfg(p, beam_width=W, pad_hold=1)
    starts_near = starts.copy()
    starts_near["goal"][:, :2] = starts_near["pos"][:, :2] + [0.9, 0.3]          # goals one or two segments away: short searches
    d_g2 = bb.torch.zeros_like(d_guess)
    bb.frontend(fe_pad, d_com, bb.to_device(starts_near), d_g2)
    bb.torch.cuda.synchronize()
    got2 = d_g2.cpu().numpy().view(abi.GUESS_DTYPE)
    n_short = 0
    for a in range(N):
        hx, hn = oracle.hulls_of_scene(p, a + 1, sc["committed"], float(starts[a]["t_start"]), sc["statics"])
        g, r = oracle.frontend_beam(p, fe_pad, a + 1, starts_near[a], hx, hn, sc["statics"])
        assert int(got2[a]["K"]) == int(g["K"])
        np.testing.assert_array_equal(got2[a]["coeff"], g["coeff"])
        if 0 < r["K"] < p.num_pol:
            n_short += 1
            assert int(g["K"]) == p.num_pol and (np.array(g["coeff"])[:2, r["K"]:, :3] == 0).all()
    assert n_short >= 1
    # the back end on the device-made guesses
    bb.replan(d_com, d_guess)
    sol = bb.solutions()
    for a in range(N):
        K = int(got_g[a]["K"])
        if K == 0:
            continue
        r = oracle.replan(p, a + 1, sc["committed"], got_g[a], sc["statics"])
        assert int(sol[a]["stats"]["status"]) == r["status"], a
        assert np.abs(np.array(sol[a]["coeff"])[:, :K, :] - r["coeff"]).max() <= COEF_TOL
    bb.close()


def test_frontend_multi_scene_and_agent_shard(be, oracle):
    """The front end with several scenes per launch and on an agent shard (first_local > 0), different
    t_start per scene: slot -> (scene, agent) bookkeeping, hull interval grid and own-hull skipping."""
    from neptune_amd import dist as ndist
    S, N = 3, 6
    scenes = [scene.make_scene(N, 5, seed=80 + s) for s in range(S)]
    for k, sc in enumerate(scenes):
        sc["statics"] = scenes[0]["statics"]
        sc["guesses"]["t_start"] += 0.35 * k                     # every scene on its own clock
        sc["committed"]["pwp"]["times"] += 0.35 * k
    p = scenes[0]["par"]
    com, _ = ndist.stack_scenes(scenes)
    fe = scene.frontend_cfg(p, beam_width=24)
    starts = np.stack([scene.frontend_starts(sc) for sc in scenes])          # [S][N]
    rng = np.random.default_rng(8)                                            # height states and goals: the z profile (getInitialZPwp)
    starts["pos"][:, :, 2] = rng.uniform(0.5, 3.0, size=(S, N)); starts["vel"][:, :, 2] = rng.normal(scale=1.5, size=(S, N))
    starts["accel"][:, :, 2] = rng.normal(scale=2.5, size=(S, N)); starts["goal"][:, :, 2] = rng.uniform(0.5, 4.5, size=(S, N))
    for first, nl in ((0, 6), (2, 2), (3, 3)):
        bb = be.BatchBackend(p, scenes[0]["statics"], first_local=first, n_local=nl, n_scenes=S)
        d_start = bb.to_device(np.ascontiguousarray(starts[:, first:first + nl]))
        d_guess = bb.torch.zeros(S * nl * abi.GUESS_DTYPE.itemsize, dtype=bb.torch.uint8, device=bb.device)
        d_res = bb.torch.zeros(S * nl * abi.FE_RESULT_DTYPE.itemsize, dtype=bb.torch.uint8, device=bb.device)
        bb.frontend(fe, bb.to_device(com), d_start, d_guess, d_res)
        bb.torch.cuda.synchronize()
        got_g = d_guess.cpu().numpy().view(abi.GUESS_DTYPE).reshape(S, nl)
        got_r = d_res.cpu().numpy().view(abi.FE_RESULT_DTYPE).reshape(S, nl)
        for s_ in range(S):
            for al in range(nl):
                a = first + al
                hx, hn = oracle.hulls_of_scene(p, a + 1, com[s_], float(starts[s_, a]["t_start"]), scenes[0]["statics"])
                g, r = oracle.frontend_beam(p, fe, a + 1, starts[s_, a], hx, hn, scenes[0]["statics"])
                assert int(got_r[s_, al]["status"]) == r["status"] and int(got_r[s_, al]["n_collision_free"]) == r["n_collision_free"], (first, s_, a)
                np.testing.assert_array_equal(got_g[s_, al]["coeff"], g["coeff"])
                assert got_g[s_, al]["t_start"] == g["t_start"] and int(got_g[s_, al]["K"]) == int(g["K"])
        bb.close()


@pytest.mark.parametrize("n_agents,n_static,seed,W", [(5, 0, 3, 64), (5, 0, 4, 8), (64, 20, 2, 48), (24, 12, 9, 33)])
def test_frontend_beam_widths_and_small_scenes(be, oracle, n_agents, n_static, seed, W):
    """The front end's LDS carve depends on the beam's width (per-rank arrays at 32 or 64) and on the scene's size (the winners'
    f values live in the shortlist's storage when that is big enough; four or three workgroups per CU): widths either side of
    32, the widest, and a scene too small for the aliasing — guesses, cost and status against the oracle, bit for bit."""
    sc = scene.make_scene(n_agents, n_static, seed=seed)
    p = sc["par"]; N = n_agents
    fe = scene.frontend_cfg(p, beam_width=W)
    starts = scene.frontend_starts(sc)
    bb = be.BatchBackend(p, sc["statics"])
    d_guess = bb.torch.zeros(N * abi.GUESS_DTYPE.itemsize, dtype=bb.torch.uint8, device=bb.device)
    d_res = bb.torch.zeros(N * abi.FE_RESULT_DTYPE.itemsize, dtype=bb.torch.uint8, device=bb.device)
    bb.frontend(fe, bb.to_device(sc["committed"]), bb.to_device(starts), d_guess, d_res)
    bb.torch.cuda.synchronize()
    got_g = d_guess.cpu().numpy().view(abi.GUESS_DTYPE); got_r = d_res.cpu().numpy().view(abi.FE_RESULT_DTYPE)
    for a in range(0, N, max(1, N // 8)):
        hx, hn = oracle.hulls_of_scene(p, a + 1, sc["committed"], float(starts[a]["t_start"]), sc["statics"])
        g, r = oracle.frontend_beam(p, fe, a + 1, starts[a], hx, hn, sc["statics"])
        for f in ("status", "K", "n_children", "n_feasible", "n_collision_free"):
            assert int(got_r[a][f]) == r[f], (a, f)
        assert float(got_r[a]["cost"]) == r["cost"]
        np.testing.assert_array_equal(np.array(got_g[a]["coeff"]), np.array(g["coeff"]), err_msg="agent %d" % a)
    bb.close()


@pytest.mark.parametrize("n_agents,n_static,seed,min_reached", [(8, 6, 5, 8), (16, 8, 1, 16), (64, 20, 0, 64)])
def test_closed_loop_fleet_flies_to_its_goals_without_collisions(be, n_agents, n_static, seed, min_reached):
    """Everything together (neptune_amd/loop.py): point A from the plan deque -> front-end guess -> separating
    lines + QP -> safety check -> plan splice + composition -> tracker, in bulk-synchronous rounds until
    the fleet has arrived.  The planner's contract: centres never closer than the inflation it plans with."""
    from neptune_amd.loop import FleetLoop
    sc = scene.make_scene(n_agents, n_static, seed=seed)
    p = sc["par"]
    loop = FleetLoop(p, sc["statics"], sc["starts"], scene.reachable_goals(sc), beam_width=32)
    st = loop.run(max_rounds=400)
    loop.close()
    assert st["reached"] >= min_reached, st
    assert st["min_pair_dist"] >= 2 * p.drone_radius, st                 # the inflation the planner works with (bbox/2 + drone_radius)
    assert st["min_static_dist"] >= 2 * p.drone_radius + 0.2 - 0.02, st   # inflation of the static obstacles (neptune.cpp:642)
    assert st["accepted"] > 0.8 * st["replans"] and st["qp_failed"] < 0.01 * st["replans"], st


def test_safety_check_and_commit(be, oracle):
    """conflict matrix (GJK on the new trajectories' hulls), id-ordered
    resolution and the committed records, bit for bit against the oracle."""
    scenes = [scene.make_scene(8, 0, seed=60 + s) for s in range(2)]
    p = scenes[0]["par"]
    prev = np.stack([s["committed"] for s in scenes])
    fresh = prev.copy()
    # scene 0: agent 6 and 8 fly copies of agent 2's trajectory next to it; scene 1 untouched
    for tgt, dx in ((5, 0.5), (7, -0.6)):
        fresh[0, tgt] = fresh[0, 1]; fresh[0, tgt]["id"] = tgt + 1
        fresh[0, tgt]["pwp"]["coeff"][0, :, 3] += dx
    fresh["pos"][:] += 0.01                                      # make new != prev everywhere
    gue = np.stack([s["guesses"] for s in scenes])
    bb = be.BatchBackend(p, [], n_scenes=2)
    d_prev = bb.to_device(prev); d_new = bb.to_device(fresh); d_gue = bb.to_device(gue)
    d_final = bb.torch.zeros_like(d_prev); d_acc = bb.torch.zeros(2 * 8, dtype=bb.torch.int32, device=bb.device)
    bb.safety_commit(d_prev, d_new, d_gue, d_final, d_acc)
    acc = d_acc.cpu().numpy().reshape(2, 8)
    fin = d_final.cpu().numpy().view(abi.TRAJ_REC_DTYPE).reshape(2, 8)
    for s_ in range(2):
        conflict, accept = oracle.safety_resolve(fresh[s_], 0.0, p.T_span, p.drone_radius)
        np.testing.assert_array_equal(bb.debug_conflicts(s_), conflict)
        np.testing.assert_array_equal(acc[s_], accept)
        for a in range(8):
            want = fresh[s_, a] if accept[a] else prev[s_, a]
            assert fin[s_, a].tobytes() == want.tobytes()
    assert list(acc[0]) == [1, 1, 1, 1, 1, 0, 1, 0] and acc[1].all()
    # with the previous-record check: scene 1's agent 5 now flies along agent 3's PREVIOUS path while agent 3's new
    # trajectory is far away — no new-new conflict, but agent 3 might be turned down and keep that previous path
    fresh[1, 2]["pwp"]["coeff"][0, :, 3] += 40.0
    fresh[1, 4] = prev[1, 2]; fresh[1, 4]["id"] = 5
    fresh[1, 4]["pwp"]["coeff"][0, :, 3] += 0.4
    bb.set_safety_check_prev(True)
    d_new = bb.to_device(fresh)
    bb.safety_commit(d_prev, d_new, d_gue, d_final, d_acc)
    acc = d_acc.cpu().numpy().reshape(2, 8)
    for s_ in range(2):
        conflict, accept = oracle.safety_resolve_prev(prev[s_], fresh[s_], 0.0, p.T_span, p.drone_radius)
        np.testing.assert_array_equal(bb.debug_conflicts(s_), conflict)
        np.testing.assert_array_equal(acc[s_], accept)
    assert acc[1, 4] == 0 and acc[1, 2] == 1
    _, plain = oracle.safety_resolve(fresh[1], 0.0, p.T_span, p.drone_radius)
    assert plain[4] == 1                                          # the plain pass would have let it through
    bb.close()
