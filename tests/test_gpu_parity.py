"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle and the golden fixtures — hulls, separating lines
(both rules), the golden QPs, every BASELINE configuration's replan, sharded and multi-scene batches.  Bars, as asserted below: hull
vertices and separating lines BIT-EXACT; on the scenes' own guesses and on the golden QPs coefficients within 1e-6 (gpu_util.COEF_TOL,
absolute: metres and polynomial coefficients) and cost within 1e-6 relative (COST_RTOL; the north star asks for 1e-4 on cost).
The other GPU parity files: test_gpu_per_agent_api.py (the drop-in handle), test_gpu_entangle_config5.py (entangle rows, config 5),
test_gpu_presolve_polish.py (presolve, polish pass, front-end guesses, hard cases), test_gpu_frontend_safety.py (rows f1/f2)."""
import numpy as np
import pytest

import helpers
from neptune_amd import abi, scene
from gpu_util import _solver, _check_scene, solver_lines_match, COEF_TOL, COST_RTOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from neptune_amd import backend
    return backend


def test_hulls_bit_exact(be, oracle):
    for seed, jit in ((0, 0.0), (1, 0.37), (2, 0.2)):
        sc = scene.make_scene(8, 0, seed=seed, t_jitter=0.0)
        p = sc["par"]
        t_start = jit
        hx, hn, h0, n0 = be.hulls_batch(sc["committed"], t_start, p.num_pol, p.T_span, p.drone_radius)
        for j in range(8):
            pw = abi.nep_pwp.from_buffer_copy(sc["committed"][j]["pwp"].tobytes())
            d = np.array([sc["committed"][j]["bbox"][0] / 2 + p.drone_radius, sc["committed"][j]["bbox"][1] / 2 + p.drone_radius])
            for i in range(p.num_pol):
                h, hu = oracle.hull_of_interval(pw, t_start + i * p.T_span, t_start + (i + 1) * p.T_span, p.T_span, d)
                assert hn[j, i] == len(h) and n0[j, i] == len(hu)
                np.testing.assert_array_equal(hx[j, i, :len(h)], h)
                np.testing.assert_array_equal(h0[j, i, :len(hu)], hu)


def test_separator_bit_exact_on_golden_lps(be, oracle):
    d = np.load(helpers.ROOT + "/tests/golden/lp_cases.npz")
    As = [A[~np.isnan(A[:, 0])] for A in d["A"]]; Bs = list(d["B"])
    ok, nd = be.separator_batch(As, Bs)
    n_ok = 0
    for k, (A, B) in enumerate(zip(As, Bs)):
        o, n = oracle.separator(A, B)
        assert bool(ok[k]) == o
        np.testing.assert_array_equal(nd[k], n)
        if o:
            assert (A @ nd[k, :2] + nd[k, 2]).min() >= 1 - 1e-9 and (B @ nd[k, :2] + nd[k, 2]).max() <= -1 + 1e-9
            n_ok += 1
        if bool(d["feasible"][k]) != o:
            assert not o
    assert n_ok > 200


def test_separator_rule_glpk_class_bit_exact(be, oracle):
    """nep_separator_batch_rule(1): the vertex a primal simplex of GLPK's default class reaches — device == oracle bit for bit
    on the golden LPs (feasibility as HiGHS) and on random point sets of every size the path poses."""
    d = np.load(helpers.ROOT + "/tests/golden/lp_cases.npz")
    As = [A[~np.isnan(A[:, 0])] for A in d["A"]]; Bs = list(d["B"])
    rng = np.random.default_rng(11)
    for k in range(400):                       # two clouds a random distance apart (some overlap: infeasible), 1..16 against 4 points
        nA = int(rng.integers(1, 17))
        c = rng.uniform(-3, 3, 2)
        As.append(rng.normal(size=(nA, 2)) * rng.uniform(0.1, 1.5)); Bs.append(c + rng.normal(size=(4, 2)) * rng.uniform(0.05, 1.0))
    As.append(np.zeros((3, 2))); Bs.append(np.ones((4, 2)))                          # coincident points
    ok, nd = be.separator_batch(As, Bs, rule=1)
    n_ok = 0
    for k in range(len(As)):
        o, n, _ = oracle.separator_glpk_class(As[k], Bs[k])
        assert bool(ok[k]) == o, k
        assert nd[k].tobytes() == n.tobytes(), (k, nd[k], n)
        n_ok += o
    assert np.array_equal(ok[:300], d["feasible"].astype(bool))
    assert n_ok > 300
    ok0, nd0 = be.separator_batch(As[:300], Bs[:300])                                # (rule 0 is what nep_separator_batch runs)
    assert np.array_equal(ok0, ok[:300]) or (ok0 <= ok[:300]).all()
    assert not np.array_equal(nd0, nd[:300])


def test_replan_with_the_glpk_class_separator_rule(be, oracle):
    """nep_batch_set_separator_rule(1): every line of a replan comes from the simplex rule — bit-identical to the oracle
    running the same rule — and the QP on those lines agrees as for the default rule; through the per-agent handle too."""
    sc = scene.make_scene(8, 20, seed=3)
    p = sc["par"]
    oracle.set_separator_rule(1)
    try:
        bb = be.BatchBackend(p, sc["statics"])
        bb.set_separator_rule(1)
        bb.set_line_cull(0.0)                   # (lines compared in the reference's call order)
        bb.replan(bb.to_device(sc["committed"]), bb.to_device(sc["guesses"]))
        sol = bb.solutions()
        n_diff = 0
        for a in range(p.num_agents):
            r = oracle.replan(p, a + 1, sc["committed"], sc["guesses"][a], sc["statics"])
            seg, nd = bb.debug_lines(a)
            np.testing.assert_array_equal(seg, r["line_seg"]); np.testing.assert_array_equal(nd, r["line_nd"])
            st = sol[a]["stats"]; K = int(sol[a]["K"])
            assert int(st["status"]) == r["status"] and int(st["n_lp"]) == r["n_lp"] and int(st["n_lp_failed"]) == r["n_lp_failed"]
            assert np.abs(np.array(sol[a]["coeff"])[:, :K, :] - r["coeff"]).max() <= COEF_TOL
            oracle.set_separator_rule(0)
            r0 = oracle.replan(p, a + 1, sc["committed"], sc["guesses"][a], sc["statics"])
            oracle.set_separator_rule(1)
            assert r0["n_lp"] == r["n_lp"] and r0["n_lp_failed"] == r["n_lp_failed"]           # same LPs, same feasibility
            n_diff += not np.array_equal(r0["line_nd"], r["line_nd"])
        assert n_diff > 0
        bb.set_separator_rule(0)
        bb.close()
        # per-agent handle
        aid = 3
        hx, hn, h0, n0 = be.hulls_batch(sc["committed"], 0.0, p.num_pol, p.T_span, p.drone_radius)
        others = [j for j in range(p.num_agents) if j != aid - 1]
        s = _solver(be, p, aid)
        s.setSeparatorRule(1)
        s.setStaticObstVert(sc["statics"])
        g = sc["guesses"][aid - 1]; K = int(g["K"])
        s.setInitTrajectory(np.arange(K + 1) * p.T_span, np.array(g["coeff"])[:, :K, :])
        s.setHulls([[hx[j, i, :hn[j, i]] for i in range(p.num_pol)] for j in others])
        s.optimize()
        r = oracle.replan(p, aid, sc["committed"], g, sc["statics"])
        solver_lines_match(s, r)
        s.close()
    finally:
        oracle.set_separator_rule(0)


def test_separator_edge_cases(be, oracle):
    sq = np.array([[1.0, 1.0], [1.0, -1.0], [-1.0, 1.0], [-1.0, -1.0]])
    cases = [(sq, np.tile([[3.0, 0.5]], (4, 1))),                       # hovering agent (coincident control points)
             (sq, np.array([[0.0, 0.0], [0.1, 0.1], [0.2, 0.0], [0.1, -0.1]])),   # inside: infeasible
             (np.array([[0.0, 0.0]]), np.tile([[2.0, 2.0]], (4, 1))),    # point vs point
             (np.array([[0.0, 0.0], [1.0, 0.0]]), np.array([[0.0, 2.0], [1.0, 2.0], [2.0, 3.0], [0.5, 4.0]])),
             (sq, sq + np.array([2.0 + 1e-9, 0.0]))]                    # gap below the floor
    ok, nd = be.separator_batch([c[0] for c in cases], [c[1] for c in cases])
    for k, (A, B) in enumerate(cases):
        o, n = oracle.separator(A, B)
        assert bool(ok[k]) == o
        np.testing.assert_array_equal(nd[k], n)
    assert list(ok) == [True, False, True, True, False]


@pytest.mark.parametrize("fixture", ["qp_cases.npz", "qp_cases_r2.npz"])
def test_qp_against_golden(be, oracle, fixture):
    """qp_cases_r2.npz: BASELINE config-4 size (~510 lines, LDS placement), config-5 size (~2 050 lines: the global-spill
    placement of the row state), K = 7, front-end guesses — each a KKT-certified optimum."""
    seen = set()
    for c in helpers.load_qp_cases(fixture):
        p = helpers.params_of_case(c)
        s = _solver(be, p)
        K = c["K"]
        s.setInitTrajectory(np.arange(K + 1) * p.T_span, c["coeff_init"])
        s.debugSetLines(c["line_seg"], c["line_nd"])
        ok, obj = s.optimize()
        st = s.stats()
        assert st["status"] == c["status"], c["tag"]
        seen.add(st["status"])
        times, coeff, traj = s.generatePwpOut(0.0, p.dc)
        th = helpers.golden_theta_out(c)
        assert np.abs(coeff - th).max() <= max(COEF_TOL, helpers.theta_tol(c)), (c["tag"], np.abs(coeff - th).max())
        r = oracle.optimize(p, 1, c["coeff_init"], [], [], lines=(c["line_seg"], c["line_nd"]))
        assert np.abs(coeff - r["coeff"]).max() <= COEF_TOL, (c["tag"], np.abs(coeff - r["coeff"]).max())
        if c["status"] != 2:
            assert ok and abs(obj - r["objective"]) <= COST_RTOL * (1 + abs(r["objective"])), c["tag"]
            assert abs(obj - c["cost"]) <= 1e-5 * (1 + abs(c["cost"])), c["tag"]
        else:
            assert not ok and obj is None
            np.testing.assert_array_equal(coeff, c["coeff_init"])      # output == initial guess
        # generatePwpOut's samples of the GPU coefficients
        ref = oracle.sample(coeff, p.T_span, p.dc)
        assert len(ref) == len(traj)
        np.testing.assert_allclose(traj, ref, rtol=0, atol=1e-12)
        s.close()
    assert seen == ({0, 1, 2} if fixture == "qp_cases.npz" else seen) and 0 in seen


def test_replan_config2_five_agents(be, oracle):
    for seed in (0, 1, 2):
        _check_scene(be, oracle, scene.make_scene(5, 0, seed=seed))


def test_replan_config3_eight_agents_twenty_obstacles(be, oracle):
    for seed in (0, 3):
        _check_scene(be, oracle, scene.make_scene(8, 20, seed=seed))


def test_replan_config1_single_agent(be, oracle):
    _check_scene(be, oracle, scene.make_scene(1, 0, seed=0, K=3))


def test_replan_short_guesses(be, oracle):
    for K in (1, 2, 4, 6):
        _check_scene(be, oracle, scene.make_scene(3, 4, seed=20 + K, K=K))


def test_replan_sharded_slice(be, oracle):
    """A rank that owns agents [4, 8) of an 8-agent scene produces what the full run produces."""
    _check_scene(be, oracle, scene.make_scene(8, 20, seed=5), first_local=4, n_local=4)


def test_full_size_properties_64_agents(be):
    """BASELINE config 4 size on one GPU (64 agents + 20 obstacles): size-independent properties
    instead of the oracle: every line separates, the solution is feasible and C2, initial state is
    kept, and the cost does not exceed the cost of any feasible guess."""
    sc = scene.make_scene(64, 20, seed=0)
    p = sc["par"]
    bb = be.BatchBackend(p, sc["statics"])
    bb.replan(bb.to_device(sc["committed"]), bb.to_device(sc["guesses"]))
    sol = bb.solutions()
    T = p.T_span
    M4 = scene.A_POS_INV * np.array([T ** 3, T ** 2, T, 1.0])[:, None]
    n_ok = 0
    for a in range(64):
        g = sc["guesses"][a]; K = int(g["K"]); ci = np.array(g["coeff"])[:, :K, :]
        st = sol[a]["stats"]; co = np.array(sol[a]["coeff"])[:, :K, :]
        # a failed LP (guess passing over a base square) silently drops its constraint (solver_gurobi_poly.cpp:491-494)
        assert int(st["n_lp_failed"]) <= 0.05 * int(st["n_lp"])
        if int(st["status"]) == 2:
            continue
        n_ok += 1
        np.testing.assert_allclose(co[:, 0, 1:], ci[:, 0, 1:], atol=1e-9)          # initial p, v, a
        tp = np.array([T ** 3, T ** 2, T, 1.0]); tv = np.array([3 * T * T, 2 * T, 1.0, 0]); ta = np.array([6 * T, 2.0, 0, 0])
        for i in range(K - 1):                                                      # C2 continuity
            assert np.abs(co[:2, i] @ tp - co[:2, i + 1, 3]).max() < 1e-8
            assert np.abs(co[:2, i] @ tv - co[:2, i + 1, 2]).max() < 1e-8
            assert np.abs(co[:2, i] @ ta - 2 * co[:2, i + 1, 1]).max() < 1e-8
        if int(st["status"]) == 0:
            assert np.abs(co[:2, K - 1] @ tv).max() < 1e-8 and np.abs(co[:2, K - 1] @ ta).max() < 1e-8   # terminal rest
        seg, nd = bb.debug_lines(a)
        cpx = co[0] @ M4; cpy = co[1] @ M4
        for s_, l in zip(seg, nd):
            assert (l[0] * cpx[s_] + l[1] * cpy[s_] + l[2] - 1).max() <= 1e-7
        assert cpx.min() >= p.x_min - 1e-7 and cpx.max() <= p.x_max + 1e-7
    assert n_ok >= 60
    bb.close()


def test_multi_scene_batch_and_chained_rounds(be, oracle):
    """Several scenes per launch (slot = scene*n_local + agent) and two chained rounds: round 2
    replans against the records committed by round 1, exactly as the oracle does on the host."""
    from neptune_amd import dist as ndist
    scenes = [scene.make_scene(6, 5, seed=40 + s) for s in range(3)]
    for sc in scenes[1:]:
        sc["statics"] = scenes[0]["statics"]          # one static set per handle
    p = scenes[0]["par"]
    com, gue = ndist.stack_scenes(scenes)
    bb = be.BatchBackend(p, scenes[0]["statics"], n_scenes=3)
    d_com = bb.to_device(com); d_gue = bb.to_device(gue)
    ex = ndist.RoundExchange(3, 6, 1, 0, device=bb.device)
    host_com = com.copy()
    for rnd in range(2):
        bb.replan(d_com, d_gue)
        sol = bb.solutions().reshape(3, 6)
        new_com = host_com.copy()
        for s_ in range(3):
            for a in range(6):
                r = oracle.replan(p, a + 1, host_com[s_], gue[s_, a], scenes[0]["statics"])
                K = int(sol[s_, a]["K"])
                co = np.array(sol[s_, a]["coeff"])[:, :K, :]
                assert int(sol[s_, a]["stats"]["status"]) == r["status"], (rnd, s_, a)
                assert int(sol[s_, a]["stats"]["n_lines"]) == r["n_lines"]
                assert np.abs(co - r["coeff"]).max() <= COEF_TOL, (rnd, s_, a)
                # what the oracle side commits: the new trajectory with absolute knot times
                rec = new_com[s_, a]
                rec["pwp"]["coeff"][:, :K, :] = r["coeff"]; rec["pwp"]["n_seg"] = K
                rec["pos"] = r["coeff"][:, 0, 3]
        ex.gather(bb.d_commit, d_com)                   # device side: committed <- this round's records
        dev_com = d_com.cpu().numpy().view(abi.TRAJ_REC_DTYPE).reshape(3, 6)
        assert np.abs(np.array(dev_com["pwp"]["coeff"]) - np.array(new_com["pwp"]["coeff"])).max() <= COEF_TOL
        np.testing.assert_array_equal(dev_com["id"], new_com["id"])
        np.testing.assert_allclose(dev_com["pwp"]["times"], new_com["pwp"]["times"], atol=1e-12)
        host_com = dev_com.copy()                        # keep both sides on identical inputs for round 2
    bb.close()


@pytest.mark.parametrize("world,entangle", [(2, False), (4, False), (4, True)])
def test_sharded_hull_blocks_equal_the_single_rank_replan(be, world, entangle):
    """Multi-GPU layout on one GPU: every "rank" computes the hull block of its own agents, the
    blocks are concatenated as the all-gather would, and separator + QP against the blocks must
    give the single-rank nep_batch_replan results bit for bit (same kernels, same inputs)."""
    import dataclasses
    from neptune_amd import dist as ndist
    N, S = 8, 3
    scenes = [scene.make_scene(N, 6, seed=70 + s) for s in range(S)]
    for sc in scenes[1:]:
        sc["statics"] = scenes[0]["statics"]
    p = dataclasses.replace(scenes[0]["par"], enable_entangle=entangle)
    com, gue = ndist.stack_scenes(scenes)
    case = np.stack([scene.synthetic_entangle(sc, seed=9, frac=0.5) for sc in scenes]) if entangle else None   # [S][N][8][N]
    full = be.BatchBackend(p, scenes[0]["statics"], n_scenes=S)
    d_ent = full.torch.from_numpy(case.reshape(-1).copy()).to(full.device) if entangle else None
    full.replan(full.to_device(com), full.to_device(gue), d_ent=d_ent)
    want = full.solutions().reshape(S, N)
    want_commit = full.commits().reshape(S, N)
    nl = N // world
    ranks = [be.BatchBackend(p, scenes[0]["statics"], first_local=r * nl, n_local=nl, n_scenes=S) for r in range(world)]
    bb = ranks[0].hull_block_bytes()
    assert all(r.hull_block_bytes() == bb for r in ranks) and bb % 256 == 0
    blocks = full.torch.zeros(world * bb, dtype=full.torch.uint8, device=full.device)
    g_loc = [ranks[r].to_device(np.ascontiguousarray(gue[:, r * nl:(r + 1) * nl])) for r in range(world)]
    for r in range(world):
        ranks[r].hulls(ranks[r].to_device(np.ascontiguousarray(com[:, r * nl:(r + 1) * nl])), g_loc[r], blocks[r * bb:(r + 1) * bb])
    n_lines = 0
    for r in range(world):
        e = None
        if entangle:
            e = full.torch.from_numpy(np.ascontiguousarray(case[:, r * nl:(r + 1) * nl]).reshape(-1).copy()).to(full.device)
        ranks[r].replan_hulls(blocks, g_loc[r], d_ent=e)
        got = ranks[r].solutions().reshape(S, nl)
        ref = want[:, r * nl:(r + 1) * nl]
        np.testing.assert_array_equal(got["coeff"], ref["coeff"])
        np.testing.assert_array_equal(got["times"], ref["times"])
        for f in ("status", "iters", "n_lines", "n_lp", "n_lp_failed", "n_rows", "objective"):
            np.testing.assert_array_equal(got["stats"][f], ref["stats"][f], err_msg=f)
        gc = ranks[r].commits().reshape(S, nl)
        np.testing.assert_array_equal(gc["pwp"]["coeff"], want_commit[:, r * nl:(r + 1) * nl]["pwp"]["coeff"])
        n_lines += int(got["stats"]["n_lines"].sum())
    assert n_lines > 0
    # the front end against the same gathered blocks equals the single-rank front end
    fe = scene.frontend_cfg(p, beam_width=16)
    starts = np.stack([scene.frontend_starts(sc) for sc in scenes])
    T_ = full.torch
    d_g = T_.zeros(S * N * abi.GUESS_DTYPE.itemsize, dtype=T_.uint8, device=full.device)
    full.frontend(fe, full.to_device(com), full.to_device(starts), d_g)
    T_.cuda.synchronize()
    want_g = d_g.cpu().numpy().view(abi.GUESS_DTYPE).reshape(S, N)
    for r in range(world):
        d_gl = T_.zeros(S * nl * abi.GUESS_DTYPE.itemsize, dtype=T_.uint8, device=full.device)
        ranks[r].frontend_hulls(fe, blocks, ranks[r].to_device(np.ascontiguousarray(starts[:, r * nl:(r + 1) * nl])), d_gl)
        T_.cuda.synchronize()
        got_g = d_gl.cpu().numpy().view(abi.GUESS_DTYPE).reshape(S, nl)
        np.testing.assert_array_equal(got_g["coeff"], want_g[:, r * nl:(r + 1) * nl]["coeff"])
        np.testing.assert_array_equal(got_g["K"], want_g[:, r * nl:(r + 1) * nl]["K"])
    for r in ranks:
        r.close()
    full.close()


@pytest.mark.parametrize("n_agents,n_static,seed,ent,radius", [(64, 20, 31, False, 2.0), (5, 0, 32, False, 2.0), (24, 12, 33, True, 2.0),
                                                               (64, 40, 34, False, 2.0),      # more than 32 static polygons: one segment of them per round of lanes
                                                               (64, 20, 35, False, 500.0),    # nothing is far: every LP of the eight segments is listed — more than the list holds at once
                                                               (40, 36, 36, True, 6.0),
                                                               (200, 20, 37, False, 4.0)])    # 72 (segment, round) items: more than a wave's lanes — the ballots go through LDS, the list is walked serially
def test_packed_separator_gives_the_unpacked_kernels_lines(be, n_agents, n_static, seed, ent, radius):
    """separator_packed_kernel (the spatial presolve's separator: several segments of a slot per wave, LPs solved 64 to a batch
    across the segments) against separator_kernel<0> on the same launch: every line bucket (near lines in call order, parked lines
    from the end), every count and every solution byte for byte — with 1, 3 and 8 segments per wave (the host picks by launch
    size; nep_batch_debug_set_separator_pack forces one), with a candidate list far shorter than a wave (5 agents) and with entangle candidates."""
    import dataclasses, os
    sc = scene.make_scene(n_agents, n_static, seed=seed)
    p = dataclasses.replace(sc["par"], enable_entangle=True) if ent else sc["par"]
    N = p.num_agents
    case = scene.synthetic_entangle(sc, seed=700 + seed, frac=0.3) if ent else None
    bb = be.BatchBackend(p, sc["statics"])
    bb.set_line_cull(radius)
    d_com = bb.to_device(sc["committed"]); d_gue = bb.to_device(sc["guesses"])
    d_ent = bb.torch.from_numpy(np.ascontiguousarray(case).reshape(-1)).to(bb.device) if ent else None

    def run(pack):
        bb.set_separator_pack(pack)
        bb.replan(d_com, d_gue, d_ent=d_ent)
        sol = bb.solutions().copy()
        lines = [bb.debug_lines(a, cap=4096) for a in range(N)]
        return sol, lines, bb.redo_count()
    ref_sol, ref_lines, ref_redo = run(-1)
    assert int(ref_sol["stats"]["n_lines"].sum()) > 0
    for pack in (1, 3, 8):
        sol, lines, redo = run(pack)
        assert sol.tobytes() == ref_sol.tobytes(), pack
        assert redo == ref_redo
        for a in range(N):
            np.testing.assert_array_equal(lines[a][0], ref_lines[a][0], err_msg="pack %s agent %d" % (pack, a))
            np.testing.assert_array_equal(lines[a][1], ref_lines[a][1], err_msg="pack %s agent %d" % (pack, a))
    bb.close()
