"""GPU tests of what decides the LAST DIGITS: the verified line presolve against the every-row solve, the active-set polish of loose and
stalled interior-point exits, the distribution of device-vs-oracle differences on front-end (lattice) guesses, the closed loop's
hard replans against HiGHS's labels, and the reference's own tolerances (nep_batch_set_tolerances)."""
import numpy as np
import pytest

import helpers
from neptune_amd import abi, scene
from gpu_util import _solver, COEF_TOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from neptune_amd import backend
    return backend


@pytest.mark.parametrize("n_agents,n_static,seed,radius", [(8, 10, 3, 3.0), (16, 8, 4, 1.0), (64, 20, 1, 4.0), (64, 20, 2, 0.3)])
def test_line_presolve_leaves_the_optimum_unchanged(be, oracle, n_agents, n_static, seed, radius):
    """nep_batch_set_line_cull: separating lines far from the guess are left out of the QP and verified afterwards
    (re-solve with all lines on a violation) — same statuses and the same trajectories as the full problem, fewer rows."""
    sc = scene.make_scene(n_agents, n_static, seed=seed)
    p = sc["par"]; N = p.num_agents
    bb = be.BatchBackend(p, sc["statics"])
    d_com = bb.to_device(sc["committed"]); d_gue = bb.to_device(sc["guesses"])
    bb.set_line_cull(0.0)              # (every row: the presolve is the handle's default)
    bb.replan(d_com, d_gue)
    full = bb.solutions()
    bb.set_line_cull(radius)
    bb.replan(d_com, d_gue)
    cut = bb.solutions()
    n_redo = bb.redo_count()
    if radius < 1.0:                   # optima farther than this from their guesses: such replans go through the redo pass (every LP, every row)
        assert n_redo > 0
        again = bb.solutions()
        assert (again["stats"]["n_rows"] == full["stats"]["n_rows"]).sum() >= n_redo
    np.testing.assert_array_equal(cut["stats"]["status"], full["stats"]["status"])
    np.testing.assert_array_equal(cut["stats"]["n_lines"], full["stats"]["n_lines"])      # still every line is counted
    np.testing.assert_array_equal(cut["stats"]["n_lp"], full["stats"]["n_lp"])
    assert (cut["stats"]["n_rows"] <= full["stats"]["n_rows"]).all()
    if n_agents >= 16 and radius >= 1.0:      # (a tiny radius sends most replans through the redo pass: all their rows)
        assert cut["stats"]["n_rows"].sum() < 0.6 * full["stats"]["n_rows"].sum()
    ok = full["stats"]["status"] != abi.NEP_FAILED
    assert np.abs(np.array(cut["coeff"])[ok] - np.array(full["coeff"])[ok]).max() <= 1e-7
    assert (full["stats"]["iters"][ok] > 0).all()
    if n_agents == 64 and radius >= 1.0:        # the presolve's other half: replans whose unconstrained minimiser is feasible need no iteration
        assert (cut["stats"]["iters"][ok] == 0).sum() > N // 2
        assert np.abs(cut["stats"]["objective"][ok] - full["stats"]["objective"][ok]).max() <= 1e-7 * (1 + np.abs(full["stats"]["objective"][ok]).max())
    for a in range(0, N, max(1, N // 8)):                                                # and against the oracle
        r = oracle.replan(p, a + 1, sc["committed"], sc["guesses"][a], sc["statics"])
        K = int(cut[a]["K"])
        assert int(cut[a]["stats"]["status"]) == r["status"]
        assert np.abs(np.array(cut[a]["coeff"])[:, :K, :] - r["coeff"]).max() <= COEF_TOL
    bb.close()


@pytest.mark.parametrize("radius", [4.0, 1.0, 0.3])
def test_line_presolve_on_front_end_guesses_with_the_polish_pass(be, radius):
    """Front-end guesses are where interior-point solves end on the loose snapshot, and a culled problem takes another path to another
    loose iterate than the full one: before the polish pass ran under the presolve the two differed by up to 9e-5 in the coefficients
    on these scenes (scripts/presolve_vs_full_fe.py).  Now both end on the certified vertex — the presolve's polish on the near lines,
    accepted only if the point passes the parked lines and the movement bound again: same statuses, coefficients within 2e-6 (what is
    left is two strictly converged interior-point paths), and the polish pass did run under the presolve."""
    from neptune_amd import dist as ndist
    S, N = 8, 64
    scs = [scene.make_scene(N, 20, seed=200 + s) for s in range(S)]
    p = scs[0]["par"]
    com, gue = ndist.stack_scenes(scs)
    bb = be.BatchBackend(p, scs[0]["statics"], n_scenes=S)
    for s in range(1, S):
        bb.set_scene_statics(s, scs[s]["statics"])
    d_com = bb.to_device(com); d_g = bb.to_device(gue)
    bb.frontend(scene.frontend_cfg(p, beam_width=32), d_com, bb.to_device(np.stack([scene.frontend_starts(s) for s in scs])), d_g, None)
    bb.set_line_cull(0.0)
    bb.replan(d_com, d_g); full = bb.solutions().copy()
    listed_full, _ = bb.polish_count()
    assert listed_full >= 1                                          # (loose exits exist on these inputs)
    bb.set_line_cull(radius)                                         # (the polish pass runs under the presolve by default since round 6)
    bb.replan(d_com, d_g); cut = bb.solutions().copy()
    listed, certified = bb.polish_count()
    assert listed >= 1 and certified >= 1
    if radius < 1.0:
        assert bb.redo_count() > S * N // 2                          # (most replans move farther than that: every LP, every row, the plain kernel's hooks)
    np.testing.assert_array_equal(cut["stats"]["status"], full["stats"]["status"])
    ok = full["stats"]["status"] != abi.NEP_FAILED
    d = np.abs(np.array(cut["coeff"]) - np.array(full["coeff"])).reshape(len(full), -1).max(axis=1)[ok]
    assert d.max() <= 2e-6 and (d > 1e-7).sum() <= 0.03 * ok.sum(), (d.max(), int((d > 1e-7).sum()))
    bb.close()


def test_front_end_guesses_converge_without_idling_to_the_iteration_cap(be, oracle):
    """Regression of the solver's stopping rule: with lattice guesses (which end at cruise speed) a few
    replans per thousand used to miss the strict window and idle to the 60-iteration cap, setting the kernel's duration.
    Scene 14 of the bench held such a replan (agent 51).  Every replan of the scene: status and coefficients against
    the oracle, and no iteration count near the cap."""
    sc = scene.make_scene(64, 20, seed=14)
    p = sc["par"]; N = p.num_agents
    statics = scene.make_scene(64, 20, seed=0)["statics"]        # the bench's handle carries seed 0's statics
    bb = be.BatchBackend(p, statics)
    d_com = bb.to_device(sc["committed"])
    d_guess = bb.torch.zeros(N * abi.GUESS_DTYPE.itemsize, dtype=bb.torch.uint8, device=bb.device)
    bb.frontend(scene.frontend_cfg(p, beam_width=32), d_com, bb.to_device(scene.frontend_starts(sc)), d_guess, None)
    bb.replan(None, d_guess)
    sol = bb.solutions()
    g = d_guess.cpu().numpy().view(abi.GUESS_DTYPE)
    iters = sol["stats"]["iters"].astype(int)
    assert iters.max() <= 30, iters.max()
    n = 0
    for a in range(0, N, 3):                                       # every third agent, and the one that used to idle
        for aa in {a, 51}:
            K = int(g[aa]["K"])
            if K == 0:
                continue
            r = oracle.replan(p, aa + 1, sc["committed"], g[aa], statics)
            assert int(sol[aa]["stats"]["status"]) == r["status"], aa
            if r["status"] != 2:
                assert np.abs(np.array(sol[aa]["coeff"])[:, :K, :] - r["coeff"]).max() <= COEF_TOL, aa
                n += 1
    assert n >= 15
    bb.close()


def test_parity_distribution_on_front_end_guesses(be, oracle):
    """Not a sample around the outliers: EVERY replan of four 64-agent scenes on front-end (lattice) guesses — the inputs on
    which the interior point works hardest (8 iterations, relaxed and failed solves) — against the oracle.  Asserted: no
    status mismatch, p99 of the coefficient difference <= 1e-6, maximum <= 1e-4, positions along the trajectories within
    5e-5 m, cost within 1e-8 relative — the bounds of profiles/r03_parity_sweep.txt (scripts/parity_sweep.py: 5 041
    replans of six sizes, no status mismatch; on front-end guesses p99 5.8e-7, max 7.7e-5 on one replan of 1 011 whose two
    interior-point paths took 17 and 18 iterations, positions within 2.5e-5 m, cost within 2.2e-9; on the scenes' own
    guesses everything within 1.5e-8).  Why the maximum is not 1e-6: the tail consists of replans that never pass the strict
    tests and end on the loose-snapshot rule; against the oracle at its limit (profiles/r03_parity_strict.txt) device and
    oracle are each the far one on some of them."""
    from neptune_amd import dist as ndist
    S, N = 4, 64
    scs = [scene.make_scene(N, 20, seed=200 + s) for s in range(S)]
    p = scs[0]["par"]
    com, gue = ndist.stack_scenes(scs)
    bb = be.BatchBackend(p, scs[0]["statics"], n_scenes=S)
    for s in range(1, S):
        bb.set_scene_statics(s, scs[s]["statics"])
    d_com = bb.to_device(com); d_g = bb.to_device(gue)
    bb.frontend(scene.frontend_cfg(p, beam_width=32), d_com, bb.to_device(np.stack([scene.frontend_starts(s) for s in scs])), d_g, None)
    bb.replan(None, d_g)
    sol = bb.solutions().reshape(S, N)
    g = d_g.cpu().numpy().view(abi.GUESS_DTYPE).reshape(S, N)
    dco, dpos, dob, seen = [], [], [], set()
    for s in range(S):
        for a in range(N):
            K = int(g[s, a]["K"])
            if K < 1:
                assert int(sol[s, a]["stats"]["status"]) == 2
                continue
            r = oracle.replan(p, a + 1, scs[s]["committed"], g[s, a], scs[s]["statics"])
            assert int(sol[s, a]["stats"]["status"]) == r["status"], (s, a)
            seen.add(r["status"])
            if r["status"] == 2:
                continue
            dc = np.array(sol[s, a]["coeff"])[:, :K, :] - r["coeff"]
            dco.append(float(np.abs(dc).max()))
            dpos.append(max(float(np.abs(((dc[..., 0] * t + dc[..., 1]) * t + dc[..., 2]) * t + dc[..., 3]).max()) for t in (0.0, 0.125, 0.25, 0.375, 0.5)))
            dob.append(abs(float(sol[s, a]["stats"]["objective"]) - r["objective"]) / (1 + abs(r["objective"])))
    dco = np.array(dco)
    print("front-end guesses, %d replans: coefficients p50 %.2e p99 %.2e max %.2e; positions max %.2e m; cost max %.2e" % (len(dco), np.percentile(dco, 50), np.percentile(dco, 99), dco.max(), max(dpos), max(dob)))
    assert len(dco) >= 230
    assert np.percentile(dco, 99) <= 1e-6 and dco.max() <= 5e-6, (np.percentile(dco, 99), dco.max())
    assert max(dpos) <= 2e-6 and max(dob) <= 1e-8, (max(dpos), max(dob))
    bb.close()


def test_hard_closed_loop_replans_status_against_highs_and_the_oracle(be, oracle):
    """The closed loop's hard replans (tests/golden/moving_hard_cases.npz, see tests/test_oracle_golden.py) through the C ABI with the
    dumped separating lines as input (nep_backend_debug_set_lines): the status PolySolverGurobi::optimize returns is decided by the
    feasibility of the rows (solver_gurobi_poly.cpp:832-861), which HiGHS judges independently of the product and of the oracle.
    Asserted: where the reference's linear rows are decisively infeasible the device fails the same way (first + relaxed problem ->
    FAILED, first only -> RELAXED); where HiGHS finds an interior and the device still gives up (a degenerate optimum: the interior
    point's gap stalls) — counted and bounded; device vs oracle: same status on all but a handful of razor-thin cases (round 4 saw 5
    of 82 disagree, on a script's output; now in the suite), and where both solve, the same optimum."""
    p, cases = helpers.load_moving_hard_cases()
    s = _solver(be, p, 1)
    s.setStaticObstVert([])
    n_lab = {0: 0, 1: 0, 2: 0}; missed, differ, dcost = [], [], []
    for k, c in enumerate(cases):
        K = c["K"]
        s.setInitTrajectory(np.arange(K + 1) * p.T_span, c["coeff"]); s.setHulls([]); s.debugSetLines(c["seg"], c["nd"])
        ok, obj = s.optimize()
        r = oracle.optimize(p, 1, c["coeff"], [], [], lines=(c["seg"], c["nd"]))
        if s.status != r["status"]:
            differ.append((k, s.status, r["status"], c["expected"]))
        elif s.status != abi.NEP_FAILED:
            dcost.append(abs(obj - r["objective"]) / (1 + abs(r["objective"])))
        e = c["expected"]
        if e < 0:
            continue
        n_lab[e] += 1
        if e == 2:
            assert s.status == abi.NEP_FAILED and not ok, (k, s.status)
            _, coeff, _ = s.generatePwpOut(0.0, p.dc)
            np.testing.assert_array_equal(coeff, c["coeff"])                     # output == the initial guess (:856-859)
        elif e == 1:
            assert s.status == abi.NEP_RELAXED, (k, s.status)
        elif s.status != abi.NEP_OK:
            missed.append((k, s.status, c["t_first"]))
    s.close()
    assert n_lab[2] >= 30 and n_lab[1] >= 5 and n_lab[0] >= 20
    print("device gave up on %d of %d replans HiGHS finds strictly feasible: %r; device != oracle on %d of %d: %r; cost where both solve: max rel %.2e"
          % (len(missed), n_lab[0], missed, len(differ), len(cases), differ, max(dcost)))
    assert len(missed) <= 2 and len(differ) <= 5                                 # (measured: 1 and 4, rounds 5 and 6; one of slack each)
    assert not [d for d in differ if d[3] > 0]                                   # never on a decisively infeasible case
    assert max(dcost) <= 1e-6


def test_polish_finishes_loose_and_stalled_solves_exactly(be, oracle):
    """The active-set polish (nep_batch_set_polish, on by default; oracle: orc_set_polish): an interior-point solve that ends on its
    loose snapshot or gives up is finished by an exact active-set solve when a KKT certificate exists.  On four 64-agent scenes'
    front-end guesses (the inputs whose tail was 8e-5 in round 4): with the polish OFF on both sides device and oracle reproduce the
    round-4 behaviour (same statuses); with it ON every replan the device listed and certified agrees with the oracle's optimum to
    1e-8 in the coefficients — two roundings of a 24 x 24 solve, not two interior-point paths — and no status gets worse."""
    from neptune_amd import dist as ndist
    S, N = 4, 64
    scs = [scene.make_scene(N, 20, seed=200 + s) for s in range(S)]
    p = scs[0]["par"]
    com, gue = ndist.stack_scenes(scs)
    bb = be.BatchBackend(p, scs[0]["statics"], n_scenes=S)
    for s in range(1, S):
        bb.set_scene_statics(s, scs[s]["statics"])
    d_com = bb.to_device(com); d_g = bb.to_device(gue)
    bb.frontend(scene.frontend_cfg(p, beam_width=32), d_com, bb.to_device(np.stack([scene.frontend_starts(s) for s in scs])), d_g, None)
    g = d_g.cpu().numpy().view(abi.GUESS_DTYPE).reshape(S, N)
    bb.set_line_cull(0.0)              # (every row: the interior point's loose exits as they are; the presolved path is the next tests')
    bb.set_polish(False); bb.replan(None, d_g); off = bb.solutions().reshape(S, N).copy()
    assert bb.polish_count() == (0, 0)
    bb.set_polish(True); bb.replan(None, d_g); on = bb.solutions().reshape(S, N).copy()
    listed, certified = bb.polish_count()
    assert listed >= 1 and 1 <= certified <= listed
    st_off = off["stats"]["status"].astype(int); st_on = on["stats"]["status"].astype(int)
    assert (st_on <= st_off).all()                                   # a certificate only ever turns a failure into a success
    changed = np.argwhere((np.abs(on["coeff"] - off["coeff"]).reshape(S, N, -1).max(axis=2) > 0) | (st_on != st_off))
    assert 1 <= len(changed) <= listed
    worst_both, worst_one, n_same, n_both = 0.0, 0.0, 0, 0
    oracle.set_polish(True)
    for s, a in changed:
        K = int(g[s, a]["K"])
        oracle.last_polished()
        r = oracle.replan(p, a + 1, scs[s]["committed"], g[s, a], scs[s]["statics"])
        both = oracle.last_polished()                            # the oracle's solve of this replan ended on the polish too
        if r["status"] != int(st_on[s, a]):
            continue                                             # (a razor-thin certificate one side found and the other did not: counted below)
        n_same += 1
        if r["status"] != 2:
            d = float(np.abs(np.array(on[s, a]["coeff"])[:, :K, :] - r["coeff"]).max())
            if both:
                n_both += 1; worst_both = max(worst_both, d)
            else:
                worst_one = max(worst_one, d)
    # both polished: two roundings of one small linear solve; only the device did (the oracle's interior point passed its strict tests):
    # the device's exact optimum against an iterate that is converged to 1e-9 in the residuals
    assert n_same >= len(changed) - 2 and worst_both <= 1e-8 and worst_one <= COEF_TOL, (n_same, len(changed), n_both, worst_both, worst_one)
    bb.close()


def test_reference_tolerances(be, oracle):
    """nep_batch_set_tolerances(1e-6, 1e-8) — Gurobi's default barrier tolerances, where the reference's solver stops
    (solver_gurobi_poly.cpp:811-812 sets OutputFlag and TimeLimit only) — on two scenes' own guesses and two scenes' front-end
    guesses: same statuses as the oracle given the same tolerances (orc_set_qp_tolerances), costs within 1e-6 relative of it
    and of the strictly converged optimum (north star: 1e-4), positions within a millimetre of the strict optimum, fewer
    iterations; and the handle goes back to the strict tests bit for bit."""
    from neptune_amd import dist as ndist
    S, N = 4, 64
    scs = [scene.make_scene(N, 20, seed=210 + s) for s in range(S)]
    p = scs[0]["par"]
    com, gue = ndist.stack_scenes(scs)
    bb = be.BatchBackend(p, scs[0]["statics"], n_scenes=S)
    for s in range(1, S):
        bb.set_scene_statics(s, scs[s]["statics"])
    d_com = bb.to_device(com); d_g = bb.to_device(gue)
    d_fe = bb.to_device(gue)
    bb.frontend(scene.frontend_cfg(p, beam_width=32), d_com, bb.to_device(np.stack([scene.frontend_starts(s) for s in scs])), d_fe, None)
    g_own = gue.reshape(S, N); g_fe = d_fe.cpu().numpy().view(abi.GUESS_DTYPE).reshape(S, N)
    g = np.concatenate([g_own[:2], g_fe[2:]])                 # scenes 0, 1: their own guesses; scenes 2, 3: lattice guesses
    d_mix = bb.to_device(g)
    bb.replan(d_com, d_mix); strict = bb.solutions().reshape(S, N).copy()
    bb.set_tolerances(1e-6, 1e-8)
    bb.replan(d_com, d_mix); loose = bb.solutions().reshape(S, N).copy()
    bb.set_tolerances()                                          # back to the defaults (1e-10 / 1e-11)
    bb.replan(d_com, d_mix); again = bb.solutions().reshape(S, N)
    assert again["coeff"].tobytes() == strict["coeff"].tobytes() and (again["stats"]["status"] == strict["stats"]["status"]).all()
    it_on = strict["stats"]["iters"] > 0                         # (the presolve — the default — ends most own-guess replans without an iteration)
    it_s, it_l = strict["stats"]["iters"][it_on].mean(), loose["stats"]["iters"][it_on].mean()
    assert it_l <= it_s - 0.3, (it_s, it_l)
    oracle.set_qp_tolerances(1e-6, 1e-8)
    try:
        dcost_o, dcost_s, dpos_s, n = [], [], [], 0
        for s in range(S):
            for a in range(0, N, 2):
                K = int(g[s, a]["K"])
                if K < 1:
                    continue
                r = oracle.replan(p, a + 1, scs[s]["committed"], g[s, a], scs[s]["statics"])
                assert int(loose[s, a]["stats"]["status"]) == r["status"], (s, a)
                if r["status"] == 2:
                    continue
                n += 1
                dcost_o.append(abs(float(loose[s, a]["stats"]["objective"]) - r["objective"]) / (1 + abs(r["objective"])))
                if int(strict[s, a]["stats"]["status"]) == r["status"]:
                    so = float(strict[s, a]["stats"]["objective"])
                    dcost_s.append(abs(float(loose[s, a]["stats"]["objective"]) - so) / (1 + abs(so)))
                    dc = np.array(loose[s, a]["coeff"])[:, :K, :] - np.array(strict[s, a]["coeff"])[:, :K, :]
                    dpos_s.append(max(float(np.abs(((dc[..., 0] * t + dc[..., 1]) * t + dc[..., 2]) * t + dc[..., 3]).max()) for t in (0.0, 0.125, 0.25, 0.375, 0.5)))
    finally:
        oracle.set_qp_tolerances()
    print("reference tolerances: %d replans, iterations %.2f -> %.2f; cost vs oracle (same tolerances) max %.2e; vs the strict optimum: cost max %.2e, position p99 %.2e max %.2e m"
          % (n, it_s, it_l, max(dcost_o), max(dcost_s), np.percentile(dpos_s, 99), max(dpos_s)))
    assert n >= 100
    assert max(dcost_o) <= 1e-6 and max(dcost_s) <= 1e-6, (max(dcost_o), max(dcost_s))
    assert max(dpos_s) <= 1e-3, max(dpos_s)
    bb.close()


def test_one_slot_launch_lists_itself_for_the_polish_pass(be):
    """A launch of ONE replan (the per-agent handle's shape: one workgroup) keeps no counter to zero beforehand — qp_reg_kernel's last
    lines set the polish pass's list themselves.  One-agent shards of a 64-agent scene, on front-end guesses whose solves end loose,
    give the full batch's trajectories bit for bit: plain (every row) and under the presolve (the default; the pass runs there too, the
    third instantiation); the counters say 1 listed where the batch listed that agent."""
    sc = scene.make_scene(64, 20, seed=203)
    p = sc["par"]; N = p.num_agents
    full = be.BatchBackend(p, sc["statics"])
    d_com = full.to_device(sc["committed"][None]); d_g = full.to_device(sc["guesses"][None])
    full.frontend(scene.frontend_cfg(p, beam_width=32), d_com, full.to_device(scene.frontend_starts(sc)[None]), d_g, None)
    g = d_g.cpu().numpy().view(abi.GUESS_DTYPE).reshape(N)
    full.set_line_cull(0.0); full.set_polish(False); full.replan(d_com, d_g); off = full.solutions().copy()
    for mode, cull in ((1, 0.0), (1, 4.0)):
        full.set_line_cull(cull); full.set_polish(mode); full.replan(d_com, d_g); want = full.solutions().copy()
        listed_all, _ = full.polish_count()
        assert listed_all >= 1 or cull > 0.0            # (under the presolve fewer solves end loose: most replans meet no active row at all)
        changed = [a for a in range(N) if np.abs(np.array(want[a]["coeff"]) - np.array(off[a]["coeff"])).max() > 0 or want[a]["stats"]["status"] != off[a]["stats"]["status"]]
        picks = (changed[:3] + [a for a in range(N) if a not in changed][:2]) if cull == 0.0 else list(range(0, N, 9))
        n_listed = 0
        for a in picks:
            one = be.BatchBackend(p, sc["statics"], first_local=a, n_local=1)
            one.set_line_cull(cull); one.set_polish(mode)
            one.replan(one.to_device(sc["committed"][None]), one.to_device(g[a:a + 1]))
            got = one.solutions()[0]
            n_listed += one.polish_count()[0]
            assert one.polish_count()[0] in (0, 1)
            assert int(got["stats"]["status"]) == int(want[a]["stats"]["status"]), (mode, a)
            np.testing.assert_array_equal(np.array(got["coeff"]), np.array(want[a]["coeff"]))
            one.close()
        if cull == 0.0:
            assert n_listed >= min(len(changed), 3) and len(changed) >= 1
    full.close()


def test_replan_as_two_enqueues_is_the_same_replan(be):
    """nep_batch_replan_lines (hulls + separating lines) then nep_batch_replan_solve (the QPs on them), on two streams ordered by an event,
    against nep_batch_replan: every byte of the solutions, sampled states and commit records, and the same lines."""
    import torch
    sc = scene.make_scene(64, 20, seed=5)
    p = sc["par"]
    bb = be.BatchBackend(p, sc["statics"])
    d_com = bb.to_device(sc["committed"]); d_gue = bb.to_device(sc["guesses"])
    bb.replan(d_com, d_gue)
    torch.cuda.synchronize()
    sol1 = bb.solutions().copy()
    one = (bb.d_states.clone(), bb.d_commit.clone())
    lines_one = [bb.debug_lines(a) for a in range(0, p.num_agents, 7)]
    for t in (bb.d_solution, bb.d_states, bb.d_commit):
        t.zero_()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ev = torch.cuda.Event()
    s1.wait_stream(torch.cuda.current_stream())
    bb.replan_lines(d_com, d_gue, stream=s1)
    ev.record(s1)
    s2.wait_event(ev)
    bb.replan_solve(d_com, d_gue, stream=s2)
    torch.cuda.synchronize()
    for a, b in zip(one, (bb.d_states, bb.d_commit)):
        assert torch.equal(a, b)
    sol2 = bb.solutions()
    np.testing.assert_array_equal(np.array(sol1["coeff"]), np.array(sol2["coeff"]))
    for f in ("status", "iters", "objective", "n_lines", "n_rows", "n_lp"):      # (everything but the measured times)
        np.testing.assert_array_equal(sol1["stats"][f], sol2["stats"][f])
    for a, (seg, nd) in zip(range(0, p.num_agents, 7), lines_one):
        seg2, nd2 = bb.debug_lines(a)
        np.testing.assert_array_equal(seg, seg2); np.testing.assert_array_equal(nd, nd2)
    assert (sol2["stats"]["status"] != abi.NEP_FAILED).sum() > 32
    bb.close()
