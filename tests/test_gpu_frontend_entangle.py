"""GPU: the front end with the entangle check on (nep_batch_frontend_ent) against the oracle bit for bit — guesses,
per-search counters and the case block — then the back end on device-made guesses AND device-made entangle cases, and
the safety pass's entangle re-check."""
import dataclasses

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


def _run_frontend(be, sc, W, inits=None, fast_caps=None, big_records=None):
    p = sc["par"]; N = p.num_agents
    fe = scene.frontend_cfg(p, beam_width=W, entangle=True)
    bb = be.BatchBackend(p, sc["statics"])
    if fast_caps is not None:
        bb.set_fe_ent_fast_caps(*fast_caps)
    if big_records is not None:
        bb.set_fe_ent_big_records(big_records)
    reps, longest = scene.static_reps(sc["statics"]) if len(sc["statics"]) else (np.zeros((0, 2, 2)), np.zeros((0, 2)))
    bb.set_static_reps(reps, longest)
    T = bb.torch
    starts = scene.frontend_starts(sc)
    d_g = T.zeros(N * abi.GUESS_DTYPE.itemsize, dtype=T.uint8, device=bb.device)
    d_r = T.zeros(N * abi.FE_RESULT_DTYPE.itemsize, dtype=T.uint8, device=bb.device)
    d_case = T.zeros(N * abi.NEP_MAX_POL * N, dtype=T.int32, device=bb.device)
    d_init = bb.to_device(inits) if inits is not None else None
    d_com = bb.to_device(sc["committed"])
    bb.frontend_ent(fe, d_com, bb.to_device(starts), d_g, d_r, d_case, d_ent_init=d_init)
    T.cuda.synchronize()
    return bb, fe, starts, d_com, d_g, d_g.cpu().numpy().view(abi.GUESS_DTYPE), d_r.cpu().numpy().view(abi.FE_RESULT_DTYPE), d_case, d_case.cpu().numpy().reshape(N, abi.NEP_MAX_POL, N)


@pytest.mark.parametrize("n_agents,n_static,seed,W,stride,bends", [(8, 6, 60, 16, 1, 0), (8, 6, 56, 32, 1, 0), (16, 8, 61, 16, 1, 0),
                                                                   (72, 40, 63, 16, 6, 0),       # (more than one 32-bit word of agents and of statics in the kernel's per-parent masks; every sixth agent against the oracle)
                                                                   (5, 0, 64, 64, 1, 0),         # (round 4: the widest beam — per-rank arrays at 64 — in a scene too small for the LDS aliasing of the winners' f values)
                                                                   (16, 8, 65, 48, 2, 0),
                                                                   (16, 8, 66, 16, 1, 1), (24, 10, 67, 32, 2, 1),      # (tethers with 2-4 bend points, as the bench's config-5 inputs have them: the multi-segment paths of the masks and of the crossing test)
                                                                   (12, 4, 68, 16, 1, 2)])                             # (5-8 bend points: beyond what those paths keep in registers)
def test_entangle_front_end_matches_the_oracle_bit_for_bit(be, oracle, n_agents, n_static, seed, W, stride, bends):
    sc = scene.tether_crossing_scene(n_agents, n_static, seed)
    if bends:
        scene.synthetic_entangle(sc, seed=900 + seed, frac=0.1, nb_range=(2, 5) if bends == 1 else (5, 9))
    p = sc["par"]; N = n_agents
    rng = np.random.default_rng(seed)
    inits = np.zeros(N, dtype=abi.FE_ENT_STATE_DTYPE)           # some searches start with a crossing already on the list
    for a in range(0, N, 3):
        j = int((a + 1 + rng.integers(0, N - 1)) % N)
        if j != a:
            inits[a]["n_alpha"] = 1; inits[a]["id"][0] = j + 1; inits[a]["cs"][0] = int(rng.integers(0, 3))
    bb, fe, starts, d_com, d_g, got_g, got_r, d_case, got_case = _run_frontend(be, sc, W, inits)
    n_cases = n_pruned = 0
    for a in range(0, N, stride):
        hx, hn = oracle.hulls_of_scene(p, a + 1, sc["committed"], float(starts[a]["t_start"]), sc["statics"])
        ent = helpers.ent_inputs(sc, a, t0=float(starts[a]["t_start"]), init=inits[a])
        g, res, case = oracle.frontend_beam_ent(p, fe, a + 1, starts[a], hx, hn, sc["statics"], ent)
        assert int(got_g[a]["K"]) == int(g["K"]), a
        np.testing.assert_array_equal(np.array(got_g[a]["coeff"]), np.array(g["coeff"]), err_msg="agent %d" % a)
        for f in ("status", "K", "n_children", "n_feasible", "n_collision_free", "n_entangled", "ent_overflow"):
            assert int(got_r[a][f]) == res[f], (a, f, int(got_r[a][f]), res[f])
        assert float(got_r[a]["cost"]) == res["cost"]
        np.testing.assert_array_equal(got_case[a], case, err_msg="case block of agent %d" % a)
        n_cases += int((case != 0).sum()); n_pruned += res["n_entangled"]
    assert n_cases > 0
    # the back end on device-made guesses and device-made cases: lines and optimum equal the oracle fed the same
    bb.set_line_cull(0.0)        # (the larger case would get the presolve by default: the lines are compared in the oracle's order here)
    bb.replan(None, d_g, d_ent=d_case)
    sol = bb.solutions()
    extra = 0
    for a in range(0, N, stride):
        K = int(got_g[a]["K"])
        if K < 1:
            assert int(sol[a]["stats"]["status"]) == 2
            continue
        r = oracle.replan(p, a + 1, sc["committed"], got_g[a], sc["statics"], case_id=got_case[a])
        r0 = oracle.replan(p, a + 1, sc["committed"], got_g[a], sc["statics"])
        extra += r["n_lp"] - r0["n_lp"]
        seg, nd = bb.debug_lines(a)
        np.testing.assert_array_equal(nd, r["line_nd"])
        assert int(sol[a]["stats"]["status"]) == r["status"]
        assert np.abs(np.array(sol[a]["coeff"])[:, :K, :] - r["coeff"]).max() <= 1e-6
    assert extra >= 0     # (whether a case yields an LP depends on the bend points' distance cull, solver_gurobi_poly.cpp:738-745:
    bb.close()            #  tests/test_gpu_entangle_config5.py::test_real_entangle_states_drive_the_entangle_rows covers scenes where they do)


@pytest.mark.parametrize("fast_caps", [(0, 32, 8), (2, 32, 8), (40, 1, 8), (40, 32, 0), (1, 0, 0)])
def test_big_records_carry_what_the_fixed_record_cannot(be, oracle, fast_caps):
    """A search node whose crossing list, new crossings of one sampled step or bend points outgrow the fixed record
    (NEP_FE_ENT_CAP / 32 / NEP_MAX_BEND) is carried in a big record of the handle's pool, bounded by the reference's own
    rule only (kinodynamic_search.cpp:850-854).  With the fixed record's limits shrunk (list entries, new crossings per
    step, bend points) an ordinary scene runs through the big records: every output is the default run's, bit for bit —
    and the oracle's, which has no capacity at all."""
    sc = scene.tether_crossing_scene(16, 8, 61)
    p = sc["par"]; N = 16
    rng = np.random.default_rng(5)
    inits = np.zeros(N, dtype=abi.FE_ENT_STATE_DTYPE)
    for a in range(0, N, 2):
        j = int((a + 1 + rng.integers(0, N - 1)) % N)
        if j != a:
            inits[a]["n_alpha"] = 1; inits[a]["id"][0] = j + 1; inits[a]["cs"][0] = int(rng.integers(0, 3))
    ref = _run_frontend(be, sc, 16, inits)
    got = _run_frontend(be, sc, 16, inits, fast_caps=fast_caps, big_records=1 << 16)
    ref[0].close(); got[0].close()
    assert (got[6]["ent_overflow"] == 0).all() and (ref[6]["ent_overflow"] == 0).all()
    big = got[6]["_pad"].astype(np.int64) >> 8
    assert (ref[6]["_pad"] == 0).all()                       # (this scene needs no big record at the default limits)
    assert big.sum() > 50, big                                # (and many with the shrunk ones)
    assert got[5].tobytes() == ref[5].tobytes()               # guesses
    np.testing.assert_array_equal(got[8], ref[8])             # case blocks
    for f in ("status", "K", "depth", "n_children", "n_feasible", "n_collision_free", "n_entangled", "cost", "dist_to_goal"):
        np.testing.assert_array_equal(got[6][f], ref[6][f], err_msg=f)
    starts = got[2]
    for a in (0, 5, 10):
        hx, hn = oracle.hulls_of_scene(p, a + 1, sc["committed"], float(starts[a]["t_start"]), sc["statics"])
        ent = helpers.ent_inputs(sc, a, t0=float(starts[a]["t_start"]), init=inits[a])
        g, res, case = oracle.frontend_beam_ent(p, got[1], a + 1, starts[a], hx, hn, sc["statics"], ent)
        np.testing.assert_array_equal(np.array(got[5][a]["coeff"]), np.array(g["coeff"]))
        np.testing.assert_array_equal(got[8][a], case)
        assert int(got[6][a]["n_entangled"]) == res["n_entangled"] and res["ent_overflow"] == 0


def test_a_bundle_of_tethers_needs_big_records_at_the_default_limits(be, oracle):
    """No test knob: agent 7 starts in front of a bundle of 59 tethers that cross its way within a third of a metre, so its first
    sampled steps add more than 32 crossings at once and its nodes carry more than 40 — the fixed record's limits.  The search is
    listed, run again on big records, and equals the oracle (which has no capacity) bit for bit; nothing is flagged."""
    a = 7
    sc = helpers.bundle_scene(60, a, (0.004, 0.03), P=(-6.0, -0.5), Q=(6.0, 0.5))
    p = sc["par"]; N = 60
    P, Q = sc["bundle"]
    fe = scene.frontend_cfg(p, beam_width=16, entangle=True)
    bb = be.BatchBackend(p, sc["statics"])
    bb.set_static_reps(np.zeros((0, 2, 2)), np.zeros((0, 2)))
    T = bb.torch
    starts = scene.frontend_starts(sc)
    d = (Q - P) / np.linalg.norm(Q - P)
    starts[a]["pos"][:2] = P; starts[a]["vel"][:2] = 1.5 * d; starts[a]["accel"][:2] = 0.0; starts[a]["goal"][:2] = Q
    d_g = T.zeros(N * abi.GUESS_DTYPE.itemsize, dtype=T.uint8, device=bb.device)
    d_r = T.zeros(N * abi.FE_RESULT_DTYPE.itemsize, dtype=T.uint8, device=bb.device)
    d_case = T.zeros(N * abi.NEP_MAX_POL * N, dtype=T.int32, device=bb.device)
    bb.frontend_ent(fe, bb.to_device(sc["committed"]), bb.to_device(starts), d_g, d_r, d_case)
    T.cuda.synchronize()
    got_g = d_g.cpu().numpy().view(abi.GUESS_DTYPE); got_r = d_r.cpu().numpy().view(abi.FE_RESULT_DTYPE); got_case = d_case.cpu().numpy().reshape(N, abi.NEP_MAX_POL, N)
    bb.close()
    assert (got_r["ent_overflow"] == 0).all()
    assert int(got_r[a]["_pad"]) >> 8 > 0, "agent 7's search was expected to need big records"
    for b in (a, 0, 30):
        hx, hn = oracle.hulls_of_scene(p, b + 1, sc["committed"], float(starts[b]["t_start"]), sc["statics"])
        ent = helpers.ent_inputs(sc, b, t0=float(starts[b]["t_start"]))
        g, res, case = oracle.frontend_beam_ent(p, fe, b + 1, starts[b], hx, hn, sc["statics"], ent)
        np.testing.assert_array_equal(np.array(got_g[b]["coeff"]), np.array(g["coeff"]), err_msg="agent %d" % b)
        for f in ("status", "K", "n_children", "n_feasible", "n_collision_free", "n_entangled", "ent_overflow"):
            assert int(got_r[b][f]) == res[f], (b, f, int(got_r[b][f]), res[f])
        np.testing.assert_array_equal(got_case[b], case)
    assert int((got_case[a] != 0).sum()) > 40 * 2       # (more than 40 active cases along agent 7's plan)


def test_an_exhausted_pool_of_big_records_is_flagged(be):
    """nep_fe_result.ent_overflow: the only capacity left is the pool itself — a child that finds it empty is pruned and
    the search says so (bit 3 of _pad)."""
    sc = scene.tether_crossing_scene(16, 8, 61)
    got = _run_frontend(be, sc, 16, fast_caps=(0, 32, 8), big_records=3)
    got[0].close()
    r = got[6]
    assert r["ent_overflow"].sum() > 0
    assert ((r["_pad"][r["ent_overflow"] != 0] & 8) != 0).all()


def test_config5_style_device_made_guesses_and_cases(be):
    """BASELINE configs[4] ingredients at a size one test can afford (64 agents + 20 obstacles, entangle check on): guesses
    and entangle cases both made on the device, then separator + QP on them; every returned plan is entangle-free by the
    host library's own propagation."""
    sc = scene.make_scene(64, 20, seed=7)
    sc["par"] = dataclasses.replace(sc["par"], enable_entangle=True)
    p = sc["par"]
    bb, fe, starts, d_com, d_g, got_g, got_r, d_case, got_case = _run_frontend(be, sc, 16)
    assert (got_r["ent_overflow"] == 0).all()
    K = got_g["K"].astype(int)
    assert (K >= 1).sum() >= 56
    bb.replan(None, d_g, d_ent=d_case)
    sol = bb.solutions()
    st = sol["stats"]["status"].astype(int)
    assert ((st != 2) | (K < 1)).mean() > 0.9
    case_h, hit_h, _ = scene.real_entangle(dict(sc, guesses=got_g))       # host library (neptune_amd/entangle.py) on the device's guesses
    for a in range(64):
        if K[a] < 1:
            continue
        assert int(hit_h[a]) == 0, a
        np.testing.assert_array_equal(got_case[a][: K[a]], case_h[a][: K[a]], err_msg="agent %d" % a)
    bb.close()


@pytest.mark.parametrize("fast_add", [32, 0])
def test_safety_pass_entangle_recheck(be, oracle, fast_add):
    """nep_batch_safety_commit_ent: the flags equal the oracle's entangleCheckGivenPwp on every new trajectory (against
    everybody's NEW trajectories), and a flagged agent keeps its previous record.  fast_add = 0: every interval with a
    crossing goes through the re-check's big records (what a step of more than 32 new crossings does)."""
    sc = scene.tether_crossing_scene(8, 6, 60)
    p = sc["par"]; N = 8
    bb = be.BatchBackend(p, sc["statics"])
    bb.set_fe_ent_fast_caps(40, fast_add, 8)
    reps, longest = scene.static_reps(sc["statics"])
    bb.set_static_reps(reps, longest)
    T = bb.torch
    d_prev = bb.to_device(sc["committed"]); d_guess = bb.to_device(sc["guesses"])
    bb.replan(d_prev, d_guess)
    fresh = bb.commits()
    # entangle states at the start that make some re-checks fire: a crossing with a neighbour already on the list
    inits = np.zeros(N, dtype=abi.FE_ENT_STATE_DTYPE)
    for a in range(N):
        j = (a + 3) % N
        inits[a]["n_alpha"] = 1; inits[a]["id"][0] = j + 1; inits[a]["cs"][0] = 1
    d_final = T.zeros_like(bb.d_commit); d_acc = T.zeros(N, dtype=T.int32, device=bb.device)
    bb.safety_commit_ent(d_prev, bb.d_commit, d_guess, d_final, d_acc, d_ent_init=bb.to_device(inits))
    T.cuda.synchronize()
    acc = d_acc.cpu().numpy(); fin = d_final.cpu().numpy().view(abi.TRAJ_REC_DTYPE)
    conflict = bb.debug_conflicts(0)
    want = []
    for a in range(N):
        ent = helpers.ent_inputs(sc, a, recs=fresh, t0=float(sc["guesses"][a]["t_start"]), init=inits[a])
        want.append(oracle.entangle_check_pwp(p, a + 1, p.tether_length, np.array(fresh[a]["pwp"]["coeff"])[0, 0], np.array(fresh[a]["pwp"]["coeff"])[1, 0], ent))
    accept = []
    for a in range(N):
        bad = want[a] or any(accept[j] and (conflict[a, j] or conflict[j, a]) for j in range(a))
        accept.append(0 if bad else 1)
    np.testing.assert_array_equal(acc, accept)
    for a in range(N):
        assert fin[a].tobytes() == (fresh[a] if accept[a] else sc["committed"][a]).tobytes()
    bb.close()
