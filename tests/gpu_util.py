"""Shared by the GPU parity test files: tolerances, the per-agent solver set up the way the reference sets it up, and the scene
checker (every replan of a scene against the oracle)."""
import numpy as np

from neptune_amd import abi, scene

COEF_TOL = 1e-6
COST_RTOL = 1e-6


def _bounds(p):
    return (p.x_min, p.x_max, p.y_min, p.y_max, p.z_min, p.z_max, p.v_max, p.a_max, p.j_max)


def _solver(be, p, agent_id=1):
    s = be.PolySolver(p.num_pol, 3, agent_id, p.T_span, p.pb, p.weight, 0.5, True)
    s.setMaxValues(*_bounds(p)); s.setMaxRuntime(0.05); s.setTetherLength(p.tether_length)
    return s


def lines_match(bb, seg, nd, r):
    """the lines a handle holds for one replan (debug_lines) against the oracle's (r = oracle.replan(...)): with every row through the
    interior point (line presolve off) the same lines bit for bit in the reference's call order; under the presolve (the default) the
    buckets hold the near lines first, then the parked ones, and LPs skipped by the box test made no line at all — every line present
    must be one of the oracle's, bit for bit, in the same segment"""
    if bb.line_cull() == 0.0:
        np.testing.assert_array_equal(seg, r["line_seg"])
        np.testing.assert_array_equal(nd, r["line_nd"])
        return
    want = {(int(s_), l.tobytes()) for s_, l in zip(r["line_seg"], r["line_nd"])}
    assert all((int(s_), np.ascontiguousarray(l).tobytes()) in want for s_, l in zip(seg, nd))


def solver_lines_match(s, r, ordered=False):
    """the per-agent handle's lines of the last optimize() (debugGetLines) against the oracle's: it never skips an LP (its hull lists
    are the caller's), so the same lines bit for bit — in the reference's call order with the presolve off (ordered=True after
    setLineCull(0)), as a multiset under the default presolve (near lines first, parked ones after)"""
    seg, nd = s.debugGetLines()
    if ordered:
        np.testing.assert_array_equal(seg, r["line_seg"]); np.testing.assert_array_equal(nd, r["line_nd"])
        return
    got = sorted((int(a), np.ascontiguousarray(l).tobytes()) for a, l in zip(seg, nd))
    want = sorted((int(a), np.ascontiguousarray(l).tobytes()) for a, l in zip(r["line_seg"], r["line_nd"]))
    assert got == want


def _check_scene(be, oracle, sc, n_scenes=1, first_local=0, n_local=None):
    """Every replan of a scene against the oracle, on BOTH solve paths of the handle: `full` = every separating-line row through the
    interior point (nep_batch_set_line_cull(0): lines bit-exact in the reference's call order, LP / row counts), and the handle's
    default = the verified line presolve with the polish pass under it (statuses, coefficients, cost, samples, commit records to the
    same tolerances; its line buckets hold the near lines first and never-made lines are absent, so lines are checked as a subset)."""
    p = sc["par"]
    worst = 0.0
    refs = {}
    for mode in ("full", "default"):
        bb = be.BatchBackend(p, sc["statics"], first_local=first_local, n_local=n_local)
        if mode == "full":
            bb.set_line_cull(0.0)
        else:
            assert bb.line_cull() == 4.0          # the default at every size (round 6)
        nl = bb.n_local
        d_comm = bb.to_device(sc["committed"]); d_guess = bb.to_device(sc["guesses"][first_local:first_local + nl])
        bb.replan(d_comm, d_guess)
        sol = bb.solutions(); states = bb.states(); com = bb.commits()
        hx, hn = bb.debug_hulls(0)
        for a in range(nl):
            aid = first_local + a + 1
            if aid not in refs:
                refs[aid] = oracle.replan(p, aid, sc["committed"], sc["guesses"][aid - 1], sc["statics"], want_hulls=True)
            r = refs[aid]
            K = int(sol[a]["K"])
            st = sol[a]["stats"]
            seg, nd = bb.debug_lines(a)
            if mode == "full":
                # hulls: oracle lists the present agents in id order (own skipped)
                others = [j for j in range(p.num_agents) if j != aid - 1]
                for oj, j in enumerate(others):
                    for i in range(p.num_pol):
                        nv = r["hull_nv"][oj * p.num_pol + i]
                        assert hn[j, i] == nv
                        np.testing.assert_array_equal(hx[j, i, :nv], r["hull_xy"][oj * p.num_pol + i, :nv])
                np.testing.assert_array_equal(seg, r["line_seg"])
                np.testing.assert_array_equal(nd, r["line_nd"])                 # bit-exact lines, reference loop order
                assert int(st["n_rows"]) == r["n_rows"]
            else:
                # every line the presolved handle holds is one of the oracle's, bit for bit, in the same segment
                want = {(int(s_), l.tobytes()) for s_, l in zip(r["line_seg"], r["line_nd"])}
                assert all((int(s_), l.tobytes()) in want for s_, l in zip(seg, nd)), aid
                assert int(st["n_rows"]) <= r["n_rows"]
            assert int(st["status"]) == r["status"] and int(st["n_lines"]) == r["n_lines"], (mode, aid)
            assert int(st["n_lp"]) == r["n_lp"] and int(st["n_lp_failed"]) == r["n_lp_failed"], (mode, aid)
            co = np.array(sol[a]["coeff"])[:, :K, :]
            err = np.abs(co - r["coeff"]).max(); worst = max(worst, err)
            assert err <= COEF_TOL, (mode, aid, err)
            if r["status"] != 2:
                assert abs(float(st["objective"]) - r["objective"]) <= COST_RTOL * (1 + abs(r["objective"])), (mode, aid)
            ref = oracle.sample(co, p.T_span, p.dc, cap=p.max_states)
            assert int(sol[a]["n_states"]) == len(ref)
            np.testing.assert_allclose(states[a, :len(ref)], ref, rtol=0, atol=1e-12)
            t0 = float(sc["guesses"][aid - 1]["t_start"])
            np.testing.assert_allclose(np.array(sol[a]["times"])[:K + 1], t0 + np.arange(K + 1) * p.T_span, atol=1e-12)
            assert int(com[a]["id"]) == aid and int(com[a]["pwp"]["n_seg"]) == K
            np.testing.assert_array_equal(np.array(com[a]["pwp"]["coeff"])[:, :K, :], co)
        bb.close()
    return worst
