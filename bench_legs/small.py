"""single_scene (the latency of one fleet's round), the other single-GPU configs of BASELINE.json batched (configs[1]: 5 agents
obstacle-free; configs[2]: 8 agents + 20 obstacles), and the DROP-IN call — the per-agent handle behind
include/neptune_poly_solver.hpp, what Neptune::replanCB would call once per replan (neptune.cpp:1504-1528) — timed inside the
library."""
import numpy as np

from . import account as acc


def single_scene(ctx, H):
    from neptune_amd.backend import BatchBackend
    aux_steps, N = ctx.aux_steps, H.N
    b1 = BatchBackend(H.p, H.statics, n_scenes=1, device=ctx.dev)
    d_c1 = b1.to_device(H.com[0]); d_g1 = b1.to_device(H.gue[0])

    def single_step():
        b1.replan(d_c1, d_g1)
        d_c1.copy_(b1.d_commit)
    dt5, ms5, _ = ctx.run_leg(single_step, [b1], aux_steps, max(ctx.args.warmup, 2))
    k1 = {n_: b1.kernel_time_ms(i_)[0] for i_, n_ in ((0, "hull"), (1, "separator"), (2, "qp"), (3, "sequence"))}
    b1.enable_timing(False)
    single = {"value": N * aux_steps / dt5, "unit": "replans/s", "steps": aux_steps, "round_ms": dt5 / aux_steps * 1e3,
              "step_ms": acc.step_quantiles(ms5), "kernel_ms": k1, "solve_us": acc.solve_us_stats(b1),
              "note": "one scene of %d agents per launch sequence: the latency of one bulk-synchronous round of a single fleet and that "
                      "fleet's throughput; `value` at the top keeps %d independent scenes in flight" % (N, H.S)}
    b1.close()
    return single


def small_configs(ctx, H):
    from neptune_amd import dist as ndist, scene
    from neptune_amd.backend import BatchBackend
    aux_steps = ctx.aux_steps
    out = {}
    for name, n_a, n_o, n_sc in (("config2_5_agents", 5, 0, 1024), ("config3_8_agents_20_obstacles", 8, 20, 512)):
        scs = scene.make_scenes(n_a, n_o, range(n_sc), workers=min(n_sc, max(1, ctx.host_cores // 2), 64))
        pc = scs[0]["par"]
        bc = BatchBackend(pc, scs[0]["statics"], n_scenes=n_sc, device=ctx.dev)
        for s_ in range(n_sc):
            if len(scs[s_]["statics"]) != len(scs[0]["statics"]):
                raise SystemExit("%s: scene %d drew another number of static obstacles" % (name, s_))
            bc.set_scene_statics(s_, scs[s_]["statics"])
        com_c, gue_c = ndist.stack_scenes(scs)
        d_cc = bc.to_device(com_c); d_gc = bc.to_device(gue_c)

        def small_step():
            bc.replan(d_cc, d_gc)
            d_cc.copy_(bc.d_commit)
        dtc, msc, _ = ctx.run_leg(small_step, [bc], aux_steps, max(ctx.args.warmup, 2))
        kc = {n_: bc.kernel_time_ms(i_)[0] for i_, n_ in ((0, "hull"), (1, "separator"), (2, "qp"), (3, "sequence"))}
        bc.enable_timing(False)
        solc = bc.solutions()
        out[name] = {"value": n_a * n_sc * aux_steps / dtc, "unit": "replans/s", "steps": aux_steps, "ms_per_step": dtc / aux_steps * 1e3,
                     "scenes_in_flight": n_sc, "replans_per_step": n_a * n_sc, "kernel_ms": kc, "solve_us": acc.solve_us_stats(bc),
                     "ipm_iters_mean": float(solc["stats"]["iters"].mean()), "lines_mean": float(solc["stats"]["n_lines"].mean()),
                     "active_rows": acc.active_summary(bc), **acc.status_counts(solc)}
        bc.close()
    return out


def per_agent_api(ctx, H):
    from neptune_amd import scene
    from neptune_amd.backend import PolySolver, hulls_batch as hulls_of
    N, M = H.N, H.M
    per_agent = {"note": "the six-call drop-in sequence of ONE replan (setInitTrajectory -> setHulls -> setHullsNoInflation -> setEntStateVector -> optimize "
                         "-> generatePwpOut, neptune.cpp:1514-1527) through the per-agent C ABI with host buffers, blocking, as a C++ caller's clock sees it "
                         "(nep_backend_debug_time_sequence: no Python between the calls): one host-to-device copy, separator + QP kernels, one device-to-host "
                         "copy.  The reference's budget for the same call is TimeLimit 0.05 s",
                 "iterations_per_agent": 200}
    for name, n_a, n_o in (("config2_5_agents", 5, 0), ("config3_8_agents_20_obstacles", 8, 20), ("config4_64_agents_20_obstacles", N, M)):
        sc_ = scene.make_scene(n_a, n_o, seed=0) if (n_a, n_o) != (N, M) else H.scene0
        pp = sc_["par"]
        hx_, hn_, h0_, n0_ = hulls_of(sc_["committed"], 0.0, pp.num_pol, pp.T_span, pp.drone_radius)
        us_all, uo_all, st_all = [], [], []
        for aid in range(1, min(n_a, 4) + 1):
            ps_ = PolySolver(pp.num_pol, 3, aid, pp.T_span, pp.pb, pp.weight, 0.5, True)
            ps_.setMaxValues(pp.x_min, pp.x_max, pp.y_min, pp.y_max, pp.z_min, pp.z_max, pp.v_max, pp.a_max, pp.j_max)
            ps_.setMaxRuntime(0.05); ps_.setTetherLength(pp.tether_length); ps_.setStaticObstVert(sc_["statics"])
            g_ = sc_["guesses"][aid - 1]; K_ = int(g_["K"])
            hl_ = [[hx_[j, i, :hn_[j, i]] for i in range(pp.num_pol)] for j in range(n_a) if j != aid - 1]
            h0l_ = [[h0_[j, i, :n0_[j, i]] for i in range(pp.num_pol)] if j != aid - 1 else [] for j in range(n_a)]
            ps_.timeSequence(np.arange(K_ + 1) * pp.T_span, np.array(g_["coeff"])[:, :K_, :], hl_, h0l_, dc=pp.dc, n_iter=20)      # warm
            st_, us_, uo_ = ps_.timeSequence(np.arange(K_ + 1) * pp.T_span, np.array(g_["coeff"])[:, :K_, :], hl_, h0l_, dc=pp.dc, n_iter=200)
            us_all.append(us_); uo_all.append(uo_); st_all.append(int(st_))
            ps_.close()
        us_all = np.concatenate(us_all); uo_all = np.concatenate(uo_all)
        per_agent[name] = {"sequence_ms": {"p50": float(np.percentile(us_all, 50)) * 1e-3, "p99": float(np.percentile(us_all, 99)) * 1e-3, "max": float(us_all.max()) * 1e-3},
                           "optimize_ms": {"p50": float(np.percentile(uo_all, 50)) * 1e-3, "p99": float(np.percentile(uo_all, 99)) * 1e-3},
                           "agents_timed": len(st_all), "status": st_all, "hull_lists": n_a - 1}
    return per_agent
