"""The rows either side of the back end, on the device, single GPU: `chain` (front-end beam search -> separating lines -> QP ->
safety check + commit) and the closed loops `moving` / `crossing` (chain + point A of the next
round from the committed trajectories, reference neptune.cpp:1366-1399) — every step poses NEW problems there."""
import numpy as np

from . import account as acc
from .headline import leg_record


def run(ctx, H):
    """-> (chain, moving, crossing) records; needs the headline's handle (one chunk, one GPU)"""
    torch, args, aux_steps, dev = ctx.torch, ctx.args, ctx.aux_steps, ctx.dev
    from neptune_amd import abi, scene
    from neptune_amd.backend import BatchBackend
    be, p, N, S, com, mine = H.be, H.p, H.N, H.S, H.com, H.mine
    cfg_fe = scene.frontend_cfg(p, beam_width=args.beam)
    cull_default = be.line_cull()
    cull_c = cull_default if args.chain_cull_radius is None else args.chain_cull_radius
    be.set_line_cull(cull_c)
    starts_np = np.stack([scene.frontend_starts(s_) for s_ in mine])
    d_st = be.to_device(starts_np)
    d_gfe = torch.zeros_like(H.d_guess)
    d_res = torch.zeros(S * N * abi.FE_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    d_com2 = be.to_device(com); d_nxt = torch.empty_like(d_com2); d_acc = torch.zeros(S * N, dtype=torch.int32, device=dev)
    fe2, sf2 = [], []

    def chain_step():
        e0 = ctx.ev()
        be.frontend(cfg_fe, d_com2, d_st, d_gfe, d_res)
        fe2.append((e0, ctx.ev()))
        be.replan(None, d_gfe)                       # (a failed / empty replan's commit slot carries the record of d_com2 over)
        e1 = ctx.ev()
        be.safety_commit(d_com2, be.d_commit, d_gfe, d_nxt, d_acc)
        sf2.append((e1, ctx.ev()))
        d_com2.copy_(d_nxt)
    dt3, ms3, _ = ctx.run_leg(chain_step, [be], aux_steps, max(args.warmup, 2), clear=(fe2, sf2))
    qp3, _ = be.kernel_time_ms(2); sep3, _ = be.kernel_time_ms(1)
    be.enable_timing(False)
    sol3 = be.solutions()
    res3 = d_res.cpu().numpy().view(abi.FE_RESULT_DTYPE)
    chain = leg_record(H, dt3, aux_steps, ms3,
                       kernel_ms={"frontend_with_hulls": ctx.mean_ms(fe2), "separator": sep3, "qp": qp3, "safety": ctx.mean_ms(sf2)},
                       beam_width=args.beam, frontend_goal_reached=int((res3["status"] == 1).sum()), frontend_no_solution=int((res3["status"] == 3).sum()),
                       ipm_iters_mean=float(sol3["stats"]["iters"].mean()), ipm_iters_max=int(sol3["stats"]["iters"].max()),
                       lp_failed=int(sol3["stats"]["n_lp_failed"].sum()), accepted_frac=float(d_acc.float().mean().item()),
                       solve_us=acc.solve_us_stats(be), terminal_ball_rows=int(sol3["stats"]["qc_active"].sum()),
                       ipm_iters_quantiles=acc.quantiles(sol3["stats"]["iters"]), line_cull_radius_m=cull_c,
                       rows_solved_mean=float(sol3["stats"]["n_rows"].mean()), presolve_redo_last_step=be.redo_count(),
                       polish_listed_certified_last_step=list(be.polish_count()), active_rows=acc.active_summary(be),
                       note="front-end beam search -> separating lines -> QP -> safety check + commit, every step; the guesses are the "
                            "device-made lattice paths (they end at cruise speed and cut corners around obstacles), not the scene's; "
                            "point A stays where it is, so after a few steps every step poses the same problems", **acc.status_counts(sol3))

    # ---- moving: the closed loop on the device.  After the commit, point A of the next round is taken half a second
    # (T_span: one interval) ahead on every agent's committed trajectory (nep_batch_next_starts) and an agent that has
    # arrived swaps its goal with its starting point, so the fleets keep flying: every step poses new problems, and the
    # launch-order key of a slot is the measured time of the SAME AGENT's previous, different replan --------------
    cfg_mv = scene.frontend_cfg(p, beam_width=args.beam, pad_hold=1)

    def closed_loop(starts_in, com_in, note):
        """one closed-loop leg from the given points A / goals and committed records -> run(cull) -> record"""
        d_st_m = be.to_device(starts_in)
        alt_np = np.ascontiguousarray(starts_in["pos"].reshape(S * N, 3))          # the way back: where the agent started
        d_alt = torch.from_numpy(alt_np.copy()).to(dev)
        d_com3 = be.to_device(com_in); d_nxt3 = torch.empty_like(d_com3)
        fe3, sf3 = [], []

        def moving_step():
            e0 = ctx.ev()
            be.frontend(cfg_mv, d_com3, d_st_m, d_gfe, d_res)
            fe3.append((e0, ctx.ev()))
            be.replan(None, d_gfe)
            e1 = ctx.ev()
            be.safety_commit(d_com3, be.d_commit, d_gfe, d_nxt3, d_acc)
            d_com3.copy_(d_nxt3)
            be.next_starts(d_com3, p.T_span, d_st_m, d_alt, 0.5)
            sf3.append((e1, ctx.ev()))

        def run_(cull):
            be.set_line_cull(cull)
            d_st_m.copy_(be.to_device(starts_in)); d_alt.copy_(torch.from_numpy(alt_np.copy()).to(dev)); d_com3.copy_(be.to_device(com_in))
            dt4, ms4, _ = ctx.run_leg(moving_step, [be], aux_steps, max(args.warmup, 2), clear=(fe3, sf3))
            qp4, _ = be.kernel_time_ms(2); sep4, _ = be.kernel_time_ms(1)
            be.enable_timing(False)
            sol4 = be.solutions(); res4 = d_res.cpu().numpy().view(abi.FE_RESULT_DTYPE)
            st_now = d_st_m.cpu().numpy().view(abi.FE_START_DTYPE)
            moved = np.hypot(*(st_now["pos"][:, :2] - starts_in.reshape(-1)["pos"][:, :2]).T)
            swaps = int((np.abs(st_now["goal"] - starts_in.reshape(-1)["goal"]).max(axis=1) > 0).sum())
            return leg_record(H, dt4, aux_steps, ms4,
                              kernel_ms={"frontend_with_hulls": ctx.mean_ms(fe3), "separator": sep4, "qp": qp4, "safety_commit_next_start": ctx.mean_ms(sf3)},
                              beam_width=args.beam, frontend_goal_reached=int((res4["status"] == 1).sum()), frontend_no_solution=int((res4["status"] == 3).sum()),
                              ipm_iters_mean=float(sol4["stats"]["iters"].mean()), ipm_iters_max=int(sol4["stats"]["iters"].max()),
                              lp_failed=int(sol4["stats"]["n_lp_failed"].sum()), accepted_frac=float(d_acc.float().mean().item()),
                              K_mean=float(sol4["K"].mean()), solve_us=acc.solve_us_stats(be),
                              terminal_ball_rows=int(sol4["stats"]["qc_active"].sum()), lines_mean=float(sol4["stats"]["n_lines"].mean()),
                              ipm_iters_quantiles=acc.quantiles(sol4["stats"]["iters"]), line_cull_radius_m=cull,
                              rows_solved_mean=float(sol4["stats"]["n_rows"].mean()), presolve_redo_last_step=be.redo_count(), polish_listed_certified_last_step=list(be.polish_count()),
                              simulated_seconds=float(st_now["t_start"].max() - starts_in["t_start"].max()),
                              displacement_m_mean=float(moved.mean()), agents_with_swapped_goal=swaps,
                              failed_frac=float((sol4["stats"]["status"] == 2).mean()), active_rows=acc.active_summary(be),
                              note=note, **acc.status_counts(sol4))
        return run_
    run_moving = closed_loop(starts_np, com,
                             "closed loop on the device, one HIP graph per round: front end -> lines -> QP -> safety check + commit -> point A of "
                             "the next round 0.5 s ahead on the committed trajectory; arrived agents turn around.  Every step solves NEW problems; the "
                             "launch-order predictor is the same agent's previous replan")
    moving = run_moving(cull_c)
    # ---- crossing: the same closed loop on the hard variant of every scene — all 64 agents start at rest on the base circle
    # and fly to the antipodal point, so the whole fleet meets in the middle (and turns around on arrival) --------------
    cross = [scene.crossing_scene(s_) for s_ in mine]
    run_cross = closed_loop(np.stack([c_[0] for c_ in cross]), np.stack([c_[1] for c_ in cross]),
                            "the closed loop of `moving` on the circle-swap variant of the same scenes: every agent starts at rest on the base circle, "
                            "its goal is the antipodal point (the start of the agent opposite), arrived agents turn around — the fleet crosses the middle of "
                            "the world together, against the scene's static obstacles.  The hard leg: see active_rows, failed_frac, ipm_iters")
    crossing = run_cross(cull_c)
    # ---- moving, as two scene groups on two streams inside the one captured step: the tail of one group's kernels (the QP
    # launch ends with a handful of failing solves of ~1.2 ms each on an otherwise empty GPU) runs beside the other group's
    # kernels.  Same scenes, same results; what a deployment that keeps several fleets in flight does ------------------
    if S % 2 == 0 and not args.no_graph:
        Sg = S // 2
        gb = []
        for k_ in range(2):
            b_ = BatchBackend(p, H.statics, n_scenes=Sg, device=dev)
            for s_ in range(Sg):
                b_.set_scene_statics(s_, H.all_statics[k_ * Sg + s_])
            b_.set_line_cull(cull_c)
            gb.append(b_)
        g_st = [gb[k_].to_device(np.ascontiguousarray(starts_np[k_ * Sg:(k_ + 1) * Sg])) for k_ in range(2)]
        g_alt = [torch.from_numpy(np.ascontiguousarray(starts_np[k_ * Sg:(k_ + 1) * Sg]["pos"].reshape(Sg * N, 3)).copy()).to(dev) for k_ in range(2)]
        g_com = [gb[k_].to_device(np.ascontiguousarray(com[k_ * Sg:(k_ + 1) * Sg])) for k_ in range(2)]
        g_nxt = [torch.empty_like(g_com[k_]) for k_ in range(2)]
        g_gfe = [torch.zeros(Sg * N * abi.GUESS_DTYPE.itemsize, dtype=torch.uint8, device=dev) for _ in range(2)]
        g_res = [torch.zeros(Sg * N * abi.FE_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev) for _ in range(2)]
        g_acc = [torch.zeros(Sg * N, dtype=torch.int32, device=dev) for _ in range(2)]
        g_streams = [torch.cuda.Stream(device=dev) for _ in range(2)]

        RG = 4      # rounds of each group per captured graph: the groups drift apart inside it, and a replay's join is paid once per four rounds

        def two_group_step():
            cur_ = torch.cuda.current_stream(dev)
            fe_done = None
            for k_ in range(2):
                g_streams[k_].wait_stream(cur_)
                if fe_done is not None:
                    g_streams[k_].wait_event(fe_done)      # the second group starts when the first one's front end is done: its front end beside the first's back end
                with torch.cuda.stream(g_streams[k_]):
                    for r_ in range(RG):
                        gb[k_].frontend(cfg_mv, g_com[k_], g_st[k_], g_gfe[k_], g_res[k_])
                        if r_ == 0 and k_ == 0:
                            fe_done = torch.cuda.Event(); fe_done.record(g_streams[k_])
                        gb[k_].replan(None, g_gfe[k_])
                        gb[k_].safety_commit(g_com[k_], gb[k_].d_commit, g_gfe[k_], g_nxt[k_], g_acc[k_])
                        g_com[k_].copy_(g_nxt[k_])
                        gb[k_].next_starts(g_com[k_], p.T_span, g_st[k_], g_alt[k_], 0.5)
            for k_ in range(2):
                cur_.wait_stream(g_streams[k_])
        n_rep = max(aux_steps // RG, 10)
        dtg, msg, gg = ctx.run_leg(two_group_step, gb, n_rep, max(args.warmup, 2), eager_after=0)
        solg = np.concatenate([b_.solutions() for b_ in gb])
        moving["two_groups"] = {"value": H.replans_per_step * RG * n_rep / dtg, "unit": "replans/s", "rounds": RG * n_rep, "ms_per_round": dtg / (RG * n_rep) * 1e3,
                                "rounds_per_graph": RG, "graph": gg is not None, "ipm_iters_mean": float(solg["stats"]["iters"].mean()),
                                "note": "the moving leg with the scenes in two groups of %d on two streams, %d rounds of each group inside one captured graph, the "
                                        "second group started when the first one's front end is done: one group's QP tail (a handful of failing solves on an "
                                        "otherwise empty GPU) runs beside the other group's front end" % (Sg, RG), **acc.status_counts(solg)}
        for b_ in gb:
            b_.close()
    # ---- all three again with the presolve OFF: every row through the interior point (rounds 1-5's way of quoting these legs) ----
    if args.chain_cull_radius is None and not args.no_full_rows:
        be.set_line_cull(0.0)
        d_com2.copy_(be.to_device(com))
        dt3p, ms3p, _ = ctx.run_leg(chain_step, [be], aux_steps, max(args.warmup, 2), clear=(fe2, sf2))
        qp3p, _ = be.kernel_time_ms(2); sep3p, _ = be.kernel_time_ms(1)
        be.enable_timing(False)
        sol3p = be.solutions()
        chain["full_rows"] = leg_record(H, dt3p, aux_steps, ms3p, cull_radius_m=0.0,
                                        kernel_ms={"frontend_with_hulls": ctx.mean_ms(fe2), "separator": sep3p, "qp": qp3p, "safety": ctx.mean_ms(sf2)},
                                        rows_solved_mean=float(sol3p["stats"]["n_rows"].mean()), ipm_iters_mean=float(sol3p["stats"]["iters"].mean()),
                                        polish_listed_certified_last_step=list(be.polish_count()), **acc.status_counts(sol3p))
        moving["full_rows"] = run_moving(0.0)
        crossing["full_rows"] = run_cross(0.0)
    be.set_line_cull(cull_default)
    return chain, moving, crossing
