"""The headline workload: BASELINE "64 agents + 20 static obstacles" (configs[3]'s scene; at N = 1 the configuration the
metric is quoted on), `--scenes` seeded scenes in flight per GPU per step, the agents of every scene block-sharded by id over
the ranks.  One step = one bulk-synchronous round (reference neptune.cpp:1512-1529 per agent; the exchange replaces
neptune_ros.cpp:379-480).  Also the legs that re-time the same step: long_run, launch_order_off, reference_tolerances,
presolve."""
import sys
import time
from types import SimpleNamespace

import numpy as np

from . import account as acc


def setup(ctx, c5=None):
    """scenes, handles, device buffers and the step function -> H (everything the other legs read)"""
    torch, tdist, args = ctx.torch, ctx.tdist, ctx.args
    from neptune_amd import abi, dist as ndist, scene
    from neptune_amd.backend import BatchBackend
    world, rank, dev = ctx.world, ctx.rank, ctx.dev
    H = SimpleNamespace()
    # Weak scaling: args.scenes scenes in flight per GPU, so S = scenes * world scenes in total; the
    # agents of EVERY scene are block-sharded over the ranks (configs[3]: 64 agents, 8 per GPU on 8
    # GPUs), which keeps scenes * agents replans per GPU per step at any N.
    N, M, S = args.agents, args.obstacles, args.scenes * world
    first_local, n_local = ndist.shard(N, world, rank)
    # each rank generates its share of the seeded scenes (seeds 0..S-1 overall), then they are shared
    mine = scene.make_scenes(N, M, range(rank * args.scenes, (rank + 1) * args.scenes),
                             workers=min(args.scenes, max(1, (ctx.host_cores // (4 if c5 is not None and c5.busy() else 2)) // world), 64))
    scene0 = mine[0] if rank == 0 else scene.make_scene(N, M, seed=0)
    if c5 is not None:
        # the config-5 pool has been running beside this one; nothing is timed while host processes are still busy (a first
        # version let it run under the timed legs: the GPU time per step was unchanged, the host's share of a 20-step region
        # went from 1 % to 60 %)
        c5.wait()
    p = scene0["par"]
    # every scene has its own static obstacles (drawn first from its seed, so any rank can rebuild any scene's set)
    all_statics = [mine[s - rank * args.scenes]["statics"] if rank * args.scenes <= s < (rank + 1) * args.scenes
                   else scene.scene_statics(N, M, s, par=p) for s in range(S)]
    statics = all_statics[0]
    com_l, gue_l = ndist.stack_scenes(mine)
    com, gue = ctx.share(com_l, S), ctx.share(gue_l, S)

    sharded_hulls = world > 1 and args.exchange == "hulls"
    native = sharded_hulls and not args.exchange_torch and not args.safety and ctx.dist_backend == "nccl"
    C = args.chunks if (sharded_hulls and not args.safety and S % max(args.chunks, 1) == 0) else 1
    # One GPU: the scenes in flight are INDEPENDENT fleets, so a step may run them as G groups, each a launch sequence of its own on its
    # own HIP stream inside the one captured graph (--groups; north star: "agents shard one-per-stream").  A group's interior-point launch
    # ends with a handful of long solves on an otherwise idle chip, its separator and presolve kernels wait on memory: another group's
    # kernels fill those gaps.  Same kernels, same results per scene; `one_stream` re-times the step as ONE sequence.
    G = args.groups if (world == 1 and not args.frontend and not args.safety and not sharded_hulls and args.groups > 1 and S % args.groups == 0 and not args.no_graph) else 1
    if G > 1:
        C = G
    Sc = S // C
    # one handle per scene chunk (C == 1: all scenes); chunk k holds scenes [k*Sc, (k+1)*Sc)
    bes = [BatchBackend(p, statics, first_local=first_local, n_local=n_local, n_scenes=Sc, device=dev) for _ in range(C)]
    be = bes[0]
    for k, b in enumerate(bes):
        if args.cull_radius is not None:
            b.set_line_cull(args.cull_radius)        # (default: the handle's own — the verified presolve at 4 m, polish pass on)
        for s_ in range(Sc):
            if len(all_statics[k * Sc + s_]) != len(statics):
                raise SystemExit("scene %d drew %d static obstacles instead of %d" % (k * Sc + s_, len(all_statics[k * Sc + s_]), len(statics)))
            b.set_scene_statics(s_, all_statics[k * Sc + s_])
    d_committed = be.to_device(com) if C == 1 else None
    d_guess_c = [bes[k].to_device(np.ascontiguousarray(gue[k * Sc:(k + 1) * Sc, first_local:first_local + n_local])) for k in range(C)]
    d_guess = d_guess_c[0]
    ex = ndist.RoundExchange(S, N, world, rank, device=dev)
    # scene groups on streams (G > 1): every group has its records, its exchange (a copy at one rank) and its stream
    d_com_g = [bes[k].to_device(np.ascontiguousarray(com[k * Sc:(k + 1) * Sc])) for k in range(C)] if G > 1 else None
    ex_g = [ndist.RoundExchange(Sc, N, world, rank, device=dev) for _ in range(C)] if G > 1 else None
    g_streams = [torch.cuda.Stream(device=dev) for _ in range(C)] if G > 1 else None
    if G > 1:
        d_committed = d_com_g[0]; ex = ex_g[0]          # (group 0's: what the one-rank RCCL check and the same-replans comparison use)
    hxs = [ndist.HullExchange(bes[k].hull_block_bytes(), world, rank, device=dev) for k in range(C)] if (sharded_hulls and args.safety) else None
    d_local_c = [bes[k].to_device(np.ascontiguousarray(com[k * Sc:(k + 1) * Sc, first_local:first_local + n_local])) for k in range(C)] if sharded_hulls else None
    d_committed_next = torch.empty_like(d_committed) if args.safety else None
    d_new = torch.empty_like(d_committed) if args.safety else None
    d_accept = torch.zeros(S * N, dtype=torch.int32, device=dev) if args.safety else None
    safety_ev, hull_ev, gather_ev, fe_ev = [], [], [], []
    REC = abi.TRAJ_REC_DTYPE.itemsize
    if args.frontend and world > 1 and not (sharded_hulls and not args.safety):
        raise SystemExit("--frontend with several GPUs needs --exchange hulls and no --safety")
    fe_cfg = scene.frontend_cfg(p, beam_width=args.beam) if args.frontend else None
    fe_starts = ctx.share(np.stack([scene.frontend_starts(s) for s in mine]), S) if args.frontend else None       # [S][N]
    d_fe_start_c = [bes[k].to_device(np.ascontiguousarray(fe_starts[k * Sc:(k + 1) * Sc, first_local:first_local + n_local])) for k in range(C)] if args.frontend else None
    d_fe_start = d_fe_start_c[0] if args.frontend else None
    d_fe_res_c = [torch.zeros(Sc * n_local * abi.FE_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev) for _ in range(C)] if args.frontend else None
    d_fe_res = d_fe_res_c[0] if args.frontend else None
    pending = [None] * C
    _ev_lists = {"hull": hull_ev, "wait": gather_ev, "frontend": fe_ev}
    rounds = None
    nranks = None
    H.steps_made, H.exchange_note = 0, None
    if sharded_hulls and not args.safety:
        def build(nat):
            for k in range(C):                                # (fresh records: a failed first native step may have half-made a round)
                d_local_c[k].copy_(bes[k].to_device(np.ascontiguousarray(com[k * Sc:(k + 1) * Sc, first_local:first_local + n_local])))
            return ndist.ShardedRounds(bes, d_local_c, d_guess_c, world, rank, native=nat,
                                       fe=(fe_cfg, d_fe_start_c, d_fe_res_c) if args.frontend else None,
                                       timer=lambda name: ctx.timed(_ev_lists[name]))
        rounds, native, H.steps_made, H.exchange_note = ctx.rounds_with_fallback(build, native)
        hxs = rounds.hx
        if native:
            nranks = native_nranks(ctx, rounds)
        elif ctx.use_dist and ctx.dist_backend == "nccl":
            nranks = [tdist.get_world_size()] * world
    elif ctx.use_dist and ctx.dist_backend == "nccl":
        nranks = [tdist.get_world_size()] * world      # (N = 1: the one-rank process group this run created)

    def start_exchange(k, src):
        """hulls of my agents' committed trajectories (chunk k) -> start the all-gather of the hull blocks"""
        e0 = ctx.ev()
        bes[k].hulls(src, d_guess_c[k], hxs[k].local)
        hull_ev.append((e0, ctx.ev()))
        pending[k] = hxs[k].gather_async()

    def step_groups():
        cur = torch.cuda.current_stream(dev)
        for k in range(C):
            g_streams[k].wait_stream(cur)
            with torch.cuda.stream(g_streams[k]):
                bes[k].replan(d_com_g[k], d_guess_c[k])
                ex_g[k].gather(bes[k].d_commit, d_com_g[k])
        for k in range(C):
            cur.wait_stream(g_streams[k])

    PIPE = G > 1 and getattr(args, "pipeline", False)
    if PIPE:
        s_geo = torch.cuda.Stream(device=dev)
        s_qp = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(C)]
        e_geo = [torch.cuda.Event() for _ in range(C)]; e_qp = [torch.cuda.Event() for _ in range(C)]
        for k in range(C):
            e_qp[k].record(torch.cuda.current_stream(dev))

    def step_pipelined():
        """every group one round: lines(k) on the geometry stream once solve(k) of the previous round is done, solve(k) on the group's stream
        once its lines are there; nothing else orders the groups (the leg's barriers synchronise the device)"""
        for k in range(C):
            bes[k].enable_timing(False)
            s_geo.wait_event(e_qp[k])
            with torch.cuda.stream(s_geo):
                bes[k].replan_lines(d_com_g[k], d_guess_c[k])
                e_geo[k].record(s_geo)
            s_qp[k].wait_event(e_geo[k])
            with torch.cuda.stream(s_qp[k]):
                bes[k].replan_solve(d_com_g[k], d_guess_c[k])
                ex_g[k].gather(bes[k].d_commit, d_com_g[k])
                e_qp[k].record(s_qp[k])
        torch.cuda.current_stream(dev).wait_event(e_geo[C - 1])      # (the per-step events of the timed region follow the geometry stream)

    def step():
        if PIPE:
            step_pipelined()
            return
        if G > 1:
            step_groups()
            return
        if rounds is not None:
            rounds.step()          # chunks pipelined: one chunk's all-gather runs under another chunk's kernels (dist.ShardedRounds)
            return
        if sharded_hulls:
            start_exchange(0, d_local_c[0])
            e1 = ctx.ev()
            pending[0].wait()
            gather_ev.append((e1, ctx.ev()))
            be.replan_hulls(hxs[0].blocks, d_guess)
        elif args.frontend:
            e0 = ctx.ev()
            be.frontend(fe_cfg, d_committed, d_fe_start, d_guess, d_fe_res)     # hulls + beam search -> d_guess
            fe_ev.append((e0, ctx.ev()))
            be.replan(None, d_guess)                                             # separator + QP on the same hulls
        else:
            be.replan(d_committed, d_guess)
        if not args.safety:
            e1 = ctx.ev()
            ex.gather(be.d_commit, d_committed)
            gather_ev.append((e1, ctx.ev()))
            return
        ex.gather(be.d_commit, d_new)                   # everyone's new trajectory
        e0 = ctx.ev()
        be.safety_commit(d_committed, d_new, d_guess, d_committed_next, d_accept)      # d_guess: [S][n_local], as passed to the replan
        safety_ev.append((e0, ctx.ev()))
        d_committed.copy_(d_committed_next)
        if sharded_hulls:
            d_local_c[0].view(S, n_local * REC).copy_(d_committed.view(S, N * REC)[:, first_local * REC:(first_local + n_local) * REC])

    # a step can be captured when it is a fixed launch sequence without host decisions: one GPU, or several with the native
    # exchange (the torch path's work handles and the gloo / records paths are host-driven)
    can_graph = (world == 1 and rounds is None) or (rounds is not None and native)
    H.__dict__.update(N=N, M=M, S=S, C=C, Sc=Sc, first_local=first_local, n_local=n_local, mine=mine, scene0=scene0, p=p,
                      all_statics=all_statics, statics=statics, com=com, gue=gue, sharded_hulls=sharded_hulls, native=native,
                      bes=bes, be=be, d_committed=d_committed, d_guess=d_guess, ex=ex, rounds=rounds, nranks=nranks, step=step,
                      safety_ev=safety_ev, hull_ev=hull_ev, gather_ev=gather_ev, fe_ev=fe_ev, d_fe_res=d_fe_res, d_accept=d_accept,
                      graph_plain=can_graph and not args.frontend and not args.safety and not PIPE, replans_per_step=S * N, rccl_one_rank_ok=None,
                      G=G, d_com_g=d_com_g, d_guess_c=d_guess_c, PIPE=PIPE)
    return H


def full_handle_view(ctx, H):
    """the legs that need ONE handle over all S scenes (chain, moving, crossing, one_stream): H itself when the headline runs as one launch
    sequence, else a copy of H whose `be` is a handle of all S scenes (the headline's group handles hold S / G scenes each)"""
    if H.G == 1:
        return H
    from neptune_amd.backend import BatchBackend
    be_full = BatchBackend(H.p, H.statics, first_local=H.first_local, n_local=H.n_local, n_scenes=H.S, device=ctx.dev)
    for s_ in range(H.S):
        be_full.set_scene_statics(s_, H.all_statics[s_])
    Hc = SimpleNamespace(**H.__dict__)
    Hc.be, Hc.bes, Hc.C, Hc.Sc, Hc.G = be_full, [be_full], 1, H.S, 1
    Hc.d_guess = be_full.to_device(np.ascontiguousarray(H.gue[:, H.first_local:H.first_local + H.n_local]))
    Hc.d_committed = be_full.to_device(H.com)
    return Hc


def one_stream_leg(ctx, H, Hc):
    """G > 1 only: the headline's step as ONE launch sequence of all S scenes on one stream (rounds 1-5's shape of the step)"""
    from neptune_amd import dist as ndist
    be = Hc.be
    ex = ndist.RoundExchange(H.S, H.N, ctx.world, ctx.rank, device=ctx.dev)

    def step1():
        be.replan(Hc.d_committed, Hc.d_guess)
        ex.gather(be.d_commit, Hc.d_committed)
    dt, ms, _ = ctx.run_leg(step1, [be], ctx.aux_steps, max(ctx.args.warmup, 2), graph_ok=True, eager_after=10)
    k_ = {n_: be.kernel_time_ms(i_)[0] for i_, n_ in ((0, "hull"), (1, "separator"), (2, "qp"), (3, "sequence"))}
    be.enable_timing(False)
    sol = be.solutions()
    return leg_record(H, dt, ctx.aux_steps, ms, kernel_ms=k_, ipm_iters_mean=float(sol["stats"]["iters"].mean()),
                      note="the same step as ONE launch sequence of all %d scenes on one stream (the headline runs it as %d groups of %d scenes on %d streams inside "
                           "one captured graph): what the groups' overlap is worth" % (H.S, H.G, H.Sc, H.G), **acc.status_counts(sol))


def native_nranks(ctx, rounds):
    """ranks that joined the C ABI's own RCCL communicator, as every rank sees it (a communicator that silently came up with one
    rank would time a copy): asserted against WORLD_SIZE"""
    torch, tdist = ctx.torch, ctx.tdist
    mine_n = rounds.native.nranks()
    if ctx.world > 1:
        t = torch.tensor([mine_n], dtype=torch.int64, device=ctx.dev)
        got = [torch.zeros_like(t) for _ in range(ctx.world)]
        tdist.all_gather(got, t)
        nranks = [int(x.item()) for x in got]
    else:
        nranks = [mine_n]
    print("[bench] rank %d of %d: native RCCL communicator has %d ranks" % (ctx.rank, ctx.world, mine_n), file=sys.stderr)
    if mine_n != ctx.world or any(n_ != ctx.world for n_ in nranks):
        raise SystemExit("native RCCL communicator: ranks joined %r, expected %d on every rank" % (nranks, ctx.world))
    return nranks


def leg_record(H, dt_, steps_, step_ms_, **kw):
    r = {"value": H.replans_per_step * steps_ / dt_, "unit": "replans/s", "steps": steps_, "ms_per_step": dt_ / steps_ * 1e3,
         "step_ms": acc.step_quantiles(step_ms_)}
    r.update(kw)
    return r


def run(ctx, H):
    """warm-up, the one-rank RCCL check, then the headline: exactly --steps steps"""
    torch, tdist, args = ctx.torch, ctx.tdist, ctx.args
    import os
    be, bes = H.be, H.bes
    for _ in range(max(args.warmup - H.steps_made, 0)):
        H.step()
    if ctx.world == 1 and ctx.use_dist and ctx.dist_backend == "nccl" and not (args.safety or args.frontend) and (H.C == 1 or H.G > 1):
        # one rank: the timed steps copy (nothing to exchange); the collective path itself — the all-gather of the committed
        # records through RCCL — is exercised once here, outside the timed region, and must give the same bytes
        chk = torch.empty_like(H.d_committed)
        H.ex.gather(be.d_commit, chk, collective=True)
        torch.cuda.synchronize(ctx.dev)
        H.rccl_one_rank_ok = bool(torch.equal(chk, be.d_commit.view_as(chk)))
        if ctx.own_group and os.environ.get("NEP_BENCH_PG_TEARDOWN"):
            # development aid.  With a live RCCL communicator in the process a hipMemsetAsync node in a replayed graph costs
            # ~0.27 ms (found on the config-5 step: 1.55 instead of 1.28 ms with identical kernel times; the presolve's redo
            # counters are now zeroed by a kernel and the step has no memset node): this tears the one-rank group down early
            tdist.destroy_process_group(); ctx.use_dist = False; ctx.rccl_torn_down = True
    dt, step_ms, graph = ctx.run_leg(H.step, bes, args.steps, 0, graph_ok=H.graph_plain, clear=(H.safety_ev, H.hull_ev, H.gather_ev))
    H.dt, H.step_ms, H.graph, H.dt_local = dt, step_ms, graph, ctx.last_wall
    H.qp_ms, H.n_launch = be.kernel_time_ms(2)           # per launch of one chunk (chunk 0)
    H.hull_ms, _ = be.kernel_time_ms(0)
    if H.sharded_hulls:
        H.hull_ms = ctx.mean_ms(H.hull_ev)
    H.sep_ms, _ = be.kernel_time_ms(1)
    H.seq_ms, _ = be.kernel_time_ms(3)
    for b in bes:
        b.enable_timing(False)
    H.sol = np.concatenate([b.solutions() for b in bes])
    H.active = acc.active_summary(be) if (H.C == 1 or H.G > 1) else None          # (group 0's handle when the scenes run as groups)
    H.solve_us = acc.solve_us_stats(be)
    H.n_states = int(H.sol[0]["n_states"])
    H.value = H.replans_per_step * args.steps / dt


def retimed_legs(ctx, H):
    """the same step over a longer timed region, with the launch order off, at the reference solver's tolerances, and with the
    verified presolve -> dict of legs"""
    args, aux_steps, be, bes, step = ctx.args, ctx.aux_steps, H.be, H.bes, H.step
    legs = {"long_run": None, "launch_order_off": None, "reference_tolerances": None, "full_rows": None}
    if not args.no_extra_legs and H.graph_plain:
        dt_l, ms_l, _ = ctx.run_leg(step, bes, aux_steps, 2, graph_ok=True, eager_after=0)
        legs["long_run"] = leg_record(H, dt_l, aux_steps, ms_l, note="the headline's step, %d steps between the barriers" % aux_steps)
        for b in bes:
            b.set_launch_order(False)
        dt_o, ms_o, _ = ctx.run_leg(step, bes, aux_steps, 2, graph_ok=True, eager_after=10)
        qp_o, _ = be.kernel_time_ms(2)
        legs["launch_order_off"] = leg_record(H, dt_o, aux_steps, ms_o, qp_ms=qp_o, solve_us=acc.solve_us_stats(be),
                                              note="QP workgroups in slot order (nep_batch_set_launch_order(0)): what the headline gains from ordering "
                                                   "them by each slot's previous measured time — in this leg and in the headline the same problems are "
                                                   "re-solved every step, so that predictor is exact; `moving` has the realistic one")
        for b in bes:
            b.enable_timing(False); b.set_launch_order(True)
        for _ in range(2):
            step()                                   # (the ordering keys are fresh again for what follows)
        # ---- the same step stopped where the reference's solver stops: Gurobi's default barrier tolerances ---------
        for b in bes:
            b.set_tolerances(1e-6, 1e-8)
        dt_t, ms_t, _ = ctx.run_leg(step, bes, aux_steps, 3, graph_ok=True, eager_after=10)
        qp_t, _ = be.kernel_time_ms(2)
        sol_t = be.solutions()
        legs["reference_tolerances"] = leg_record(
            H, dt_t, aux_steps, ms_t, qp_ms=qp_t, solve_us=acc.solve_us_stats(be), ipm_iters_mean=float(sol_t["stats"]["iters"].mean()),
            residual_tol=1e-6, gap_tol=1e-8,
            note="nep_batch_set_tolerances(1e-6, 1e-8): the strict tests at Gurobi's defaults (FeasibilityTol = OptimalityTol = "
                 "1e-6, BarConvTol = 1e-8), which is where the reference's solver stops (PolySolverGurobi sets OutputFlag and "
                 "TimeLimit only, solver_gurobi_poly.cpp:811-812); the headline and every other leg use the library's defaults, 1e-10 / 1e-11", **acc.status_counts(sol_t))
        for b in bes:
            b.enable_timing(False); b.set_tolerances()
        for _ in range(2):
            step()
    # ---- full_rows: the same steps with the presolve OFF — every separating-line row of every replan through the interior point
    # (rounds 1-5's headline; the handle's default is the verified presolve) ---------------------------------
    if args.cull_radius is None and not args.no_full_rows and not args.no_extra_legs:
        cull_default = be.line_cull()
        # the SAME replans on both paths (a leg's steps feed each other — every step's new trajectories are the next step's obstacles — so
        # the last steps of two legs are not the same problems): one launch of each path against one snapshot of the committed records
        vs_default = None
        if (H.C == 1 or H.G > 1) and not H.sharded_hulls and H.d_committed is not None and not args.frontend and not args.safety:
            snap = H.d_committed.clone()
            be.replan(snap, H.d_guess); sol_a = be.solutions().copy()
            be.set_line_cull(0.0)
            be.replan(snap, H.d_guess); sol_b = be.solutions().copy()
            same = sol_a["stats"]["status"] == sol_b["stats"]["status"]
            okk = same & (sol_a["stats"]["status"] != 2)
            dco = np.abs(np.array(sol_a["coeff"]) - np.array(sol_b["coeff"])).reshape(len(sol_a), -1).max(axis=1)
            vs_default = {"replans": int(len(sol_a)), "status_mismatches": int((~same).sum()), "coeff_diff_max": float(dco[okk].max()) if okk.any() else None,
                          "coeff_diff_above_1e-6": int((dco[okk] > 1e-6).sum()),
                          "note": "one launch of the default path and one of the every-row path against the same snapshot of the committed records: statuses and coefficients"}
        for b in bes:
            b.set_line_cull(0.0)
        dt2, ms2, _ = ctx.run_leg(step, bes, aux_steps, max(args.warmup, 2), graph_ok=H.graph_plain)
        qp2, _ = be.kernel_time_ms(2); hull2, _ = be.kernel_time_ms(0); sep2, _ = be.kernel_time_ms(1); seq2, _ = be.kernel_time_ms(3)
        for b in bes:
            b.enable_timing(False)
        sol2 = np.concatenate([b.solutions() for b in bes])
        legs["full_rows"] = leg_record(
            H, dt2, aux_steps, ms2, cull_radius_m=0.0, qp_kernel=be.qp_kernel_name(),
            kernel_ms={"hull": hull2, "separator": sep2, "qp": qp2, "sequence": seq2},
            rows_solved_mean=float(sol2["stats"]["n_rows"].mean()),
            ipm_iters_mean=float(sol2["stats"]["iters"].mean()), ipm_iters_max=int(sol2["stats"]["iters"].max()),
            solve_us=acc.solve_us_stats(be), active_rows=acc.active_summary(be),
            polish_listed_certified_last_step=list(be.polish_count()),
            vs_default=vs_default,
            note="nep_batch_set_line_cull(0): every separating-line row of every replan through the interior point — the conservative reading "
                 "of the metric that rounds 1-5 quoted as the headline; the default path (verified presolve + polish) returns the same optimum",
            **acc.status_counts(sol2))
        for b in bes:
            b.set_line_cull(cull_default)
        for _ in range(2):
            step()
    return legs


def per_rank_records(ctx, H):
    """every rank's own view of the step (the line's kernel_ms is rank 0's): kernel times and what its stream waited for"""
    if ctx.world == 1:
        return None
    mine_rec = {"rank": ctx.rank, "device": ctx.torch.cuda.get_device_name(ctx.dev),
                "kernel_ms": {"hull": H.hull_ms, "separator": H.sep_ms, "qp": H.qp_ms, "sequence": H.seq_ms, "exchange_wait": ctx.mean_ms(H.gather_ev)},
                "step_ms_p50": float(np.percentile(H.step_ms, 50)), "step_ms_max": float(H.step_ms.max()), "wall_s": H.dt_local}
    per_rank = [None] * ctx.world
    ctx.tdist.all_gather_object(per_rank, mine_rec)
    return per_rank


def record(ctx, H):
    """rank 0: the headline's part of the detail record (contract fields, solver statistics, roofline)"""
    args, world = ctx.args, ctx.world
    be, sol, p, N, M, S, C, Sc, n_local = H.be, H.sol, H.p, H.N, H.M, H.S, H.C, H.Sc, H.n_local
    status = sol["stats"]["status"].astype(int)
    iters = sol["stats"]["iters"].astype(int)
    if C == 1 and not H.sharded_hulls and not args.frontend:
        _, hn = be.debug_hulls(0)          # vertex counts of scene 0 as the last timed launch saw them
    else:
        from neptune_amd.backend import hulls_batch
        _, hn, _, _ = hulls_batch(H.com[0], float(H.gue[0, 0]["t_start"]), p.num_pol, p.T_span, p.drone_radius)   # scene 0 at the start
    bytes_per_replan = acc.algorithmic_bytes(p, H.scene0, hn, H.n_states)
    launch_replans = Sc * n_local
    qp_ms, hull_ms, sep_ms, seq_ms = H.qp_ms, H.hull_ms, H.sep_ms, H.seq_ms
    # the dominant kernel of the launch sequence by ITS measured duration (HIP events on the launch stream, nep_batch_kernel_time)
    cull_m = be.line_cull()
    k_ms = {"hull": hull_ms, "separator": sep_ms, "qp": qp_ms}
    dom = max(k_ms, key=lambda n_: k_ms[n_])
    dom_ms = k_ms[dom]
    dom_name = {"hull": "hull_group_kernel" if Sc * (n_local if H.sharded_hulls else N) > 2048 else "hull_kernel",
                "separator": "separator_packed_kernel" if cull_m > 0.0 else "separator_kernel", "qp": be.qp_kernel_name()}[dom]
    achieved = bytes_per_replan * launch_replans / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    K8 = int(sol[0]["K"])
    # the same compulsory bytes split by the kernel that moves them (per replan), each over its own duration
    Kg = int(H.scene0["guesses"][0]["K"]); L_mean = float(sol["stats"]["n_lines"].mean())
    b_guess = 8 * (12 * Kg + (Kg + 1)); b_hull_v = 16.0 * hn[:, :Kg].sum() / N          # one agent's hull vertices
    b_rec = 8 * (13 * Kg + 1)                                                           # one committed trajectory
    b_static = 16 * sum(len(s_) for s_ in H.statics); b_out = 8 * (12 * Kg + 1) + 4 + 96 * H.n_states
    per_kernel_bytes = {"hull": (b_rec + b_hull_v) * (Sc * (n_local if H.sharded_hulls else N)) / launch_replans,   # records in, hull vertices out
                        "separator": b_guess + b_hull_v * (N - 1) + b_static + 16 * N + 24 * L_mean,          # every other agent's hulls in, lines out
                        "qp": b_guess + 24 * L_mean + b_out}
    per_kernel = {}
    for name, ms_k in (("hull", hull_ms), ("separator", sep_ms), ("qp", qp_ms)):
        gbs = per_kernel_bytes[name] * launch_replans / (ms_k * 1e-3) / 1e9 if ms_k > 0 else 0.0
        per_kernel[name] = {"bytes_per_replan": per_kernel_bytes[name], "ms": ms_k, "GB/s": gbs, "frac": gbs / 8000.0}
    seq_gbs = bytes_per_replan * launch_replans / (seq_ms * 1e-3) / 1e9 if seq_ms > 0 else 0.0
    flops = acc.algorithmic_flops(K8, L_mean, float(hn[:, :K8][hn[:, :K8] > 0].mean()) if (hn[:, :K8] > 0).any() else 4.0, float(iters.mean()))
    fp64_ach = flops * launch_replans / (qp_ms * 1e-3) / 1e12 if qp_ms > 0 else 0.0
    fp64 = {"bound": "fp64 vector (reported next to the HBM roofline, SURVEY 8d)", "achieved": fp64_ach, "peak": 78.6, "unit": "TFLOP/s",
            "frac": fp64_ach / 78.6, "algorithmic_flops_per_replan": flops,
            "note": "an upper bound: SURVEY 8d's count assumes the dense G'WG product; the kernel's structured assembly over 64 base rows executes fewer "
                    "(executed_*: the fp64 instructions the kernel issues, from the hardware counters of profiles/pmc_summary_latest.txt)"}
    ex_f = acc.executed_flops("nep::" + be.qp_kernel_name()) if launch_replans == 8192 else None
    if ex_f and qp_ms > 0:
        # the fp64 work the kernel ISSUES per launch (hardware instruction counters of the committed PMC summary of this command,
        # 8 192 replans per launch) over THIS run's kernel duration — next to the SURVEY-count figure above, which prices a dense product
        fp64["executed_flops_per_replan"] = ex_f["flops_per_launch"] / launch_replans
        fp64["executed_achieved"] = ex_f["flops_per_launch"] / (qp_ms * 1e-3) / 1e12
        fp64["executed_frac"] = fp64["executed_achieved"] / 78.6
        if fp64["executed_frac"] > 1.0:      # (the committed counter summary is of another build's kernel — e.g. the every-row solve — than the one this run timed)
            fp64["executed_note"] = "profiles/pmc_summary_latest.txt does not belong to this build's kernel (its instruction count over this run's duration exceeds the peak): ignored"
            fp64["executed_achieved"] = None; fp64["executed_frac"] = None
        fp64["executed_mfma_share"] = ex_f["mfma_flops_per_launch"] / ex_f["flops_per_launch"] if ex_f["flops_per_launch"] > 0 else None
        fp64["executed_source"] = ex_f["source"]
    if world == 1:
        sharding = "one GPU: all %d agents of every scene" % N
        if H.G > 1:
            sharding += ("; the %d scenes in flight run as %d groups of %d scenes, pipelined on streams: the groups' geometry halves in turn on one stream, each group's QP half on its own, ordered by events" % (S, H.G, Sc)
                         if getattr(H, "PIPE", False) else
                         "; the %d scenes in flight run as %d groups of %d scenes, each group a launch sequence on its own HIP stream, all inside one captured graph per step" % (S, H.G, Sc))
    elif H.sharded_hulls:
        sharding = ("agents of every scene block-sharded by id, %d per GPU; per step and scene chunk (%d chunks, pipelined) one all-gather "
                    "(RCCL, %s) of the interval hulls of the local agents' committed trajectories (%d B per agent and scene)"
                    % (n_local, C, "native binding on a side stream inside the captured step" if H.native else "torch.distributed", be.hull_block_bytes() // (Sc * n_local)))
    else:
        sharding = "agents of every scene block-sharded by id, %d per GPU; per step one all-gather (RCCL) of the committed trajectory records" % n_local
    from neptune_amd import abi
    fe_res = H.d_fe_res.cpu().numpy().view(abi.FE_RESULT_DTYPE) if args.frontend else None
    return {
        "metric": "backend_replans_per_sec", "value": H.value, "unit": "replans/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": H.dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%d agents + %d static obstacles, K=8, %d seeded scenes in flight per GPU per step (seeds 0..%d over %d GPU%s); %s"
                               % (N, M, args.scenes, S - 1, world, "" if world == 1 else "s",
                                  (("the handle's default solve path: verified line presolve %g m, polish on" % cull_m) if cull_m > 0.0 else
                                   "line presolve off: every separating-line row through the interior point, polish on")
                                  + (("; %d scene groups pipelined on streams" % H.G) if getattr(H, "PIPE", False) else ("; %d scene groups on %d HIP streams" % (H.G, H.G) if H.G > 1 else ""))),
                   "agents": N, "obstacles": M, "scenes_in_flight": S, "scenes_per_gpu": args.scenes,
                   "replans_per_step": H.replans_per_step, "replans_per_gpu_per_step": S * n_local,
                   "sharding": sharding, "scene_groups": H.G,
                   "params": "reference neptune_mtlp_benchmark.yaml (T_span 0.5, num_pol 8, weight 1000, v 2, a 3)"},
        "solver": {"status_ok": int((status == 0).sum()), "status_relaxed": int((status == 1).sum()),
                   "status_failed": int((status == 2).sum()), "ipm_iters_mean": float(iters.mean()),
                   "ipm_iters_quantiles": {"p50": float(np.percentile(iters, 50)), "p90": float(np.percentile(iters, 90)),
                                           "p99": float(np.percentile(iters, 99)), "max": int(iters.max())},
                   "ipm_iters_mean_by_status": {name: (float(iters[status == k].mean()) if (status == k).any() else None)
                                                for k, name in ((0, "ok"), (1, "relaxed"), (2, "failed"))},
                   "lines_mean": L_mean, "lp_failed": int(sol["stats"]["n_lp_failed"].sum()),
                   "rows_solved_mean": float(sol["stats"]["n_rows"].mean()), "line_cull_radius_m": be.line_cull(), "solved_without_iteration": int((iters == 0).sum()),
                   "presolve_redo_last_step": be.redo_count(), "polish_listed_certified_last_step": list(be.polish_count()),
                   "lp_failed_note": "separator LPs without a separating line: the constraint is skipped as in the reference "
                                     "(solver_gurobi_poly.cpp:483-494).  Round 0 has none (scenes are sampled so that every LP is feasible); "
                                     "later rounds replan the same guesses against the others' optimised trajectories, which may cross them",
                   "active_rows": H.active},
        # Per-replan latency by SURVEY 8(d)'s definition of one replan (setters + separator loop + QP solve + generatePwpOut).  In a batch
        # every replan completes with its launch sequence, so a batched replan's latency IS the sequence's duration (hull + lines + QP
        # + polish of replans_per_launch replans: batch_sequence_ms); bench.py replaces these two by the blocking six-call drop-in
        # sequence of ONE replan through the per-agent C ABI (per_agent_api, config-4 size) when that leg ran.  The interior-point
        # workgroup's own device time — what rounds 1-5 printed here — is qp_workgroup_us.
        "p50_solve_ms": seq_ms + (hull_ms if H.sharded_hulls else 0.0), "p99_solve_ms": float(np.percentile(H.step_ms, 99)),
        "solve_ms_definition": "one batched replan completes with its launch sequence: p50 = the sequence's HIP-event duration (hulls + separating lines + "
                               "QP + polish of %d replans), p99 = the step's p99" % launch_replans,
        "qp_workgroup_us": dict(H.solve_us, note="device time of each replan's interior-point workgroup alone (setup + interior point + outputs, nep_stats.solve_us); "
                                                 "hulls (%.3f ms per launch) and separating lines (%.3f ms) are shared by the batch" % (hull_ms, sep_ms)),
        "batch_sequence_ms": seq_ms + (hull_ms if H.sharded_hulls else 0.0),
        "step_ms": acc.step_quantiles(H.step_ms),
        "kernel_ms": {"hull": hull_ms, "separator": sep_ms, "qp": qp_ms, "sequence": seq_ms, "exchange_wait": ctx.mean_ms(H.gather_ev),
                      "launches": H.n_launch, "launches_per_step": C},
        "launch": ("one captured HIP graph per step, replayed (per-kernel events from %d eager steps after the timed region)" % min(args.steps, 40)
                   if H.graph is not None else ("host launches: two enqueues per scene group and step (nep_batch_replan_lines / _solve) on the groups' streams"
                                                if getattr(H, "PIPE", False) else ("; ".join(ctx.graph_notes) or "host launches"))),
        "frontend": ({"ms": ctx.mean_ms(H.fe_ev), "beam_width": args.beam,
                      "status_goal_reached": int((fe_res["status"] == 1).sum()), "status_no_solution": int((fe_res["status"] == 3).sum()),
                      "children_mean": float(fe_res["n_children"].mean())} if args.frontend else None),
        "safety": ({"ms": ctx.mean_ms(H.safety_ev), "accepted_frac": float(H.d_accept.float().mean().item())} if args.safety else None),
        "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                     "frac": achieved / 8000.0,
                     # the committed PMC summary is of the default single-GPU command (8 192 replans per launch): a replay of that file
                     "traffic": acc.measured_traffic("nep::" + dom_name) if launch_replans == 8192 else None,
                     "traffic_source": "profiles/pmc_summary_latest.txt (committed rocprofv3 --pmc summary of this command; counters cannot be read in-process)",
                     "algorithmic_bytes_per_replan": bytes_per_replan, "replans_per_launch": launch_replans, "kernel_ms": dom_ms,
                     # the same kernel priced on the bytes IT owns (its own inputs and outputs, per_kernel below) — with the presolve the whole-replan
                     # count above prices hull vertices that no kernel reads any more (a 32-byte box decides a far obstacle), so `frac` is the
                     # contract's quotient (an effective rate), `own` the statement about this kernel and the memory system
                     "own": {"bytes_per_replan": per_kernel_bytes[dom], "achieved": per_kernel[dom]["GB/s"], "frac": per_kernel[dom]["frac"]},
                     "sequence": {"achieved": seq_gbs, "frac": seq_gbs / 8000.0, "ms": seq_ms,
                                  "note": "the whole replan's bytes over the whole launch sequence (hull + separator + qp)"},
                     "per_kernel": per_kernel,
                     "note": "achieved = the whole replan's algorithmic bytes (SURVEY 8d) x replans per launch / the dominant kernel's duration (%s, by "
                             "its HIP-event time), as the contract defines it; `own` and per_kernel give each kernel's own bytes over its own time, `traffic` "
                             "the HBM bytes the counters saw.  Not a memory-bound path: fp64 VALU issue and dependent latency (roofline_fp64); %.2f "
                             "interior-point iterations per replan on average, %d of %d replans solved without one (their unconstrained minimiser "
                             "satisfies every row)" % (dom_name, float(iters.mean()), int((iters == 0).sum()), len(iters))},
        "per_gpu_value": H.value / world,
        "rccl": dict(rccl_record(ctx, H.nranks, H.native, H.rccl_one_rank_ok), **({"note": H.exchange_note} if H.exchange_note else {})),
        "roofline_fp64": fp64,
        "reference_budget": "reference TimeLimit 0.05 s/solve, replan timer 20 Hz/agent => <= %d replans/s for %d agents" % (20 * N, N),
    }


def rccl_record(ctx, nranks, native, one_rank_ok=None):
    world = ctx.world
    if (ctx.use_dist or ctx.rccl_torn_down) and ctx.dist_backend == "nccl":
        return {"process_group": "nccl (RCCL), world %d" % world, "initialised": True, "one_rank_all_gather_matches": one_rank_ok,
                "nranks": nranks, "exchange": ("native (nep_batch_exchange_hulls: ncclAllGather inside the captured step)" if native else
                                               ("torch.distributed" if world > 1 else "none (one rank)")),
                **({"torn_down_before_timing": True} if ctx.rccl_torn_down else {})}
    return {"initialised": False, "note": ctx.rccl_note or ctx.dist_backend}
