"""BASELINE configs[4]: 256 agents + 100 static obstacles, enable_entangle_check on (synthetic ent_state per SURVEY §8d: one
active case for 10 % of the agent pairs, 2-4 bend points per tether).

  leg()       the single-GPU legs of the default command: the handle's default (verified presolve), full_rows, chain
  workload()  `bench.py --workload config5 [--gpus N]`: the N-GPU step — 256 / N agents of every scene per GPU, one all-gather
              of the interval hulls (with the tethers' samples and bend points in the blocks) per round and scene chunk; the
              exchange replaces reference neptune_ros.cpp:379-480
"""
import dataclasses
import hashlib
import os
import time

import numpy as np

from . import account as acc


def _config5_scene(job):
    """pool worker: one scene with its synthetic entangle inputs"""
    n_agents, n_static, seed = job
    from neptune_amd import scene
    sc = scene.make_scene(n_agents, n_static, seed=seed)
    case_id = scene.synthetic_entangle(sc, seed=1000 + seed, frac=0.1)
    return sc, case_id


class ScenePool:
    """config-5 scenes take ~20 s of host time each: a pool of host processes makes them while the GPU runs other legs, and is
    awaited before anything is timed.  NEP_BENCH_SCENE_CACHE: development aid (profiling scripts call bench.py several times on
    one box)."""

    def __init__(self, n_agents, n_static, seeds, cores, cache=None):
        self.made, self.wait_s, self.pool, self.futs = None, 0.0, None, None
        self.cache = cache
        seeds = list(seeds)
        if cache and os.path.exists(cache):
            import pickle
            self.made = pickle.load(open(cache, "rb"))
            if len(self.made) != len(seeds):
                self.made = None
        if self.made is None:
            import multiprocessing as mp
            from concurrent.futures import ProcessPoolExecutor
            self.pool = ProcessPoolExecutor(max_workers=min(len(seeds), max(1, cores)), mp_context=mp.get_context("spawn"))
            self.futs = [self.pool.submit(_config5_scene, (n_agents, n_static, s)) for s in seeds]

    def busy(self):
        return self.pool is not None

    def wait(self):
        if self.pool is not None:
            t0 = time.perf_counter()
            self.made = [f.result() for f in self.futs]
            self.pool.shutdown(); self.pool = None
            self.wait_s = time.perf_counter() - t0
            if self.cache and not os.path.exists(self.cache):
                import pickle
                pickle.dump(self.made, open(self.cache, "wb"))
        return self.made


def _carry_ranges():
    """byte ranges of a record the QP kernel's commit slot supplies (position, polynomial); id / flags / bend points are the
    scene's and stay: a committed record as the QP kernel writes it carries the base only"""
    from neptune_amd import abi
    f5 = abi.TRAJ_REC_DTYPE.fields
    o_pos, o_bend, o_pwp = f5["pos"][1], f5["bend"][1], f5["pwp"][1]
    return ((o_pos, o_bend), (o_pwp, abi.TRAJ_REC_DTYPE.itemsize))


def _ent_bytes(N, bend_counts):
    return 4.0 * 8 * N + 16.0 * bend_counts.sum()                # the dense case block of one replan + every agent's bend points


def leg(ctx, made, t_wait):
    """rank 0, one GPU: the config-5 step with the handle's defaults, with every row through the interior point, and the chain"""
    torch, args, aux_steps, dev = ctx.torch, ctx.args, ctx.aux_steps, ctx.dev
    from neptune_amd import abi, dist as ndist, scene
    from neptune_amd.backend import BatchBackend
    S5, N5 = len(made), made[0][0]["par"].num_agents
    sc5 = [m[0] for m in made]
    p5 = dataclasses.replace(sc5[0]["par"], enable_entangle=True)
    b5 = BatchBackend(p5, sc5[0]["statics"], n_scenes=S5, device=dev)
    for s_ in range(S5):
        b5.set_scene_statics(s_, sc5[s_]["statics"])
    com5, gue5 = ndist.stack_scenes(sc5)
    case5 = np.stack([m[1] for m in made])                       # [S5][N][8][N] int32 (bend points are in the records)
    d_c5 = b5.to_device(com5); d_g5 = b5.to_device(gue5)
    d_e5 = torch.from_numpy(np.ascontiguousarray(case5).reshape(-1)).to(dev)
    bend5 = com5["n_bend"].astype(np.float64)

    # the new trajectories are the next step's obstacles, as in the headline; the tethers' bend points are inputs of the
    # scene and stay: position and polynomial are copied over, id / flags / bend points are left alone
    ranges = _carry_ranges()
    REC5 = abi.TRAJ_REC_DTYPE.itemsize
    v_c5 = d_c5.view(S5 * N5, REC5)

    def c5_step():
        b5.replan(d_c5, d_g5, d_ent=d_e5)
        cm = b5.d_commit.view(S5 * N5, REC5)
        for lo, hi in ranges:
            v_c5[:, lo:hi].copy_(cm[:, lo:hi])

    def c5_leg():
        steps_ = aux_steps if not args.config5_only else args.steps
        dt_, ms_, _ = ctx.run_leg(c5_step, [b5], steps_, max(args.warmup, 2), eager_after=10)
        k_ = {n_: b5.kernel_time_ms(i_)[0] for i_, n_ in ((0, "hull"), (1, "separator"), (2, "qp"), (3, "sequence"))}
        b5.enable_timing(False)
        return dt_, steps_, ms_, k_, b5.solutions()
    cull5 = b5.line_cull(); kern5 = b5.qp_kernel_name()
    dt6, steps6, ms6, k6, sol6 = c5_leg()
    us6 = acc.solve_us_stats(b5)
    redo6 = {"replans": b5.redo_count(), **b5.redo_reasons}
    _, hn5 = b5.debug_hulls(0)
    ns5 = int(sol6[0]["n_states"])
    bytes5 = acc.algorithmic_bytes(p5, sc5[0], hn5, ns5, ent_bytes=_ent_bytes(N5, bend5[0]))
    dom = max(("hull", "separator", "qp"), key=lambda n_: k6[n_])
    dom_name = {"hull": "hull_group_kernel", "separator": "separator_packed_kernel" if cull5 > 0.0 else "separator_kernel", "qp": kern5}[dom]
    ach_dom = bytes5 * S5 * N5 / (k6[dom] * 1e-3) / 1e9 if k6[dom] > 0 else 0.0
    traffic5 = acc.measured_traffic("nep::" + dom_name, "pmc_summary_config5_latest.txt")
    config5 = {"value": S5 * N5 * steps6 / dt6, "unit": "replans/s", "steps": steps6, "ms_per_step": dt6 / steps6 * 1e3,
               "step_ms": acc.step_quantiles(ms6),
               "workload": "%d agents + %d static obstacles, enable_entangle_check on (synthetic ent_state: one active case for 10 %% of the agent pairs, "
                           "2-4 bend points per agent, SURVEY 8d), K=8, %d seeded scenes in flight (seeds 0..%d)" % (N5, len(sc5[0]["statics"]), S5, S5 - 1),
               "replans_per_step": S5 * N5, "qp_kernel": kern5, "line_cull_radius_m": cull5,
               "kernel_ms": k6, "solve_us": us6, "presolve_redo_last_step": redo6,
               "lines_mean": float(sol6["stats"]["n_lines"].mean()), "rows_solved_mean": float(sol6["stats"]["n_rows"].mean()),
               "ipm_iters_mean": float(sol6["stats"]["iters"].mean()), "ipm_iters_max": int(sol6["stats"]["iters"].max()),
               "solved_without_iteration": int((sol6["stats"]["iters"] == 0).sum()), "lp_failed": int(sol6["stats"]["n_lp_failed"].sum()),
               "roofline": ({"bound": "hbm", "kernel": dom_name, "kernel_ms": k6[dom], "frac": None, "achieved": None, "peak": 8000.0, "unit": "GB/s",
                             "algorithmic_bytes_per_replan": bytes5, "replans_per_launch": S5 * N5,
                             "traffic": traffic5, "traffic_over_algorithmic": (traffic5 / (bytes5 * S5 * N5)) if traffic5 else None,
                             "traffic_source": "profiles/pmc_summary_config5_latest.txt (committed rocprofv3 --pmc summary of `bench.py --config5-only`)",
                             "note": "no fraction is printed for this leg: with the presolve the separator reads a 32-byte box instead of the hull of every obstacle "
                                     "it skips, so the kernel moves a small part of the bytes SURVEY 8d prices (traffic_over_algorithmic) and bytes / time would "
                                     "say nothing about the memory system — the kernels are VALU-issue bound.  The fractions of this size are `full_rows.roofline`"}
                            if cull5 > 0.0 else
                            {"bound": "hbm", "kernel": dom_name, "achieved": ach_dom, "peak": 8000.0, "unit": "GB/s", "frac": ach_dom / 8000.0,
                             "algorithmic_bytes_per_replan": bytes5, "replans_per_launch": S5 * N5, "kernel_ms": k6[dom]}),
               "scene_generation_wait_s": t_wait,
               "note": "the handle's default for this size: verified line presolve at %.1f m (lines farther from the guess are parked, checked at the "
                       "solution, re-solved with all rows on a violation), interior point on %s" % (cull5, kern5), **acc.status_counts(sol6)}
    if not args.config5_only:
        # every row through the interior point (presolve explicitly off): the LDS placement with its global spill
        b5.set_line_cull(0.0)
        kern5f = b5.qp_kernel_name()
        d_c5.copy_(b5.to_device(com5))
        dt7, steps7, ms7, k7, sol7 = c5_leg()
        config5["full_rows"] = {"value": S5 * N5 * steps7 / dt7, "unit": "replans/s", "steps": steps7, "ms_per_step": dt7 / steps7 * 1e3,
                                "qp_kernel": kern5f, "kernel_ms": k7, "solve_us": acc.solve_us_stats(b5),
                                "rows_solved_mean": float(sol7["stats"]["n_rows"].mean()), "ipm_iters_mean": float(sol7["stats"]["iters"].mean()),
                                "active_rows": acc.active_summary(b5),
                                "roofline": {"bound": "hbm", "peak": 8000.0, "unit": "GB/s", "algorithmic_bytes_per_replan": bytes5, "replans_per_launch": S5 * N5,
                                             **{kn: {"ms": k7[kk], "achieved": bytes5 * S5 * N5 / (k7[kk] * 1e-3) / 1e9 if k7[kk] > 0 else 0.0,
                                                     "frac": bytes5 * S5 * N5 / (k7[kk] * 1e-3) / 1e9 / 8000.0 if k7[kk] > 0 else 0.0}
                                                for kn, kk in (("separator_kernel", "separator"), (kern5f, "qp"))}},
                                "note": "nep_batch_set_line_cull(0): every separating-line row through the interior point", **acc.status_counts(sol7)}
    if not args.no_chain and not args.config5_only:
        # the whole chain at this size with the entangle check on: front end with per-node entangle states (guesses AND the
        # entangle cases are device-made) -> lines + QP -> safety check with the entangle re-check + commit
        b5.set_line_cull(cull5)
        cfg5 = scene.frontend_cfg(p5, beam_width=args.beam, entangle=True)
        for s_ in range(S5):
            reps_, long_ = scene.static_reps(sc5[s_]["statics"])
            b5.set_static_reps(reps_, long_, scene=s_)
        d_st5 = b5.to_device(np.stack([scene.frontend_starts(s_) for s_ in sc5]))
        d_gf5 = torch.zeros_like(d_g5); d_res5 = torch.zeros(S5 * N5 * abi.FE_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        d_case5 = torch.zeros(S5 * N5 * abi.NEP_MAX_POL * N5, dtype=torch.int32, device=dev)
        d_cc5 = b5.to_device(com5); d_nx5 = torch.empty_like(d_cc5); d_ac5 = torch.zeros(S5 * N5, dtype=torch.int32, device=dev)
        fe5, sf5 = [], []

        def c5_chain_step():
            e0 = ctx.ev()
            b5.frontend_ent(cfg5, d_cc5, d_st5, d_gf5, d_res5, d_case5)
            fe5.append((e0, ctx.ev()))
            b5.replan(None, d_gf5, d_ent=d_case5)
            e1 = ctx.ev()
            b5.safety_commit_ent(d_cc5, b5.d_commit, d_gf5, d_nx5, d_ac5)
            sf5.append((e1, ctx.ev()))
            d_cc5.copy_(d_nx5)
        steps8 = max(20, aux_steps // 10)
        dt8, ms8, _ = ctx.run_leg(c5_chain_step, [b5], steps8, 2, eager_after=5, clear=(fe5, sf5))
        k8 = {n_: b5.kernel_time_ms(i_)[0] for i_, n_ in ((1, "separator"), (2, "qp"))}
        b5.enable_timing(False)
        sol8 = b5.solutions(); res8 = d_res5.cpu().numpy().view(abi.FE_RESULT_DTYPE)
        config5["chain"] = {"value": S5 * N5 * steps8 / dt8, "unit": "replans/s", "steps": steps8, "ms_per_step": dt8 / steps8 * 1e3,
                            "kernel_ms": {"frontend_ent_with_hulls": ctx.mean_ms(fe5), "separator": k8["separator"], "qp": k8["qp"], "safety_ent": ctx.mean_ms(sf5)},
                            "beam_width": args.beam, "frontend_goal_reached": int((res8["status"] == 1).sum()), "frontend_no_solution": int((res8["status"] == 3).sum()),
                            "children_pruned_by_the_entangle_check": int(res8["n_entangled"].sum()), "ent_overflow": int(res8["ent_overflow"].sum()),
                            "big_records": {"searches": int(((res8["_pad"].astype(np.int64) >> 8) > 0).sum()), "children": int((res8["_pad"].astype(np.int64) >> 8).sum()),
                                            "note": "searches in which a child's entangle state outgrew the fixed record: re-run by the big-record instantiation after the "
                                                    "launch, bounded by the reference's own rule (num_agents + statics) only; ent_overflow counts the searches the POOL of big records failed (0)"},
                            "active_entangle_cases": int((d_case5 != 0).sum().item()),
                            "ipm_iters_mean": float(sol8["stats"]["iters"].mean()), "accepted_frac": float(d_ac5.float().mean().item()),
                            "solve_us": acc.solve_us_stats(b5),
                            "note": "frontend_kernel<true> (entangle states per search node) -> separator + QP on device-made guesses and device-made "
                                    "entangle cases -> safety check with entangleCheckGivenPwp + commit", **acc.status_counts(sol8)}
    b5.close()
    return config5


def _digest(sol, S, n_local):
    """[S][n_local] 64-bit digests of (status, K, coefficients) of every replan: what two runs of the same scenes must agree on"""
    sol = sol.reshape(S, n_local)
    out = np.zeros((S, n_local), dtype=np.uint64)
    for s in range(S):
        for a in range(n_local):
            r = sol[s, a]
            h = hashlib.sha1(np.ascontiguousarray(r["coeff"]).tobytes() + bytes([int(r["stats"]["status"]) & 255, int(r["K"]) & 255])).digest()
            out[s, a] = int.from_bytes(h[:8], "little")
    return out


def workload(ctx):
    """`--workload config5`: the headline step of BASELINE configs[4] on N GPUs (N = 1 included, through the same sharded
    machinery so that the per-N values compare like with like).  -> detail record on rank 0 (None elsewhere)"""
    torch, tdist, args, dev = ctx.torch, ctx.tdist, ctx.args, ctx.dev
    from neptune_amd import abi, dist as ndist
    from neptune_amd.backend import BatchBackend
    from .headline import native_nranks, rccl_record
    world, rank = ctx.world, ctx.rank
    N, M = args.agents, args.obstacles
    spg = args.scenes                                    # scenes in flight per GPU
    S = spg * world
    first_local, n_local = ndist.shard(N, world, rank)
    pool = ScenePool(N, M, range(rank * spg, (rank + 1) * spg), max(1, (ctx.host_cores // 2) // world),      # (the ranks share the host)
                     cache=os.environ.get("NEP_BENCH_SCENE_CACHE") if world == 1 else None)
    made = pool.wait()
    sc_l = [m[0] for m in made]
    p5 = dataclasses.replace(sc_l[0]["par"], enable_entangle=True)
    com_l, gue_l = ndist.stack_scenes(sc_l)
    case_l = np.stack([m[1] for m in made]).astype(np.int32)     # [spg][N][8][N]
    n_st = np.array([len(s["statics"]) for s in sc_l], dtype=np.int32)
    # statics as fixed-size arrays so that they can be shared like the records: [spg][M][16][2] + vertex counts
    st_xy = np.zeros((spg, M, abi.NEP_HULL_MAX_V, 2)); st_nv = np.zeros((spg, M), dtype=np.int32)
    for k, s in enumerate(sc_l):
        if len(s["statics"]) != M:
            raise SystemExit("scene %d drew %d static obstacles instead of %d" % (rank * spg + k, len(s["statics"]), M))
        for j, poly in enumerate(s["statics"]):
            st_nv[k, j] = len(poly); st_xy[k, j, :len(poly)] = poly
    com, gue = ctx.share(com_l, S), ctx.share(gue_l, S)
    st_xy, st_nv = ctx.share(st_xy, S), ctx.share(st_nv, S)
    # every rank needs the case blocks of ITS agents of every scene: [S][n_local][8][N]
    case = np.ascontiguousarray(ctx.share(case_l, S)[:, first_local:first_local + n_local]) if world > 1 else case_l
    statics_of = lambda s: [st_xy[s, j, :st_nv[s, j]].copy() for j in range(M)]
    del n_st

    # the all-gather through the C ABI's own RCCL binding inside the captured step (one rank: the same call, degenerate); gloo
    # (NEP_BENCH_ONE_DEVICE, a development aid) and --exchange-torch go through torch.distributed, launched from the host
    native = not args.exchange_torch and (world == 1 or ctx.dist_backend == "nccl")
    C = args.chunks if (world > 1 and S % max(args.chunks, 1) == 0 and S >= args.chunks) else 1      # (one rank: nothing to overlap a chunk's kernels with)
    Sc = S // C
    bes = []
    for k in range(C):
        b = BatchBackend(p5, statics_of(k * Sc), first_local=first_local, n_local=n_local, n_scenes=Sc, device=dev)
        for s_ in range(Sc):
            b.set_scene_statics(s_, statics_of(k * Sc + s_))
        if args.cull_radius is not None:
            b.set_line_cull(args.cull_radius)
        bes.append(b)
    be = bes[0]
    d_local = [bes[k].to_device(np.ascontiguousarray(com[k * Sc:(k + 1) * Sc, first_local:first_local + n_local])) for k in range(C)]
    d_guess = [bes[k].to_device(np.ascontiguousarray(gue[k * Sc:(k + 1) * Sc, first_local:first_local + n_local])) for k in range(C)]
    d_ent = [torch.from_numpy(np.ascontiguousarray(case[k * Sc:(k + 1) * Sc]).reshape(-1)).to(dev) for k in range(C)]
    hull_ev, gather_ev = [], []
    _ev_lists = {"hull": hull_ev, "wait": gather_ev, "frontend": []}
    def build(nat):
        for k in range(C):                                    # (fresh records: a failed first native step may have half-made a round)
            d_local[k].copy_(bes[k].to_device(np.ascontiguousarray(com[k * Sc:(k + 1) * Sc, first_local:first_local + n_local])))
        return ndist.ShardedRounds(bes, d_local, d_guess, world, rank, native=nat, d_ent=d_ent, carry=_carry_ranges(),
                                   timer=lambda name: ctx.timed(_ev_lists[name]))
    rounds, native, steps_made, exchange_note = ctx.rounds_with_fallback(build, native)
    nranks = native_nranks(ctx, rounds) if native else ([tdist.get_world_size()] * world if ctx.use_dist and ctx.dist_backend == "nccl" else None)
    step = rounds.step
    for _ in range(max(args.warmup - steps_made, 0)):
        step()
    # what the scenes' replans are after the warm-up rounds: two runs (another N, another chunking) must agree byte for byte
    torch.cuda.synchronize(dev)
    dig_l = np.concatenate([_digest(b.solutions(), Sc, n_local) for b in bes])          # [S][n_local]
    if world > 1:
        pieces = [None] * world
        tdist.all_gather_object(pieces, dig_l)
        dig = np.concatenate(pieces, axis=1)                                               # rank order = agent-id order
    else:
        dig = dig_l
    scene_digest = [hashlib.sha1(np.ascontiguousarray(dig[s]).tobytes()).hexdigest()[:16] for s in range(min(S, 8))]
    dt, step_ms, graph = ctx.run_leg(step, bes, args.steps, 0, graph_ok=native, clear=(hull_ev, gather_ev))
    dt_local = ctx.last_wall
    k_ms = {n_: be.kernel_time_ms(i_)[0] for i_, n_ in ((1, "separator"), (2, "qp"), (3, "sequence"))}
    k_ms["hull"] = ctx.mean_ms(hull_ev); k_ms["exchange_wait"] = ctx.mean_ms(gather_ev)
    for b in bes:
        b.enable_timing(False)
    sol = np.concatenate([b.solutions() for b in bes])
    solve_us = acc.solve_us_stats(be)
    value = S * N * args.steps / dt
    mine_rec = {"rank": rank, "device": torch.cuda.get_device_name(dev), "kernel_ms": k_ms,
                "step_ms_p50": float(np.percentile(step_ms, 50)), "step_ms_max": float(step_ms.max()), "wall_s": dt_local}
    per_rank = None
    if world > 1:
        per_rank = [None] * world
        tdist.all_gather_object(per_rank, mine_rec)
    if rank != 0:
        return None
    from neptune_amd.backend import hulls_batch
    _, hn, _, _ = hulls_batch(com[0], float(gue[0, 0]["t_start"]), p5.num_pol, p5.T_span, p5.drone_radius)
    bend0 = com[0]["n_bend"].astype(np.float64)
    bytes5 = acc.algorithmic_bytes(p5, {"guesses": gue[0], "statics": statics_of(0)}, hn, int(sol[0]["n_states"]), ent_bytes=_ent_bytes(N, bend0))
    launch_replans = Sc * n_local
    cull = be.line_cull(); kern = be.qp_kernel_name()
    dom = max(("separator", "qp"), key=lambda n_: k_ms[n_])
    dom_name = {"separator": "separator_packed_kernel" if cull > 0.0 else "separator_kernel", "qp": kern}[dom]
    ach = bytes5 * launch_replans / (k_ms[dom] * 1e-3) / 1e9 if k_ms[dom] > 0 else 0.0
    traffic = acc.measured_traffic("nep::" + dom_name, "pmc_summary_config5_latest.txt") if launch_replans == 8192 else None
    status = sol["stats"]["status"].astype(int); iters = sol["stats"]["iters"].astype(int)
    out = {
        "metric": "backend_replans_per_sec", "value": value, "unit": "replans/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "config5: %d agents + %d static obstacles, enable_entangle_check on (synthetic ent_state, SURVEY 8d), K=8, %d seeded scenes in "
                               "flight per GPU per step (seeds 0..%d over %d GPU%s)" % (N, M, spg, S - 1, world, "" if world == 1 else "s"),
                   "agents": N, "obstacles": M, "scenes_in_flight": S, "scenes_per_gpu": spg,
                   "replans_per_step": S * N, "replans_per_gpu_per_step": S * n_local,
                   "sharding": ("agents of every scene block-sharded by id, %d per GPU; per step and scene chunk (%d chunks, pipelined) one all-gather (%s) of the "
                                "local agents' hull blocks (interval hulls, tether samples, bend points: %d B per agent and scene); the entangle case blocks stay local"
                                % (n_local, C, "RCCL, native binding on a side stream inside the captured step" if native else ("torch.distributed " + ctx.dist_backend),
                                   be.hull_block_bytes() // (Sc * n_local))),
                   "params": "reference neptune_mtlp_benchmark.yaml (T_span 0.5, num_pol 8, weight 1000, v 2, a 3), world scaled to the density of 5 agents in 24 m x 24 m"},
        "solver": {"status_ok": int((status == 0).sum()), "status_relaxed": int((status == 1).sum()), "status_failed": int((status == 2).sum()),
                   "ipm_iters_mean": float(iters.mean()), "lines_mean": float(sol["stats"]["n_lines"].mean()),
                   "rows_solved_mean": float(sol["stats"]["n_rows"].mean()), "lp_failed": int(sol["stats"]["n_lp_failed"].sum()),
                   "solved_without_iteration": int((iters == 0).sum()), "line_cull_radius": cull, "qp_kernel": kern,
                   "presolve_redo_last_step": be.redo_count(), "active_rows": acc.active_summary(be)},
        # (a batched replan completes with its launch sequence: SURVEY 8d's replan = setters + separator loop + QP + generatePwpOut)
        "qp_workgroup_us": solve_us, "p50_solve_ms": k_ms["sequence"], "p99_solve_ms": float(np.percentile(step_ms, 99)),
        "solve_ms_definition": "one batched replan completes with its launch sequence (hulls + separating lines + QP + polish of %d replans): p50 = its HIP-event duration, p99 = the step's p99" % launch_replans,
        "step_ms": acc.step_quantiles(step_ms),
        "kernel_ms": dict(k_ms, launches_per_step=C),
        "launch": ("one captured HIP graph per step, replayed (per-kernel events from eager steps after the timed region)" if graph is not None
                   else ("; ".join(ctx.graph_notes) or "host launches")),
        "roofline": {"bound": "hbm", "kernel": dom_name, "kernel_ms": k_ms[dom], "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                     "algorithmic_bytes_per_replan": bytes5, "replans_per_launch": launch_replans, "traffic": traffic,
                     "traffic_over_algorithmic": (traffic / (bytes5 * launch_replans)) if traffic else None,
                     "note": ("with the presolve the separator reads a 32-byte box instead of the hull of every obstacle it skips, so it moves a small part of the bytes "
                              "SURVEY 8d prices (traffic_over_algorithmic): frac is the contract's quotient, not a statement about the memory system" if cull > 0.0 else "")},
        "per_gpu_value": value / world, "per_rank": per_rank,
        "rccl": dict(rccl_record(ctx, nranks, native), **({"note": exchange_note} if exchange_note else {})),
        "scene_digest": scene_digest, "scene_digest_note": "sha1 over (status, K, coefficients) of the 256 replans of each of the first scenes after the warm-up rounds: equal "
                                                           "for any N and any chunking (tests/test_gpu_bench_launch.py)",
        "scene_generation_wait_s": pool.wait_s,
    }
    if ctx.graph_notes:
        out["graph_notes"] = ctx.graph_notes
    if not args.no_cpu_baseline and world == 1:
        from .cpu import cpu_baseline
        out["cpu_baseline"] = cpu_baseline(p5, [{"committed": com[s], "guesses": gue[s], "statics": statics_of(s)} for s in range(min(S, 8))],
                                           ent=[case_l[s] for s in range(min(S, 8))])
        out["reference_solvers"] = out["cpu_baseline"].pop("reference_solvers")
    rounds.native and rounds.native.close()
    for b in bes:
        b.close()
    return out
