#!/usr/bin/env python3
"""Headline benchmark of the back-end path: whole-node back-end replans per second.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
          --master-port P bench.py --gpus N --steps K --warmup W; a plain `python bench.py --gpus N` launches those ranks itself)

Workload (BASELINE.json north_star / configs[3] scene on the named GPU count): 64 agents + 20 static polytope obstacles,
K = 8 segments, reference yaml parameters, 128 seeded scenes in flight per GPU per step (--scenes).  One
step = one bulk-synchronous round: every agent of every scene does one full back-end replan (MINVO hulls of the other
agents' committed trajectories -> separating-line LPs -> spline QP -> sampled states -> committed record; reference
neptune.cpp:1512-1529, timed call solver_gurobi_poly.cpp:823-826); the new trajectories are the obstacles of the next step.
Inputs are resident in HBM before the timed region.  `--workload config5`: BASELINE configs[4] instead (256 agents + 100
obstacles, enable_entangle_check on), 256 / N agents per GPU.

Output: ONE short JSON line on stdout (bench_legs/compact.py: the contract's fields, `roofline`, `cpu_baseline`, a few
scalar highlights; <= 4 KB — the driver parses the last line of an 8 KB tail).  The full record of every leg — long_run,
launch_order_off, reference_tolerances, presolve, chain, moving, crossing, single_scene, small_configs, per_agent_api,
config5 (+ full_rows, chain), per-kernel rooflines, notes — goes to bench_detail.json beside this file (--detail) and, as one
line, to stderr.  The legs live in bench_legs/ (see its __init__).

N > 1: the agents of every scene are block-sharded by id across the ranks and the number of scenes grows with N, so every GPU
does the same number of replans per step at any N: "scaling" is "weak".  The exchange step is one RCCL all-gather per round
and scene chunk of what the other agents' replans consume of a committed trajectory — its interval hulls — issued through
the C ABI's own RCCL binding on a side stream inside the step; the whole per-rank step is captured into one HIP graph.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=["config4", "config5"], default="config4",
                    help="config4: 64 agents + 20 obstacles (the headline the metric is quoted on); config5: BASELINE configs[4], 256 agents + 100 "
                         "obstacles with the entangle rows, 256 / N agents per GPU (no extra legs)")
    ap.add_argument("--agents", type=int, default=None, help="agents per scene (default 64; 256 with --workload config5)")
    ap.add_argument("--obstacles", type=int, default=None, help="static obstacles per scene (default 20; 100 with --workload config5)")
    ap.add_argument("--scenes", type=int, default=None, help="seeded scenes in flight PER GPU (default 128; 32 with --workload config5: 8 192 replans per GPU at N = 1)")
    ap.add_argument("--aux-steps", type=int, default=200, help="timed steps of the separately reported legs (at least --steps)")
    ap.add_argument("--exchange", choices=["hulls", "records"], default="hulls",
                    help="N > 1: all-gather the interval hulls of the local agents' committed trajectories (hull work "
                         "sharded with the agents) or the trajectory records themselves (every rank rebuilds all hulls)")
    ap.add_argument("--groups", type=int, default=1,
                    help="one GPU: run the scenes in flight as this many groups, each a launch sequence on its own HIP stream inside the one captured "
                         "graph per step (the scenes are independent fleets).  Default 1 = one launch sequence on one stream; measured in round 6 "
                         "(same box, 50 steps): 1 group 16.65 M replans/s, 2 groups 17.00 M, 4 groups 11.5 M, 8 groups 8.5 M — every group's "
                         "interior-point launch keeps its own tail of long solves, so the overlap buys 2 % at best; with G > 1 the one-sequence "
                         "step is reported as `one_stream`")
    ap.add_argument("--pipeline", action="store_true",
                    help="with --groups G: the groups' rounds PIPELINED on streams instead of forked and joined inside one graph per step — the geometry halves "
                         "(nep_batch_replan_lines) of all groups in turn on one stream, each group's QP half (nep_batch_replan_solve) on the group's own stream, "
                         "ordered by events only; host-launched (DESIGN section 16)")
    ap.add_argument("--chunks", type=int, default=2,
                    help="N > 1 with --exchange hulls: scene chunks pipelined so that one chunk's all-gather overlaps the other's kernels")
    ap.add_argument("--presolve-radius", type=float, default=4.0, help="(older command lines; the presolve is the handle's default since round 6: ignored)")
    ap.add_argument("--no-full-rows", action="store_true", help="skip the separately reported full_rows legs (presolve off: every separating-line row through the interior point)")
    ap.add_argument("--cull-radius", type=float, default=None,
                    help="verified presolve of the separating-line rows (nep_batch_set_line_cull): lines farther than this many metres "
                         "from the guess are left out of the QP and verified after the solve.  Default: the handle's own default (4 m at every scene "
                         "size); 0 = off, every row through the interior point (rounds 1-5's headline)")
    ap.add_argument("--chain-cull-radius", type=float, default=None,
                    help="the same for the chain / moving / crossing legs (default: the handle's default; 0: every row)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange-torch", action="store_true",
                    help="N > 1 with --exchange hulls: the all-gather through torch.distributed (host-launched steps) instead of the "
                         "C ABI's own RCCL binding inside one captured HIP graph per step")
    ap.add_argument("--exchange-native", action="store_true", help="(the default now; kept for older command lines)")
    ap.add_argument("--no-graph", action="store_true", help="launch every step from the host instead of replaying one captured HIP graph")
    ap.add_argument("--no-process-group", action="store_true", help="single GPU: do not create the one-rank RCCL process group")
    ap.add_argument("--no-chain", action="store_true", help="skip the separately reported front end + safety legs (chain, moving)")
    ap.add_argument("--no-config5", action="store_true", help="skip the separately reported BASELINE configs[4] leg")
    ap.add_argument("--config5-scenes", type=int, default=32, help="scenes in flight of the config-5 leg (x 256 agents = replans per step)")
    ap.add_argument("--config5-only", action="store_true", help="development aid: only the config-5 leg (profiling)")
    ap.add_argument("--no-extra-legs", action="store_true", help="only the headline (and what --frontend / --safety ask for)")
    ap.add_argument("--frontend", action="store_true",
                    help="also run the front-end beam search (SURVEY §8f rank 2) in every step: the guesses are made on the device "
                         "from point A and the goal instead of being read from the scene (single GPU)")
    ap.add_argument("--beam", type=int, default=32)
    ap.add_argument("--safety", action="store_true",
                    help="also run the post-solve safety check + commit (SURVEY §8f rank 1) in every step")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"), help="where every leg's full record goes ('' = nowhere)")
    args = ap.parse_args(argv)
    c5 = args.workload == "config5"
    if args.agents is None:
        args.agents = 256 if c5 else 64
    if args.obstacles is None:
        args.obstacles = 100 if c5 else 20
    if args.scenes is None:
        args.scenes = 32 if c5 else 128
    return args


def launch_ranks(args):
    """plain `python bench.py --gpus N` (the shape of the driver's single-GPU command): launch the N ranks ourselves, one
    process per GPU under torch.distributed.run, rendezvous on 127.0.0.1; rank 0's JSON line is the only thing on stdout"""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if env.get("NEP_BENCH_ONE_DEVICE") == "1":
        env.setdefault("NEP_BENCH_BACKEND", "gloo")        # several ranks on one GPU: RCCL refuses duplicate devices
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def run_config4(ctx):
    """the default command: headline + the separately reported legs -> detail record (rank 0) or None"""
    from bench_legs import chain as chain_legs, config5, cpu, headline, small
    args, world, rank = ctx.args, ctx.world, ctx.rank
    extra = world == 1 and not args.no_extra_legs and not args.frontend and not args.safety and args.cull_radius is None
    want_c5 = (extra and not args.no_config5) or args.config5_only
    pool = None
    if want_c5 and rank == 0:
        pool = config5.ScenePool(256, 100, range(args.config5_scenes), ctx.host_cores // 2, cache=os.environ.get("NEP_BENCH_SCENE_CACHE"))
    ctx.init_process_group(want_one_rank_group=not args.no_process_group and not args.config5_only)
    if args.config5_only:
        c5 = config5.leg(ctx, pool.wait(), pool.wait_s) if rank == 0 else None
        return {"metric": "backend_replans_per_sec", "config5": c5, "graph_notes": ctx.graph_notes, **{k: c5[k] for k in ("value", "unit")}} if c5 else None
    H = headline.setup(ctx, pool)
    headline.run(ctx, H)
    legs = headline.retimed_legs(ctx, H)
    Hc = headline.full_handle_view(ctx, H) if (extra and (H.C == 1 or H.G > 1)) else None
    if Hc is not None and H.G > 1:
        legs["one_stream"] = headline.one_stream_leg(ctx, H, Hc)
        if getattr(H, "PIPE", False):      # (per-kernel HIP events belong to ONE launch sequence: the pipelined groups overlap, so the line's kernel_ms are the one-stream leg's)
            k1 = legs["one_stream"]["kernel_ms"]
            H.hull_ms, H.sep_ms, H.qp_ms, H.seq_ms = k1["hull"], k1["separator"], k1["qp"], k1["sequence"]
    if Hc is not None and not args.no_chain:
        legs["chain"], legs["moving"], legs["crossing"] = chain_legs.run(ctx, Hc)
    if extra:
        legs["single_scene"] = small.single_scene(ctx, H)
        if rank == 0:
            legs["small_configs"] = small.small_configs(ctx, H)
            legs["per_agent_api"] = small.per_agent_api(ctx, H)
    if want_c5 and rank == 0:
        legs["config5"] = config5.leg(ctx, pool.wait(), pool.wait_s)
    per_rank = headline.per_rank_records(ctx, H)
    if rank != 0:
        return None
    out = headline.record(ctx, H)
    mv, cr = legs.get("moving"), legs.get("crossing")
    if H.G > 1 and getattr(H, "PIPE", False):
        how = ("%d scene groups PIPELINED on streams (the geometry halves in turn on one stream, each group's QP half on its own, ordered by events only; every "
               "step is one round of every group, all of them inside the timed region; kernel_ms and one_stream: the same step as one launch sequence); " % H.G)
    elif H.G > 1:
        how = "%d scene groups on %d HIP streams inside one captured graph per step (one_stream: the same step as one launch sequence); " % (H.G, H.G)
    else:
        how = ""
    out["what_value_is"] = (how +
                            "throughput of %d INDEPENDENT scenes in flight on the handle's default solve path (verified line presolve + polish; full_rows has every row "
                            "through the interior point), QP workgroups ordered by the previous step's measured times (exact here: the same problems every step).  The representative figures are the closed-loop legs, "
                            "where every step poses new problems from device-made guesses: moving %s replans/s, crossing (the whole fleet through the middle) %s "
                            "replans/s — see also launch_order_off, single_scene, active_rows"
                            % (H.S, ("%.3g" % mv["value"]) if mv else "n/a", ("%.3g" % cr["value"]) if cr else "n/a"))
    out.update(legs)
    pa4 = next((v for k, v in (legs.get("per_agent_api") or {}).items() if k.startswith("config4") and isinstance(v, dict)), None)
    if pa4:
        # the metric's "p50 solve ms": one replan = setters + separator loop + QP + generatePwpOut — measured where that is one
        # blocking call sequence, the drop-in's (neptune.cpp:1514-1527), at this workload's size; the batch view stays in batch_sequence_ms
        out["p50_solve_ms"] = pa4["sequence_ms"]["p50"]; out["p99_solve_ms"] = pa4["sequence_ms"]["p99"]
        out["solve_ms_definition"] = ("the six-call drop-in sequence of ONE replan through the per-agent C ABI (setInitTrajectory .. generatePwpOut, host buffers in and "
                                      "out, blocking) against %d hull lists: per_agent_api; a batched replan completes with its launch sequence, batch_sequence_ms" % pa4["hull_lists"])
    out["per_rank"] = per_rank
    if ctx.graph_notes:
        out["graph_notes"] = ctx.graph_notes
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu.cpu_baseline(H.p, H.mine)
        out["reference_solvers"] = out["cpu_baseline"].pop("reference_solvers")
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    from bench_legs import compact, config5
    from bench_legs.context import Ctx
    ctx = Ctx(args)
    if args.workload == "config5":
        ctx.init_process_group(want_one_rank_group=not args.no_process_group)
        detail = config5.workload(ctx)
    else:
        detail = run_config4(ctx)
    ctx.teardown()
    if ctx.rank == 0 and detail is not None:
        where = None
        if args.detail:
            try:
                with open(args.detail, "w") as f:
                    json.dump(detail, f, indent=1)
                where = os.path.relpath(args.detail, ROOT) if args.detail.startswith(ROOT) else args.detail
            except OSError as e:
                print("[bench] detail file not written: %r" % (e,), file=sys.stderr)
        print("[bench detail] " + json.dumps(detail), file=sys.stderr)
        sys.stderr.flush()
        line = json.dumps(compact.compact_line(detail, where))
        sys.stdout.write(line + "\n")                       # the LAST thing this process prints
        sys.stdout.flush()


if __name__ == "__main__":
    main()
