#!/usr/bin/env python3
"""Headline benchmark of the back-end path: whole-node back-end replans per second.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
          --master-port P bench.py --gpus N --steps K --warmup W)

Workload (BASELINE.json north_star / configs[3] scene on the named GPU count): 64 agents + 20
static polytope obstacles, K = 8 segments, reference yaml parameters, 128 seeded scenes (SURVEY.md
§8d) in flight per GPU per step (--scenes).  One step = one bulk-synchronous round: every agent of every scene
does one full back-end replan (MINVO hulls of the other agents' committed trajectories ->
separating-line LPs -> spline QP -> sampled states -> committed record); the new trajectories are
the obstacles of the next step.  Inputs are resident in HBM before the timed region.  Every scene carries its
own static obstacles (nep_batch_set_scene_statics); the CPU baseline solves the same scenes.

`value` is that leg, timed over exactly --steps steps.  Because a number means little without what it depends on, the
same JSON line carries further legs, each timed the same way (barrier + synchronize on both sides; at least 200 steps):
  long_run          the headline's step over >= 200 steps (the driver's 20 steps are 33 ms)
  launch_order_off  the same with the QP workgroups in slot order (the headline orders them by the previous replan's
                    measured time, and re-solves the same problems every step: its predictor is exact)
  presolve          the verified row presolve on
  chain             front-end beam search -> lines -> QP -> safety check + commit: device-made guesses
  moving            the closed loop on the device: chain + point A of the next round from the committed trajectories
                    (nep_batch_next_starts), goals swapped on arrival — the problems change every step and the
                    launch-order predictor is the previous replan of the same agent
  single_scene      ONE fleet of 64 agents (the latency of a round, and the throughput of a single fleet)
  config5           BASELINE configs[4]: 256 agents + 100 obstacles, enable_entangle_check on
and `solve_us`: the per-replan device time distribution (p50 / p99) the metric asks for.

N > 1: the agents of every scene are block-sharded by id across the ranks (64/N per GPU) and the
number of scenes grows with N (128 per GPU), so every GPU does 8192 replans per step at any N:
"scaling" is "weak".  The exchange step is one RCCL all-gather per round and scene chunk of what the other agents'
replans consume of a committed trajectory — its interval hulls — issued through the C ABI's own RCCL binding on a
side stream inside the step, and the whole per-rank step (hulls, all-gather, separator, order, QP, every chunk) is
captured into one HIP graph and replayed (--exchange-torch: torch.distributed's collective, launched from the host;
--exchange records all-gathers the trajectory records instead and rebuilds every hull on every rank).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def algorithmic_bytes(p, sc, hull_nv, n_states, ent_bytes=0.0):
    """fp64 compulsory traffic of one replan: guess + other agents' hull vertices
    + statics + bases (+ entangle inputs) in; coefficients, cost, status and sampled states out."""
    K = int(sc["guesses"][0]["K"])
    guess = 8 * (12 * K + (K + 1))
    N = p.num_agents
    hull = 16.0 * hull_nv[:, :K].sum() * (N - 1) / N          # per agent: every other agent's hulls
    statics = 16 * sum(len(s) for s in sc["statics"])
    bases = 16 * N
    out = 8 * (12 * K + 1) + 4 + 96 * n_states
    return guess + hull + statics + bases + ent_bytes + out


def algorithmic_flops(K, lines_mean, vertices_mean, iters_mean):
    """fp64 operations of one replan by SURVEY.md §8d's count: separator L (V+4) 3 2 I_lp with I_lp = 10; QP per
    interior-point iteration m n^2 + n^3/3 + 4 m n for the (x, y) system (n = 2K, m = 32K + 4L) and the z system
    (n = K, m = 16K)."""
    L = lines_mean
    sep = L * (vertices_mean + 4) * 3 * 2 * 10
    n_xy, m_xy, n_z, m_z = 2 * K, 32 * K + 4 * L, K, 16 * K
    per_iter = (m_xy * n_xy ** 2 + n_xy ** 3 / 3 + 4 * m_xy * n_xy) + (m_z * n_z ** 2 + n_z ** 3 / 3 + 4 * m_z * n_z)
    return sep + iters_mean * per_iter


def measured_traffic(kernel, name="pmc_summary_latest.txt"):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc summary of this same command (profiles/):
    FETCH_SIZE and WRITE_SIZE are reported in KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (its
    calibration is for 16 B/lane streams; our 8 B/lane reads are uncalibrated, so this is an upper bound).  A replay of a
    committed file, not a measurement of this run (counters cannot be read from inside the process); None when no
    summary is committed."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    cur, vals = None, {}
    for line in open(path):
        if line.startswith("nep::"):
            cur = line.strip()
        elif cur is not None and cur.split("<")[0] == kernel and "mean" in line:
            parts = line.split()
            vals[parts[0]] = float(parts[2])
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    return None


def cpu_baseline(p, scenes, budget_s=12.0):
    """The CPU oracle (kind "port": the reference needs Gurobi/GLPK/CGAL, absent here) timed on the
    host cores on a bounded sample of the same workload."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    oracle.lib()
    cores = os.cpu_count() or 1
    jobs = [(s, a) for s in scenes for a in range(p.num_agents)] * 64   # bounded by time below
    t0 = time.perf_counter()
    done = 0

    def one(job):
        s, a = job
        return oracle.replan(p, a + 1, s["committed"], s["guesses"][a], s["statics"])["status"]
    chunk = max(cores * 4, 64)
    with ThreadPoolExecutor(cores) as ex:     # ctypes releases the GIL: one solver thread per core
        for k in range(0, len(jobs), chunk):
            list(ex.map(one, jobs[k:k + chunk]))
            done += len(jobs[k:k + chunk])
            if time.perf_counter() - t0 > budget_s:
                break
    dt = time.perf_counter() - t0
    out = {"value": done / dt, "unit": "replans/s", "cores": cores, "kind": "port",
           "sample": "%d replans of the same scenes (seeds 0..), one oracle thread per core, %.1f s" % (done, dt)}
    # the reference's own solvers, where a box has them (neither is in this image: then the line says so)
    from oracle import reference_solvers as rs
    out["reference_solvers"] = rs.probe()
    if rs.glpk_lib() is not None:
        rng = np.random.default_rng(0)
        A = rng.uniform(-1, 1, (200, 8, 2)); B = rng.uniform(-1, 1, (200, 4, 2)) + np.array([3.0, 0.0])
        t1 = time.perf_counter()
        for a_, b_ in zip(A, B):
            rs.glpk_separator(a_, b_)
        out["reference_solvers"]["glpk_us_per_lp"] = (time.perf_counter() - t1) / 200 * 1e6
    return out


def _config5_scene(job):
    """pool worker: one BASELINE configs[4] scene with its synthetic entangle inputs"""
    n_agents, n_static, seed = job
    from neptune_amd import scene
    sc = scene.make_scene(n_agents, n_static, seed=seed)
    case_id = scene.synthetic_entangle(sc, seed=1000 + seed, frac=0.1)
    return sc, case_id


def quantiles(a):
    a = np.asarray(a, dtype=np.float64)
    return {"p50": float(np.percentile(a, 50)), "p90": float(np.percentile(a, 90)), "p99": float(np.percentile(a, 99)), "max": float(a.max()), "mean": float(a.mean())}


def active_summary(be_, tol=1e-6):
    """which replans are constrained at all: inequality rows with slack < tol at the optimum, over EVERY replan of the leg's last step
    (nep_batch_active_rows: box rows = position / velocity / acceleration bounds, line rows = separating lines)"""
    ar = be_.active_rows(tol)
    nb, nl = ar[:, 0], ar[:, 1]
    return {"sample": "every replan of the last step (%d)" % len(ar), "replans_with_active_rows_frac": float(((nb + nl) > 0).mean()),
            "replans_with_active_line_rows_frac": float((nl > 0).mean()), "replans_with_active_box_rows_frac": float((nb > 0).mean()),
            "active_box_rows_mean": float(nb.mean()), "active_line_rows_mean": float(nl.mean()), "tol_m": tol}


def status_counts(sol):
    st = sol["stats"]["status"].astype(int)
    return {"status_ok": int((st == 0).sum()), "status_relaxed": int((st == 1).sum()), "status_failed": int((st == 2).sum())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--agents", type=int, default=64)
    ap.add_argument("--obstacles", type=int, default=20)
    ap.add_argument("--scenes", type=int, default=128, help="seeded scenes in flight PER GPU")
    ap.add_argument("--aux-steps", type=int, default=200, help="timed steps of the separately reported legs (at least --steps)")
    ap.add_argument("--exchange", choices=["hulls", "records"], default="hulls",
                    help="N > 1: all-gather the interval hulls of the local agents' committed trajectories (hull work "
                         "sharded with the agents) or the trajectory records themselves (every rank rebuilds all hulls)")
    ap.add_argument("--chunks", type=int, default=2,
                    help="N > 1 with --exchange hulls: scene chunks pipelined so that one chunk's all-gather overlaps the other's kernels")
    ap.add_argument("--presolve-radius", type=float, default=4.0,
                    help="radius of the separately reported presolve leg (0: skip it); ignored when --cull-radius is set")
    ap.add_argument("--cull-radius", type=float, default=0.0,
                    help="presolve of the separating-line rows (nep_batch_set_line_cull): lines farther than this many metres "
                         "from the guess are left out of the QP and verified after the solve; 0 = off")
    ap.add_argument("--chain-cull-radius", type=float, default=0.0,
                    help="line presolve radius of the chain and moving legs (0: every row through the interior point)")
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
    args = ap.parse_args()
    aux_steps = max(args.aux_steps, args.steps)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N` (the shape of the driver's single-GPU command): launch the N ranks ourselves, one
        # process per GPU under torch.distributed.run, rendezvous on 127.0.0.1; rank 0's JSON line is the only thing on stdout
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
        raise SystemExit(subprocess.call(cmd, env=env))

    import torch
    from neptune_amd import abi, dist as ndist, scene
    from neptune_amd.backend import BatchBackend

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the back end has no CPU path")
    if world > 1 and os.environ.get("NEP_BENCH_ONE_DEVICE") != "1" and torch.cuda.device_count() < world:
        raise SystemExit("--gpus %d: this box shows %d GPU(s) (development aid: NEP_BENCH_ONE_DEVICE=1 runs the ranks on one device over gloo)"
                         % (world, torch.cuda.device_count()))
    # development aid: several ranks on ONE GPU over gloo (the driver's runs use one GPU per rank over RCCL)
    one_device = os.environ.get("NEP_BENCH_ONE_DEVICE") == "1"
    dist_backend = os.environ.get("NEP_BENCH_BACKEND", "nccl")
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as tdist
    use_dist = world > 1 or "RANK" in os.environ          # launched by torch.distributed.run
    own_group = False; rccl_torn_down = False              # (plain `python bench.py`: the one-rank group made below)
    rccl_note = None
    extra = world == 1 and not args.no_extra_legs and not args.frontend and not args.safety and args.cull_radius == 0.0

    # the config-5 scenes take ~20 s of host time each: a pool makes them while the GPU runs the other legs
    c5_pool, c5_futs, c5_made, c5_wait = None, None, None, 0.0
    want_c5 = (extra and not args.no_config5) or args.config5_only
    c5_cache = os.environ.get("NEP_BENCH_SCENE_CACHE")        # development aid (profiling scripts call this file several times on one box)
    if want_c5 and rank == 0 and c5_cache and os.path.exists(c5_cache):
        import pickle
        c5_made = pickle.load(open(c5_cache, "rb"))
        if len(c5_made) != args.config5_scenes:
            c5_made = None
    if want_c5 and rank == 0 and c5_made is None:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        c5_pool = ProcessPoolExecutor(max_workers=min(args.config5_scenes, max(1, (os.cpu_count() or 1) // 2)), mp_context=mp.get_context("spawn"))
        c5_futs = [c5_pool.submit(_config5_scene, (256, 100, s)) for s in range(args.config5_scenes)]

    class _stdout_to_stderr:
        """RCCL prints a version banner on stdout when its first communicator comes up; stdout carries the one JSON line"""
        def __enter__(self):
            sys.stdout.flush()
            self.saved = os.dup(1); os.dup2(2, 1)
        def __exit__(self, *a):
            import ctypes
            try:
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            os.dup2(self.saved, 1); os.close(self.saved)

    def init_group(backend, **kw):
        with _stdout_to_stderr():
            if backend == "nccl":
                tdist.init_process_group("nccl", device_id=dev, **kw)
                t = torch.ones(1, device=dev)
                tdist.all_reduce(t)                              # brings the communicator up now (and its banner with it)
                torch.cuda.synchronize(dev)
            else:
                tdist.init_process_group(backend, **kw)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        init_group(dist_backend)
    elif not args.no_process_group and not args.config5_only:
        # plain `python bench.py`: a one-rank RCCL process group, so that the single-GPU record also shows the collective
        # library initialising on the box and the round's all-gather call path running (degenerate: one rank)
        try:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
            init_group("nccl", rank=0, world_size=1)
            use_dist = True; own_group = True
        except Exception as e:                                 # never lose the measurement to the extra
            rccl_note = "one-rank process group not created: %r" % (e,)

    def barrier():
        torch.cuda.synchronize(dev)
        if use_dist:
            tdist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(dt):
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if dist_backend == "nccl" else "cpu")
            tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
            return float(t.item())
        return dt

    ev_on = [True]

    def ev():
        if not ev_on[0]:                 # (while a step is being captured into a graph: timing events cannot live inside one)
            return None
        e = torch.cuda.Event(enable_timing=True); e.record(); return e

    class _timed:
        def __init__(self, lst): self.lst = lst
        def __enter__(self): self.e0 = ev()
        def __exit__(self, *a): self.lst.append((self.e0, ev()))

    def mean_ms(pairs):
        pairs = [(a, b) for a, b in pairs if a is not None and b is not None]     # (pairs "recorded" while a graph was being captured are placeholders)
        return float(np.mean([a.elapsed_time(b) for a, b in pairs])) if pairs else 0.0

    graph_notes = []

    def capture(fn, handles, allow=True):
        """fn() enqueues one step on the current stream -> a captured graph of it, or None (then the host launches).
        One step = a fixed sequence of launches on fixed buffers (at N > 1 including the RCCL all-gathers on their side
        stream): captured once, replayed in the timed region — no per-launch host work, no host jitter between kernels."""
        if args.no_graph or not allow:
            return None
        try:
            for b in handles:
                b.enable_timing(False)
            torch.cuda.synchronize(dev)
            g_ = torch.cuda.CUDAGraph()
            ev_on[0] = False
            with torch.cuda.graph(g_):
                fn()
            ev_on[0] = True
            g_.replay(); g_.replay()
            torch.cuda.synchronize(dev)
            return g_
        except Exception as e:                       # (falls back to launching from the host)
            ev_on[0] = True
            graph_notes.append("graph capture failed: %r" % (e,))
            torch.cuda.synchronize(dev)
            return None

    last_wall = [0.0]

    def run_leg(step_fn, handles, steps, warm, graph_ok=True, eager_after=40, clear=()):
        """warm untimed steps, then exactly `steps` steps between barriers (max over ranks), replaying one captured graph
        when possible.  Per-kernel HIP events (handle timing) cannot live inside a graph: with a graph they are taken from
        `eager_after` host-launched steps after the timed region.  -> (seconds, per-step GPU ms, graph used)"""
        for _ in range(warm):
            step_fn()
        barrier()
        g_ = capture(step_fn, handles, graph_ok)
        for b in handles:
            b.enable_timing(g_ is None); b.reset_timing()
        for lst in clear:
            lst.clear()
        barrier()
        evs = [ev()]
        t0 = time.perf_counter()
        for _ in range(steps):
            if g_ is not None:
                g_.replay()
            else:
                step_fn()
            evs.append(ev())
        barrier()
        last_wall[0] = time.perf_counter() - t0
        dt = max_over_ranks(last_wall[0])
        step_ms = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(steps)])
        if g_ is not None and eager_after > 0:
            for b in handles:
                b.enable_timing(True); b.reset_timing()
            for lst in clear:
                lst.clear()
            for _ in range(min(steps, eager_after)):
                step_fn()
            barrier()
        return dt, step_ms, g_

    def solve_us_stats(be_):
        """per-replan device time of the interior-point workgroup (nep_stats.solve_us) of the last step"""
        us = be_.solutions(timing=True)["stats"]["solve_us"]
        return {"p50": float(np.percentile(us, 50)), "p90": float(np.percentile(us, 90)), "p99": float(np.percentile(us, 99)),
                "max": float(us.max()), "mean": float(us.mean()), "n": int(us.size)}

    out = None
    if not args.config5_only:
        # Weak scaling: args.scenes scenes in flight per GPU, so S = scenes * world scenes in total; the
        # agents of EVERY scene are block-sharded over the ranks (configs[3]: 64 agents, 8 per GPU on 8
        # GPUs), which keeps scenes * agents replans per GPU per step at any N.
        N, M, S = args.agents, args.obstacles, args.scenes * world
        first_local, n_local = ndist.shard(N, world, rank)
        # each rank generates its share of the seeded scenes (seeds 0..S-1 overall), then they are shared
        host_cores = os.cpu_count() or 1
        mine = scene.make_scenes(N, M, range(rank * args.scenes, (rank + 1) * args.scenes),
                                 workers=min(args.scenes, max(1, (host_cores // (4 if want_c5 else 2)) // world), 64))
        scene0 = mine[0] if rank == 0 else scene.make_scene(N, M, seed=0)
        if c5_futs is not None:
            # the config-5 pool has been running beside this one; nothing is timed while host processes are still busy (a first
            # version let it run under the timed legs: the GPU time per step was unchanged, the host's share of a 20-step region
            # went from 1 % to 60 %)
            t_c5 = time.perf_counter()
            c5_made = [f.result() for f in c5_futs]
            c5_pool.shutdown(); c5_pool = None
            c5_wait = time.perf_counter() - t_c5
        p = scene0["par"]
        # every scene has its own static obstacles (drawn first from its seed, so any rank can rebuild any scene's set)
        all_statics = [mine[s - rank * args.scenes]["statics"] if rank * args.scenes <= s < (rank + 1) * args.scenes
                       else scene.scene_statics(N, M, s, par=p) for s in range(S)]
        statics = all_statics[0]
        com_l, gue_l = ndist.stack_scenes(mine)

        def share(arr):                      # [scenes per GPU][N] per rank -> [S][N] on every rank
            if world == 1:
                return arr
            t = torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1).copy())
            if dist_backend == "nccl":
                t = t.to(dev)
                o_ = torch.empty(world * t.numel(), dtype=torch.uint8, device=dev)
                tdist.all_gather_into_tensor(o_, t)
                o_ = o_.cpu()
            else:
                pieces = [torch.empty_like(t) for _ in range(world)]
                tdist.all_gather(pieces, t)
                o_ = torch.cat(pieces)
            return o_.numpy().view(arr.dtype).reshape((S,) + arr.shape[1:])
        com, gue = share(com_l), share(gue_l)

        sharded_hulls = world > 1 and args.exchange == "hulls"
        native = sharded_hulls and not args.exchange_torch and not args.safety and dist_backend == "nccl"
        C = args.chunks if (sharded_hulls and not args.safety and S % max(args.chunks, 1) == 0) else 1
        Sc = S // C
        # one handle per scene chunk (C == 1: all scenes); chunk k holds scenes [k*Sc, (k+1)*Sc)
        bes = [BatchBackend(p, statics, first_local=first_local, n_local=n_local, n_scenes=Sc, device=dev) for _ in range(C)]
        be = bes[0]
        for k, b in enumerate(bes):
            b.set_line_cull(args.cull_radius)
            for s_ in range(Sc):
                if len(all_statics[k * Sc + s_]) != len(statics):
                    raise SystemExit("scene %d drew %d static obstacles instead of %d" % (k * Sc + s_, len(all_statics[k * Sc + s_]), len(statics)))
                b.set_scene_statics(s_, all_statics[k * Sc + s_])
        d_committed = be.to_device(com) if C == 1 else None
        d_guess_c = [bes[k].to_device(np.ascontiguousarray(gue[k * Sc:(k + 1) * Sc, first_local:first_local + n_local])) for k in range(C)]
        d_guess = d_guess_c[0]
        ex = ndist.RoundExchange(S, N, world, rank, device=dev)
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
        fe_starts = share(np.stack([scene.frontend_starts(s) for s in mine])) if args.frontend else None       # [S][N]
        d_fe_start_c = [bes[k].to_device(np.ascontiguousarray(fe_starts[k * Sc:(k + 1) * Sc, first_local:first_local + n_local])) for k in range(C)] if args.frontend else None
        d_fe_start = d_fe_start_c[0] if args.frontend else None
        d_fe_res_c = [torch.zeros(Sc * n_local * abi.FE_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev) for _ in range(C)] if args.frontend else None
        d_fe_res = d_fe_res_c[0] if args.frontend else None
        pending = [None] * C
        _ev_lists = {"hull": hull_ev, "wait": gather_ev, "frontend": fe_ev}
        rounds = None
        nranks = None
        if sharded_hulls and not args.safety:
            with _stdout_to_stderr():                      # (a second communicator: RCCL may print again)
                rounds = ndist.ShardedRounds(bes, d_local_c, d_guess_c, world, rank, native=native,
                                             fe=(fe_cfg, d_fe_start_c, d_fe_res_c) if args.frontend else None,
                                             timer=lambda name: _timed(_ev_lists[name]))
            hxs = rounds.hx
            if native:
                mine_n = rounds.native.nranks()
                t = torch.tensor([mine_n], dtype=torch.int64, device=dev)
                got = [torch.zeros_like(t) for _ in range(world)]
                tdist.all_gather(got, t)
                nranks = [int(x.item()) for x in got]
                print("[bench] rank %d of %d: native RCCL communicator has %d ranks" % (rank, world, mine_n), file=sys.stderr)
                if mine_n != world or any(n_ != world for n_ in nranks):
                    raise SystemExit("native RCCL communicator: ranks joined %r, expected %d on every rank" % (nranks, world))
        elif use_dist and dist_backend == "nccl":
            nranks = [tdist.get_world_size()] * world      # (N = 1: the one-rank process group this run created)

        def start_exchange(k, src):
            """hulls of my agents' committed trajectories (chunk k) -> start the all-gather of the hull blocks"""
            e0 = ev()
            bes[k].hulls(src, d_guess_c[k], hxs[k].local)
            hull_ev.append((e0, ev()))
            pending[k] = hxs[k].gather_async()

        def step():
            if rounds is not None:
                rounds.step()          # chunks pipelined: one chunk's all-gather runs under another chunk's kernels (dist.ShardedRounds)
                return
            if sharded_hulls:
                start_exchange(0, d_local_c[0])
                e1 = ev()
                pending[0].wait()
                gather_ev.append((e1, ev()))
                be.replan_hulls(hxs[0].blocks, d_guess)
            elif args.frontend:
                e0 = ev()
                be.frontend(fe_cfg, d_committed, d_fe_start, d_guess, d_fe_res)     # hulls + beam search -> d_guess
                fe_ev.append((e0, ev()))
                be.replan(None, d_guess)                                             # separator + QP on the same hulls
            else:
                be.replan(d_committed, d_guess)
            if not args.safety:
                e1 = ev()
                ex.gather(be.d_commit, d_committed)
                gather_ev.append((e1, ev()))
                return
            ex.gather(be.d_commit, d_new)                   # everyone's new trajectory
            e0 = ev()
            be.safety_commit(d_committed, d_new, d_guess, d_committed_next, d_accept)      # d_guess: [S][n_local], as passed to the replan
            safety_ev.append((e0, ev()))
            d_committed.copy_(d_committed_next)
            if sharded_hulls:
                d_local_c[0].view(S, n_local * REC).copy_(d_committed.view(S, N * REC)[:, first_local * REC:(first_local + n_local) * REC])

        # a step can be captured when it is a fixed launch sequence without host decisions: one GPU, or several with the native
        # exchange (the torch path's work handles and the gloo / records paths are host-driven)
        can_graph = (world == 1 and rounds is None) or (rounds is not None and native)
        graph_plain = can_graph and not args.frontend and not args.safety

        for _ in range(args.warmup):
            step()
        rccl_one_rank_ok = None
        if world == 1 and use_dist and dist_backend == "nccl" and not (args.safety or args.frontend) and C == 1:
            # one rank: the timed steps copy (nothing to exchange); the collective path itself — the all-gather of the committed
            # records through RCCL — is exercised once here, outside the timed region, and must give the same bytes
            chk = torch.empty_like(d_committed)
            ex.gather(be.d_commit, chk, collective=True)
            torch.cuda.synchronize(dev)
            rccl_one_rank_ok = bool(torch.equal(chk, be.d_commit.view_as(chk)))
            if own_group and os.environ.get("NEP_BENCH_PG_TEARDOWN"):
                # development aid.  With a live RCCL communicator in the process a hipMemsetAsync node in a replayed graph costs
                # ~0.27 ms (found on the config-5 step: 1.55 instead of 1.28 ms with identical kernel times; the presolve's redo
                # counters are now zeroed by a kernel and the step has no memset node): this tears the one-rank group down early
                tdist.destroy_process_group(); use_dist = False; rccl_torn_down = True
        # ---- headline: exactly --steps steps -------------------------------------------------------------------------
        dt, step_ms, graph = run_leg(step, bes, args.steps, 0, graph_ok=graph_plain, clear=(safety_ev, hull_ev, gather_ev))
        dt_local = last_wall[0]
        qp_ms, n_launch = be.kernel_time_ms(2)           # per launch of one chunk (chunk 0)
        hull_ms, _ = be.kernel_time_ms(0)
        if sharded_hulls:
            hull_ms = mean_ms(hull_ev)
        sep_ms, _ = be.kernel_time_ms(1)
        seq_ms, _ = be.kernel_time_ms(3)
        for b in bes:
            b.enable_timing(False)
        sol = np.concatenate([b.solutions() for b in bes])
        active = active_summary(be) if C == 1 else None          # (chunk 0's handle when the scenes are chunked: see sharding)
        solve_us = solve_us_stats(be)
        status = sol["stats"]["status"].astype(int)
        iters = sol["stats"]["iters"].astype(int)
        n_states = int(sol[0]["n_states"])
        replans_per_step = S * N
        value = replans_per_step * args.steps / dt

        def leg_record(dt_, steps_, step_ms_, **kw):
            r = {"value": replans_per_step * steps_ / dt_, "unit": "replans/s", "steps": steps_, "ms_per_step": dt_ / steps_ * 1e3,
                 "step_ms": {"p50": float(np.percentile(step_ms_, 50)), "p99": float(np.percentile(step_ms_, 99)), "max": float(step_ms_.max())}}
            r.update(kw)
            return r

        # ---- the same step over a longer timed region, and with the launch order off -----------------------------------
        long_run = order_off = ref_tol = None
        if not args.no_extra_legs and graph_plain:
            dt_l, ms_l, _ = run_leg(step, bes, aux_steps, 2, graph_ok=True, eager_after=0)
            long_run = leg_record(dt_l, aux_steps, ms_l, note="the headline's step, %d steps between the barriers" % aux_steps)
            for b in bes:
                b.set_launch_order(False)
            dt_o, ms_o, _ = run_leg(step, bes, aux_steps, 2, graph_ok=True, eager_after=10)
            qp_o, _ = be.kernel_time_ms(2)
            order_off = leg_record(dt_o, aux_steps, ms_o, qp_ms=qp_o, solve_us=solve_us_stats(be),
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
            dt_t, ms_t, _ = run_leg(step, bes, aux_steps, 3, graph_ok=True, eager_after=10)
            qp_t, _ = be.kernel_time_ms(2)
            sol_t = be.solutions()
            ref_tol = leg_record(dt_t, aux_steps, ms_t, qp_ms=qp_t, solve_us=solve_us_stats(be), ipm_iters_mean=float(sol_t["stats"]["iters"].mean()),
                                 residual_tol=1e-6, gap_tol=1e-8,
                                 note="nep_batch_set_tolerances(1e-6, 1e-8): the strict tests at Gurobi's defaults (FeasibilityTol = OptimalityTol = "
                                      "1e-6, BarConvTol = 1e-8), which is where the reference's solver stops (PolySolverGurobi sets OutputFlag and "
                                      "TimeLimit only, solver_gurobi_poly.cpp:811-812); the headline and every other leg use 1e-9 / 1e-10", **status_counts(sol_t))
            for b in bes:
                b.enable_timing(False); b.set_tolerances(1e-9, 1e-10)
            for _ in range(2):
                step()

        # ---- presolve: the same steps with the verified line presolve on ------------------------------------
        presolve = None
        if args.cull_radius == 0.0 and args.presolve_radius > 0.0 and not args.no_extra_legs:
            for b in bes:
                b.set_line_cull(args.presolve_radius)
            dt2, ms2, _ = run_leg(step, bes, aux_steps, max(args.warmup, 2), graph_ok=graph_plain)
            qp2, _ = be.kernel_time_ms(2)
            for b in bes:
                b.enable_timing(False)
            sol2 = np.concatenate([b.solutions() for b in bes])
            presolve = leg_record(dt2, aux_steps, ms2, cull_radius_m=args.presolve_radius, qp_ms=qp2,
                                  rows_solved_mean=float(sol2["stats"]["n_rows"].mean()),
                                  ipm_iters_mean=float(sol2["stats"]["iters"].mean()), ipm_iters_max=int(sol2["stats"]["iters"].max()),
                                  solved_without_iteration=int((sol2["stats"]["iters"] == 0).sum()), solve_us=solve_us_stats(be),
                                  active_rows=active_summary(be),
                                  note="verified shortcuts, same optimum as the headline run: (1) lines farther than the radius from the guess are parked, "
                                       "checked against the solution and the QP re-solved with all of them on a violation; (2) if the minimiser of the "
                                       "cost without inequality rows satisfies every row it is the optimum (KKT with zero multipliers) and no "
                                       "interior-point iteration runs", **status_counts(sol2))
            for b in bes:
                b.set_line_cull(0.0)

        # ---- chain (single GPU): front-end beam search from point A and the goal, separating lines + QP on the same hulls,
        # post-solve safety check and commit — the guesses are device-made -----------------------------------------------
        chain = moving = crossing = None
        if extra and not args.no_chain and C == 1:
            cfg_fe = scene.frontend_cfg(p, beam_width=args.beam)
            be.set_line_cull(args.chain_cull_radius)
            starts_np = np.stack([scene.frontend_starts(s_) for s_ in mine])
            d_st = be.to_device(starts_np)
            d_gfe = torch.zeros_like(d_guess)
            d_res = torch.zeros(S * N * abi.FE_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            d_com2 = be.to_device(com); d_nxt = torch.empty_like(d_com2); d_acc = torch.zeros(S * N, dtype=torch.int32, device=dev)
            fe2, sf2 = [], []

            def chain_step():
                e0 = ev()
                be.frontend(cfg_fe, d_com2, d_st, d_gfe, d_res)
                fe2.append((e0, ev()))
                be.replan(None, d_gfe)                       # (a failed / empty replan's commit slot carries the record of d_com2 over)
                e1 = ev()
                be.safety_commit(d_com2, be.d_commit, d_gfe, d_nxt, d_acc)
                sf2.append((e1, ev()))
                d_com2.copy_(d_nxt)
            dt3, ms3, _ = run_leg(chain_step, [be], aux_steps, max(args.warmup, 2), clear=(fe2, sf2))
            qp3, _ = be.kernel_time_ms(2); sep3, _ = be.kernel_time_ms(1)
            be.enable_timing(False)
            sol3 = be.solutions()
            res3 = d_res.cpu().numpy().view(abi.FE_RESULT_DTYPE)
            chain = leg_record(dt3, aux_steps, ms3,
                               kernel_ms={"frontend_with_hulls": mean_ms(fe2), "separator": sep3, "qp": qp3, "safety": mean_ms(sf2)},
                               beam_width=args.beam, frontend_goal_reached=int((res3["status"] == 1).sum()), frontend_no_solution=int((res3["status"] == 3).sum()),
                               ipm_iters_mean=float(sol3["stats"]["iters"].mean()), ipm_iters_max=int(sol3["stats"]["iters"].max()),
                               lp_failed=int(sol3["stats"]["n_lp_failed"].sum()), accepted_frac=float(d_acc.float().mean().item()),
                               solve_us=solve_us_stats(be), terminal_ball_rows=int(sol3["stats"]["qc_active"].sum()),
                               ipm_iters_quantiles=quantiles(sol3["stats"]["iters"]), line_cull_radius_m=args.chain_cull_radius,
                               rows_solved_mean=float(sol3["stats"]["n_rows"].mean()), presolve_redo_last_step=be.redo_count(),
                               active_rows=active_summary(be),
                               note="front-end beam search -> separating lines -> QP -> safety check + commit, every step; the guesses are the "
                                    "device-made lattice paths (they end at cruise speed and cut corners around obstacles), not the scene's; "
                                    "point A stays where it is, so after a few steps every step poses the same problems", **status_counts(sol3))

            # ---- moving: the closed loop on the device.  After the commit, point A of the next round is taken half a second
            # (T_span: one interval) ahead on every agent's committed trajectory (nep_batch_next_starts) and an agent that has
            # arrived swaps its goal with its starting point, so the fleets keep flying: every step poses new problems, and the
            # launch-order key of a slot is the measured time of the SAME AGENT's previous, different replan --------------
            cfg_mv = scene.frontend_cfg(p, beam_width=args.beam, pad_hold=1)

            def closed_loop(starts_in, com_in, note):
                """one closed-loop leg from the given points A / goals and committed records -> (record, rerun(cull) -> record)"""
                d_st_m = be.to_device(starts_in)
                alt_np = np.ascontiguousarray(starts_in["pos"].reshape(S * N, 3))          # the way back: where the agent started
                d_alt = torch.from_numpy(alt_np.copy()).to(dev)
                d_com3 = be.to_device(com_in); d_nxt3 = torch.empty_like(d_com3)
                fe3, sf3 = [], []

                def moving_step():
                    e0 = ev()
                    be.frontend(cfg_mv, d_com3, d_st_m, d_gfe, d_res)
                    fe3.append((e0, ev()))
                    be.replan(None, d_gfe)
                    e1 = ev()
                    be.safety_commit(d_com3, be.d_commit, d_gfe, d_nxt3, d_acc)
                    d_com3.copy_(d_nxt3)
                    be.next_starts(d_com3, p.T_span, d_st_m, d_alt, 0.5)
                    sf3.append((e1, ev()))

                def run(cull):
                    be.set_line_cull(cull)
                    d_st_m.copy_(be.to_device(starts_in)); d_alt.copy_(torch.from_numpy(alt_np.copy()).to(dev)); d_com3.copy_(be.to_device(com_in))
                    dt4, ms4, _ = run_leg(moving_step, [be], aux_steps, max(args.warmup, 2), clear=(fe3, sf3))
                    qp4, _ = be.kernel_time_ms(2); sep4, _ = be.kernel_time_ms(1)
                    be.enable_timing(False)
                    sol4 = be.solutions(); res4 = d_res.cpu().numpy().view(abi.FE_RESULT_DTYPE)
                    st_now = d_st_m.cpu().numpy().view(abi.FE_START_DTYPE)
                    moved = np.hypot(*(st_now["pos"][:, :2] - starts_in.reshape(-1)["pos"][:, :2]).T)
                    swaps = int((np.abs(st_now["goal"] - starts_in.reshape(-1)["goal"]).max(axis=1) > 0).sum())
                    return leg_record(dt4, aux_steps, ms4,
                                      kernel_ms={"frontend_with_hulls": mean_ms(fe3), "separator": sep4, "qp": qp4, "safety_commit_next_start": mean_ms(sf3)},
                                      beam_width=args.beam, frontend_goal_reached=int((res4["status"] == 1).sum()), frontend_no_solution=int((res4["status"] == 3).sum()),
                                      ipm_iters_mean=float(sol4["stats"]["iters"].mean()), ipm_iters_max=int(sol4["stats"]["iters"].max()),
                                      lp_failed=int(sol4["stats"]["n_lp_failed"].sum()), accepted_frac=float(d_acc.float().mean().item()),
                                      K_mean=float(sol4["K"].mean()), solve_us=solve_us_stats(be),
                                      terminal_ball_rows=int(sol4["stats"]["qc_active"].sum()), lines_mean=float(sol4["stats"]["n_lines"].mean()),
                                      ipm_iters_quantiles=quantiles(sol4["stats"]["iters"]), line_cull_radius_m=cull,
                                      rows_solved_mean=float(sol4["stats"]["n_rows"].mean()), presolve_redo_last_step=be.redo_count(),
                                      simulated_seconds=float(st_now["t_start"].max() - starts_in["t_start"].max()),
                                      displacement_m_mean=float(moved.mean()), agents_with_swapped_goal=swaps,
                                      failed_frac=float((sol4["stats"]["status"] == 2).mean()), active_rows=active_summary(be),
                                      note=note, **status_counts(sol4))
                return run
            run_moving = closed_loop(starts_np, com,
                                     "closed loop on the device, one HIP graph per round: front end -> lines -> QP -> safety check + commit -> point A of "
                                     "the next round 0.5 s ahead on the committed trajectory; arrived agents turn around.  Every step solves NEW problems; the "
                                     "launch-order predictor is the same agent's previous replan")
            moving = run_moving(args.chain_cull_radius)
            # ---- crossing: the same closed loop on the hard variant of every scene — all 64 agents start at rest on the base circle
            # and fly to the antipodal point, so the whole fleet meets in the middle (and turns around on arrival) --------------
            cross = [scene.crossing_scene(s_) for s_ in mine]
            run_cross = closed_loop(np.stack([c_[0] for c_ in cross]), np.stack([c_[1] for c_ in cross]),
                                    "the closed loop of `moving` on the circle-swap variant of the same scenes: every agent starts at rest on the base circle, "
                                    "its goal is the antipodal point (the start of the agent opposite), arrived agents turn around — the fleet crosses the middle of "
                                    "the world together, against the scene's static obstacles.  The hard leg: see active_rows, failed_frac, ipm_iters")
            crossing = run_cross(args.chain_cull_radius)
            # ---- moving, as two scene groups on two streams inside the one captured step: the tail of one group's kernels (the QP
            # launch ends with a handful of failing solves of ~1.2 ms each on an otherwise empty GPU) runs beside the other group's
            # kernels.  Same scenes, same results; what a deployment that keeps several fleets in flight does ------------------
            if S % 2 == 0 and not args.no_graph:
                Sg = S // 2
                gb = []
                for k_ in range(2):
                    b_ = BatchBackend(p, statics, n_scenes=Sg, device=dev)
                    for s_ in range(Sg):
                        b_.set_scene_statics(s_, all_statics[k_ * Sg + s_])
                    b_.set_line_cull(args.chain_cull_radius)
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
                dtg, msg, gg = run_leg(two_group_step, gb, n_rep, max(args.warmup, 2), eager_after=0)
                solg = np.concatenate([b_.solutions() for b_ in gb])
                moving["two_groups"] = {"value": replans_per_step * RG * n_rep / dtg, "unit": "replans/s", "rounds": RG * n_rep, "ms_per_round": dtg / (RG * n_rep) * 1e3,
                                        "rounds_per_graph": RG, "graph": gg is not None, "ipm_iters_mean": float(solg["stats"]["iters"].mean()),
                                        "note": "the moving leg with the scenes in two groups of %d on two streams, %d rounds of each group inside one captured graph, the "
                                                "second group started when the first one's front end is done: one group's QP tail (a handful of failing solves on an "
                                                "otherwise empty GPU) runs beside the other group's front end" % (Sg, RG), **status_counts(solg)}
                for b_ in gb:
                    b_.close()
            # ---- both again with the verified presolve (what a deployment runs, and the handle's default at config-5 size) ----
            if args.chain_cull_radius == 0.0 and args.presolve_radius > 0.0:
                be.set_line_cull(args.presolve_radius)
                d_com2.copy_(be.to_device(com))
                dt3p, ms3p, _ = run_leg(chain_step, [be], aux_steps, max(args.warmup, 2), clear=(fe2, sf2))
                qp3p, _ = be.kernel_time_ms(2); sep3p, _ = be.kernel_time_ms(1)
                be.enable_timing(False)
                sol3p = be.solutions()
                chain["with_presolve"] = leg_record(dt3p, aux_steps, ms3p, cull_radius_m=args.presolve_radius,
                                                    kernel_ms={"frontend_with_hulls": mean_ms(fe2), "separator": sep3p, "qp": qp3p, "safety": mean_ms(sf2)},
                                                    rows_solved_mean=float(sol3p["stats"]["n_rows"].mean()), ipm_iters_mean=float(sol3p["stats"]["iters"].mean()),
                                                    presolve_redo_last_step=be.redo_count(), **status_counts(sol3p))
                moving["with_presolve"] = run_moving(args.presolve_radius)
                crossing["with_presolve"] = run_cross(args.presolve_radius)
            be.set_line_cull(0.0)

        # ---- single_scene: ONE fleet -------------------------------------------------------------------------------------
        single = None
        if extra:
            b1 = BatchBackend(p, statics, n_scenes=1, device=dev)
            d_c1 = b1.to_device(com[0]); d_g1 = b1.to_device(gue[0])

            def single_step():
                b1.replan(d_c1, d_g1)
                d_c1.copy_(b1.d_commit)
            dt5, ms5, _ = run_leg(single_step, [b1], aux_steps, max(args.warmup, 2))
            k1 = {n_: b1.kernel_time_ms(i_)[0] for i_, n_ in ((0, "hull"), (1, "separator"), (2, "qp"), (3, "sequence"))}
            b1.enable_timing(False)
            single = {"value": N * aux_steps / dt5, "unit": "replans/s", "steps": aux_steps, "round_ms": dt5 / aux_steps * 1e3,
                      "step_ms": {"p50": float(np.percentile(ms5, 50)), "p99": float(np.percentile(ms5, 99)), "max": float(ms5.max())},
                      "kernel_ms": k1, "solve_us": solve_us_stats(b1),
                      "note": "one scene of %d agents per launch sequence: the latency of one bulk-synchronous round of a single fleet and that "
                              "fleet's throughput; `value` at the top keeps %d independent scenes in flight" % (N, S)}
            b1.close()

    # ---- the other single-GPU configs of BASELINE.json (configs[1]: 5 agents obstacle-free; configs[2]: 8 agents + 20 obstacles):
    # the batched step at those sizes, and the DROP-IN call — the per-agent handle behind include/neptune_poly_solver.hpp, what
    # Neptune::replanCB would call once per replan (neptune.cpp:1504-1528) — timed inside the library ------------------------
    small_configs = per_agent = None
    if extra and rank == 0 and not args.config5_only:
        from neptune_amd.backend import PolySolver, hulls_batch as hulls_of
        small_configs = {}
        for name, n_a, n_o, n_sc in (("config2_5_agents", 5, 0, 1024), ("config3_8_agents_20_obstacles", 8, 20, 512)):
            scs = scene.make_scenes(n_a, n_o, range(n_sc), workers=min(n_sc, max(1, host_cores // 2), 64))
            pc = scs[0]["par"]
            bc = BatchBackend(pc, scs[0]["statics"], n_scenes=n_sc, device=dev)
            for s_ in range(n_sc):
                if len(scs[s_]["statics"]) != len(scs[0]["statics"]):
                    raise SystemExit("%s: scene %d drew another number of static obstacles" % (name, s_))
                bc.set_scene_statics(s_, scs[s_]["statics"])
            com_c, gue_c = ndist.stack_scenes(scs)
            d_cc = bc.to_device(com_c); d_gc = bc.to_device(gue_c)

            def small_step():
                bc.replan(d_cc, d_gc)
                d_cc.copy_(bc.d_commit)
            dtc, msc, _ = run_leg(small_step, [bc], aux_steps, max(args.warmup, 2))
            kc = {n_: bc.kernel_time_ms(i_)[0] for i_, n_ in ((0, "hull"), (1, "separator"), (2, "qp"), (3, "sequence"))}
            bc.enable_timing(False)
            solc = bc.solutions()
            small_configs[name] = {"value": n_a * n_sc * aux_steps / dtc, "unit": "replans/s", "steps": aux_steps, "ms_per_step": dtc / aux_steps * 1e3,
                                   "scenes_in_flight": n_sc, "replans_per_step": n_a * n_sc, "kernel_ms": kc, "solve_us": solve_us_stats(bc),
                                   "ipm_iters_mean": float(solc["stats"]["iters"].mean()), "lines_mean": float(solc["stats"]["n_lines"].mean()),
                                   "active_rows": active_summary(bc), **status_counts(solc)}
            bc.close()
        per_agent = {"note": "the six-call drop-in sequence of ONE replan (setInitTrajectory -> setHulls -> setHullsNoInflation -> setEntStateVector -> optimize "
                             "-> generatePwpOut, neptune.cpp:1514-1527) through the per-agent C ABI with host buffers, blocking, as a C++ caller's clock sees it "
                             "(nep_backend_debug_time_sequence: no Python between the calls): one host-to-device copy, separator + QP kernels, one device-to-host "
                             "copy.  The reference's budget for the same call is TimeLimit 0.05 s",
                     "iterations_per_agent": 200}
        for name, n_a, n_o in (("config2_5_agents", 5, 0), ("config3_8_agents_20_obstacles", 8, 20), ("config4_64_agents_20_obstacles", N, M)):
            sc_ = scene.make_scene(n_a, n_o, seed=0) if (n_a, n_o) != (N, M) else scene0
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

    # ---- config5: BASELINE configs[4], 256 agents + 100 obstacles, enable_entangle_check on -------------------------------
    config5 = None
    if want_c5 and rank == 0:
        import dataclasses
        if c5_pool is not None:                       # (--config5-only)
            t_c5 = time.perf_counter()
            c5_made = [f.result() for f in c5_futs]
            c5_pool.shutdown()
            c5_wait = time.perf_counter() - t_c5
        if c5_cache and not os.path.exists(c5_cache):
            import pickle
            pickle.dump(c5_made, open(c5_cache, "wb"))
        made, t_wait = c5_made, c5_wait
        S5, N5 = len(made), 256
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
        # scene and stay (a committed record as the QP kernel writes it carries the base only): position and polynomial
        # are copied over, id / flags / bend points are left alone
        f5 = abi.TRAJ_REC_DTYPE.fields
        o_pos, o_bend, o_pwp = f5["pos"][1], f5["bend"][1], f5["pwp"][1]
        REC5 = abi.TRAJ_REC_DTYPE.itemsize
        v_c5 = d_c5.view(S5 * N5, REC5)

        def c5_step():
            b5.replan(d_c5, d_g5, d_ent=d_e5)
            cm = b5.d_commit.view(S5 * N5, REC5)
            v_c5[:, o_pos:o_bend].copy_(cm[:, o_pos:o_bend])
            v_c5[:, o_pwp:].copy_(cm[:, o_pwp:])

        def c5_leg(label):
            dt_, ms_, _ = run_leg(c5_step, [b5], aux_steps if not args.config5_only else args.steps, max(args.warmup, 2), eager_after=10)
            steps_ = aux_steps if not args.config5_only else args.steps
            k_ = {n_: b5.kernel_time_ms(i_)[0] for i_, n_ in ((0, "hull"), (1, "separator"), (2, "qp"), (3, "sequence"))}
            b5.enable_timing(False)
            s_ = b5.solutions()
            return dt_, steps_, ms_, k_, s_
        cull5 = b5.line_cull(); kern5 = b5.qp_kernel_name()
        dt6, steps6, ms6, k6, sol6 = c5_leg("default")
        us6 = solve_us_stats(b5)
        redo6 = {"replans": b5.redo_count(), **b5.redo_reasons}
        _, hn5 = b5.debug_hulls(0)
        ns5 = int(sol6[0]["n_states"])
        ent_b = 4.0 * 8 * N5 + 16.0 * bend5[0].sum()                # the dense case block of one replan + every agent's bend points
        bytes5 = algorithmic_bytes(p5, sc5[0], hn5, ns5, ent_bytes=ent_b)
        dom = max(("hull", "separator", "qp"), key=lambda n_: k6[n_])
        dom_name = {"hull": "hull_group_kernel", "separator": "separator_packed_kernel" if cull5 > 0.0 else "separator_kernel", "qp": kern5}[dom]
        ach_dom = bytes5 * S5 * N5 / (k6[dom] * 1e-3) / 1e9 if k6[dom] > 0 else 0.0
        traffic5 = measured_traffic("nep::" + dom_name, "pmc_summary_config5_latest.txt")
        config5 = {"value": S5 * N5 * steps6 / dt6, "unit": "replans/s", "steps": steps6, "ms_per_step": dt6 / steps6 * 1e3,
                   "step_ms": {"p50": float(np.percentile(ms6, 50)), "p99": float(np.percentile(ms6, 99)), "max": float(ms6.max())},
                   "workload": "256 agents + 100 static obstacles, enable_entangle_check on (synthetic ent_state: one active case for 10 %% of the agent pairs, "
                               "2-4 bend points per agent, SURVEY 8d), K=8, %d seeded scenes in flight (seeds 0..%d)" % (S5, S5 - 1),
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
                           "solution, re-solved with all rows on a violation), interior point on %s" % (cull5, kern5), **status_counts(sol6)}
        if not args.config5_only:
            # every row through the interior point (presolve explicitly off): the LDS placement with its global spill
            b5.set_line_cull(0.0)
            kern5f = b5.qp_kernel_name()
            d_c5.copy_(b5.to_device(com5))
            dt7, steps7, ms7, k7, sol7 = c5_leg("full_rows")
            config5["full_rows"] = {"value": S5 * N5 * steps7 / dt7, "unit": "replans/s", "steps": steps7, "ms_per_step": dt7 / steps7 * 1e3,
                                    "qp_kernel": kern5f, "kernel_ms": k7, "solve_us": solve_us_stats(b5),
                                    "rows_solved_mean": float(sol7["stats"]["n_rows"].mean()), "ipm_iters_mean": float(sol7["stats"]["iters"].mean()),
                                    "active_rows": active_summary(b5),
                                    "roofline": {"bound": "hbm", "peak": 8000.0, "unit": "GB/s", "algorithmic_bytes_per_replan": bytes5, "replans_per_launch": S5 * N5,
                                                 **{kn: {"ms": k7[kk], "achieved": bytes5 * S5 * N5 / (k7[kk] * 1e-3) / 1e9 if k7[kk] > 0 else 0.0,
                                                         "frac": bytes5 * S5 * N5 / (k7[kk] * 1e-3) / 1e9 / 8000.0 if k7[kk] > 0 else 0.0}
                                                    for kn, kk in (("separator_kernel", "separator"), (kern5f, "qp"))}},
                                    "note": "nep_batch_set_line_cull(0): every separating-line row through the interior point", **status_counts(sol7)}
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
                e0 = ev()
                b5.frontend_ent(cfg5, d_cc5, d_st5, d_gf5, d_res5, d_case5)
                fe5.append((e0, ev()))
                b5.replan(None, d_gf5, d_ent=d_case5)
                e1 = ev()
                b5.safety_commit_ent(d_cc5, b5.d_commit, d_gf5, d_nx5, d_ac5)
                sf5.append((e1, ev()))
                d_cc5.copy_(d_nx5)
            steps8 = max(20, aux_steps // 10)
            dt8, ms8, _ = run_leg(c5_chain_step, [b5], steps8, 2, eager_after=5, clear=(fe5, sf5))
            k8 = {n_: b5.kernel_time_ms(i_)[0] for i_, n_ in ((1, "separator"), (2, "qp"))}
            b5.enable_timing(False)
            sol8 = b5.solutions(); res8 = d_res5.cpu().numpy().view(abi.FE_RESULT_DTYPE)
            config5["chain"] = {"value": S5 * N5 * steps8 / dt8, "unit": "replans/s", "steps": steps8, "ms_per_step": dt8 / steps8 * 1e3,
                                "kernel_ms": {"frontend_ent_with_hulls": mean_ms(fe5), "separator": k8["separator"], "qp": k8["qp"], "safety_ent": mean_ms(sf5)},
                                "beam_width": args.beam, "frontend_goal_reached": int((res8["status"] == 1).sum()), "frontend_no_solution": int((res8["status"] == 3).sum()),
                                "children_pruned_by_the_entangle_check": int(res8["n_entangled"].sum()), "ent_overflow": int(res8["ent_overflow"].sum()),
                                "big_records": {"searches": int(((res8["_pad"].astype(np.int64) >> 8) > 0).sum()), "children": int((res8["_pad"].astype(np.int64) >> 8).sum()),
                                                "note": "searches in which a child's entangle state outgrew the fixed record (40 crossings, 32 new ones per sampled step, 8 bend points): re-run by the big-record instantiation after the launch, bounded by the reference's own rule (num_agents + statics) only; ent_overflow counts the searches the POOL of big records failed (0)"},
                                "active_entangle_cases": int((d_case5 != 0).sum().item()),
                                "ipm_iters_mean": float(sol8["stats"]["iters"].mean()), "accepted_frac": float(d_ac5.float().mean().item()),
                                "solve_us": solve_us_stats(b5),
                                "note": "frontend_kernel<true> (entangle states per search node) -> separator + QP on device-made guesses and device-made "
                                        "entangle cases -> safety check with entangleCheckGivenPwp + commit", **status_counts(sol8)}
        b5.close()
        if args.config5_only:
            print(json.dumps({"metric": "backend_replans_per_sec", "config5": config5, "graph_notes": graph_notes}))

    per_rank = None
    if world > 1 and not args.config5_only:
        # every rank's own view of the step (the line's kernel_ms is rank 0's): kernel times and what its stream waited for
        mine_rec = {"rank": rank, "device": torch.cuda.get_device_name(dev), "kernel_ms": {"hull": hull_ms, "separator": sep_ms, "qp": qp_ms, "sequence": seq_ms,
                                                                                             "exchange_wait": mean_ms(gather_ev)},
                    "step_ms_p50": float(np.percentile(step_ms, 50)), "step_ms_max": float(step_ms.max()), "wall_s": dt_local}
        per_rank = [None] * world
        tdist.all_gather_object(per_rank, mine_rec)
    if rank == 0 and not args.config5_only:
        if C == 1 and not sharded_hulls and not args.frontend:
            _, hn = be.debug_hulls(0)          # vertex counts of scene 0 as the last timed launch saw them
        else:
            from neptune_amd.backend import hulls_batch
            _, hn, _, _ = hulls_batch(com[0], float(gue[0, 0]["t_start"]), p.num_pol, p.T_span, p.drone_radius)   # scene 0 at the start
        bytes_per_replan = algorithmic_bytes(p, scene0, hn, n_states)
        launch_replans = Sc * n_local
        achieved = bytes_per_replan * launch_replans / (qp_ms * 1e-3) / 1e9 if qp_ms > 0 else 0.0
        K8 = int(sol[0]["K"])
        # the same compulsory bytes split by the kernel that moves them (per replan), each over its own duration
        Kg = int(scene0["guesses"][0]["K"]); L_mean = float(sol["stats"]["n_lines"].mean())
        b_guess = 8 * (12 * Kg + (Kg + 1)); b_hull_v = 16.0 * hn[:, :Kg].sum() / N          # one agent's hull vertices
        b_rec = 8 * (13 * Kg + 1)                                                           # one committed trajectory
        b_static = 16 * sum(len(s_) for s_ in statics); b_out = 8 * (12 * Kg + 1) + 4 + 96 * n_states
        per_kernel_bytes = {"hull": (b_rec + b_hull_v) * (Sc * (n_local if sharded_hulls else N)) / launch_replans,   # records in, hull vertices out
                            "separator": b_guess + b_hull_v * (N - 1) + b_static + 16 * N + 24 * L_mean,          # every other agent's hulls in, lines out
                            "qp": b_guess + 24 * L_mean + b_out}
        per_kernel = {}
        for name, ms_k in (("hull", hull_ms), ("separator", sep_ms), ("qp", qp_ms)):
            gbs = per_kernel_bytes[name] * launch_replans / (ms_k * 1e-3) / 1e9 if ms_k > 0 else 0.0
            per_kernel[name] = {"bytes_per_replan": per_kernel_bytes[name], "ms": ms_k, "GB/s": gbs, "frac": gbs / 8000.0}
        seq_gbs = bytes_per_replan * launch_replans / (seq_ms * 1e-3) / 1e9 if seq_ms > 0 else 0.0
        flops = algorithmic_flops(K8, float(sol["stats"]["n_lines"].mean()), float(hn[:, :K8][hn[:, :K8] > 0].mean()) if (hn[:, :K8] > 0).any() else 4.0, float(iters.mean()))
        fp64_ach = flops * launch_replans / (qp_ms * 1e-3) / 1e12 if qp_ms > 0 else 0.0
        fp64 = {"bound": "fp64 vector (reported next to the HBM roofline, SURVEY 8d)", "achieved": fp64_ach, "peak": 78.6, "unit": "TFLOP/s",
                "frac": fp64_ach / 78.6, "algorithmic_flops_per_replan": flops,
                "note": "an upper bound: SURVEY 8d's count assumes the dense G'WG product; the kernel's structured assembly over 64 base rows executes fewer"}
        if world == 1:
            sharding = "one GPU: all %d agents of every scene" % N
        elif sharded_hulls:
            sharding = ("agents of every scene block-sharded by id, %d per GPU; per step and scene chunk (%d chunks, pipelined) one all-gather "
                        "(RCCL, %s) of the interval hulls of the local agents' committed trajectories (%d B per agent and scene)"
                        % (n_local, C, "native binding on a side stream inside the captured step" if native else "torch.distributed", be.hull_block_bytes() // (Sc * n_local)))
        else:
            sharding = "agents of every scene block-sharded by id, %d per GPU; per step one all-gather (RCCL) of the committed trajectory records" % n_local
        out = {
            "metric": "backend_replans_per_sec", "value": value, "unit": "replans/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d agents + %d static obstacles, K=8, %d seeded scenes in flight per GPU per step (seeds 0..%d over %d GPU%s)"
                                   % (N, M, args.scenes, S - 1, world, "" if world == 1 else "s"),
                       "agents": N, "obstacles": M, "scenes_in_flight": S, "scenes_per_gpu": args.scenes,
                       "replans_per_step": replans_per_step, "replans_per_gpu_per_step": S * n_local,
                       "sharding": sharding,
                       "params": "reference neptune_mtlp_benchmark.yaml (T_span 0.5, num_pol 8, weight 1000, v 2, a 3)"},
            "what_value_is": "throughput of %d INDEPENDENT scenes in flight, every row through the interior point, QP workgroups ordered by the previous "
                             "step's measured times (exact here: the same problems every step).  The representative figures are the closed-loop legs, "
                             "where every step poses new problems from device-made guesses: moving %s replans/s, crossing (the whole fleet through the middle) %s "
                             "replans/s — see also launch_order_off, single_scene, active_rows" % (S, ("%.3g" % moving["value"]) if moving else "n/a", ("%.3g" % crossing["value"]) if crossing else "n/a"),
            "solver": {"status_ok": int((status == 0).sum()), "status_relaxed": int((status == 1).sum()),
                       "status_failed": int((status == 2).sum()), "ipm_iters_mean": float(iters.mean()),
                       "ipm_iters_quantiles": {"p50": float(np.percentile(iters, 50)), "p90": float(np.percentile(iters, 90)),
                                               "p99": float(np.percentile(iters, 99)), "max": int(iters.max())},
                       "ipm_iters_mean_by_status": {name: (float(iters[status == k].mean()) if (status == k).any() else None)
                                                    for k, name in ((0, "ok"), (1, "relaxed"), (2, "failed"))},
                       "lines_mean": float(sol["stats"]["n_lines"].mean()), "lp_failed": int(sol["stats"]["n_lp_failed"].sum()),
                       "rows_solved_mean": float(sol["stats"]["n_rows"].mean()), "line_cull_radius": args.cull_radius,
                       "lp_failed_note": "separator LPs without a separating line: the constraint is skipped as in the reference "
                                         "(solver_gurobi_poly.cpp:483-494).  Round 0 has none (scenes are sampled so that every LP is feasible); "
                                         "later rounds replan the same guesses against the others' optimised trajectories, which may cross them",
                       "active_rows": active},
            # the per-replan solve time the metric asks for: device time of each replan's interior-point workgroup (nep_stats.solve_us,
            # last timed step), and the batch view — every replan of a step completes with its batch
            "solve_us": dict(solve_us, note="device time per replan of the QP workgroup (setup + interior point + outputs); the separator's "
                                            "%.3f ms per launch is shared by the batch" % sep_ms),
            "p50_solve_ms": solve_us["p50"] * 1e-3, "p99_solve_ms": solve_us["p99"] * 1e-3,
            "batch_sequence_ms": seq_ms + (hull_ms if sharded_hulls else 0.0),
            "step_ms": {"p50": float(np.percentile(step_ms, 50)), "p99": float(np.percentile(step_ms, 99)), "max": float(step_ms.max())},
            "kernel_ms": {"hull": hull_ms, "separator": sep_ms, "qp": qp_ms, "sequence": seq_ms, "exchange_wait": mean_ms(gather_ev),
                          "launches": n_launch, "launches_per_step": C},
            "launch": ("one captured HIP graph per step, replayed (per-kernel events from %d eager steps after the timed region)" % min(args.steps, 40)
                       if graph is not None else ("; ".join(graph_notes) or "host launches")),
            "frontend": ({"ms": mean_ms(fe_ev), "beam_width": args.beam,
                          "status_goal_reached": int((d_fe_res.cpu().numpy().view(abi.FE_RESULT_DTYPE)["status"] == 1).sum()),
                          "status_no_solution": int((d_fe_res.cpu().numpy().view(abi.FE_RESULT_DTYPE)["status"] == 3).sum()),
                          "children_mean": float(d_fe_res.cpu().numpy().view(abi.FE_RESULT_DTYPE)["n_children"].mean())}
                         if args.frontend else None),
            "safety": ({"ms": mean_ms(safety_ev), "accepted_frac": float(d_accept.float().mean().item())}
                       if args.safety else None),
            "roofline": {"bound": "hbm", "kernel": be.qp_kernel_name(), "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0,
                         # the committed PMC summary is of the default single-GPU command (8 192 replans per launch): a replay of that file
                         "traffic": measured_traffic("nep::" + be.qp_kernel_name()) if launch_replans == 8192 else None,
                         "traffic_source": "profiles/pmc_summary_latest.txt (committed rocprofv3 --pmc summary of this command; counters cannot be read in-process)",
                         "algorithmic_bytes_per_replan": bytes_per_replan, "replans_per_launch": launch_replans,
                         "sequence": {"achieved": seq_gbs, "frac": seq_gbs / 8000.0, "ms": seq_ms,
                                      "note": "the whole replan's bytes over the whole launch sequence (hull + separator + qp)"},
                         "per_kernel": per_kernel,
                         "note": "achieved = the whole replan's algorithmic bytes (SURVEY 8d) x replans per launch / qp_kernel's duration, as the "
                                 "contract defines it; most of those bytes (other agents' hull vertices) are read by separator_kernel: per_kernel "
                                 "gives each kernel's own bytes over its own time.  Latency-bound path: ~%d dependent interior-point "
                                 "iterations per replan" % round(float(iters.mean()))},
            "long_run": long_run,
            "launch_order_off": order_off,
            "reference_tolerances": ref_tol,
            "presolve": presolve,
            "chain": chain,
            "moving": moving,
            "crossing": crossing,
            "single_scene": single,
            "small_configs": small_configs,
            "per_agent_api": per_agent,
            "config5": config5,
            "per_gpu_value": value / world,
            "per_rank": per_rank,
            "rccl": ({"process_group": "nccl (RCCL), world %d" % world, "initialised": True, "one_rank_all_gather_matches": rccl_one_rank_ok,
                      "nranks": nranks, "exchange": ("native (nep_batch_exchange_hulls: ncclAllGather inside the captured step)" if native else
                                                     ("torch.distributed" if world > 1 else "none (one rank)")),
                      **({"torn_down_before_timing": True} if rccl_torn_down else {})}
                     if ((use_dist or rccl_torn_down) and dist_backend == "nccl")
                     else {"initialised": False, "note": rccl_note or dist_backend}),
            "roofline_fp64": fp64,
            "reference_budget": "reference TimeLimit 0.05 s/solve, replan timer 20 Hz/agent => <= %d replans/s for %d agents" % (20 * N, N),
        }
        if graph_notes:
            out["graph_notes"] = graph_notes
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(p, mine)
            out["reference_solvers"] = out["cpu_baseline"]["reference_solvers"]
        print(json.dumps(out))
    if use_dist:
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
