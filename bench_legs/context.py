"""Process group, barriers, HIP-graph capture and THE timed region of every leg (Ctx.run_leg): warm-up steps, a barrier +
synchronize, exactly `steps` steps, a barrier + synchronize, the maximum over the ranks."""
import os
import sys
import time

import numpy as np


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


class _timed:
    def __init__(self, ctx, lst): self.ctx, self.lst = ctx, lst
    def __enter__(self): self.e0 = self.ctx.ev()
    def __exit__(self, *a): self.lst.append((self.e0, self.ctx.ev()))


class Ctx:
    """One rank of the bench: device, process group and the timing primitives every leg uses."""

    stdout_to_stderr = _stdout_to_stderr

    def __init__(self, args):
        import torch
        import torch.distributed as tdist
        self.args, self.torch, self.tdist = args, torch, tdist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if args.gpus > 1 and self.world != args.gpus:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the back end has no CPU path")
        # development aid: several ranks on ONE GPU over gloo (the driver's runs use one GPU per rank over RCCL)
        self.one_device = os.environ.get("NEP_BENCH_ONE_DEVICE") == "1"
        if self.world > 1 and not self.one_device and torch.cuda.device_count() < self.world:
            raise SystemExit("--gpus %d: this box shows %d GPU(s) (development aid: NEP_BENCH_ONE_DEVICE=1 runs the ranks on one device over gloo)"
                             % (self.world, torch.cuda.device_count()))
        self.dist_backend = os.environ.get("NEP_BENCH_BACKEND", "nccl")
        if self.one_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        self.use_dist = self.world > 1 or "RANK" in os.environ          # launched by torch.distributed.run
        self.own_group = False; self.rccl_torn_down = False              # (plain `python bench.py`: the one-rank group made below)
        self.rccl_note = None
        self.host_cores = os.cpu_count() or 1
        self.aux_steps = max(args.aux_steps, args.steps)
        self.ev_on = True
        self.graph_notes = []
        self.last_wall = 0.0

    # ---- process group -------------------------------------------------------------------------------------------------
    def _init_group(self, backend, **kw):
        torch, tdist = self.torch, self.tdist
        with _stdout_to_stderr():
            if backend == "nccl":
                tdist.init_process_group("nccl", device_id=self.dev, **kw)
                t = torch.ones(1, device=self.dev)
                tdist.all_reduce(t)                              # brings the communicator up now (and its banner with it)
                torch.cuda.synchronize(self.dev)
            else:
                tdist.init_process_group(backend, **kw)

    def init_process_group(self, want_one_rank_group=True):
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29512")
            self._init_group(self.dist_backend)
        elif want_one_rank_group:
            # plain `python bench.py`: a one-rank RCCL process group, so that the single-GPU record also shows the collective
            # library initialising on the box and the round's all-gather call path running (degenerate: one rank)
            try:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
                self._init_group("nccl", rank=0, world_size=1)
                self.use_dist = True; self.own_group = True
            except Exception as e:                                 # never lose the measurement to the extra
                self.rccl_note = "one-rank process group not created: %r" % (e,)

    def teardown(self):
        if self.use_dist:
            with _stdout_to_stderr():                              # (nothing but the JSON line on stdout)
                self.tdist.barrier()
                self.tdist.destroy_process_group()
            self.use_dist = False

    def barrier(self):
        self.torch.cuda.synchronize(self.dev)
        if self.use_dist:
            self.tdist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, dt):
        if self.use_dist:
            t = self.torch.tensor([dt], dtype=self.torch.float64, device=self.dev if self.dist_backend == "nccl" else "cpu")
            self.tdist.all_reduce(t, op=self.tdist.ReduceOp.MAX)
            return float(t.item())
        return dt

    def agree(self, ok):
        """True iff every rank says so (one all-reduce): the ranks must take the same branch"""
        if not self.use_dist or self.world == 1:
            return bool(ok)
        t = self.torch.tensor([1 if ok else 0], dtype=self.torch.int32, device=self.dev if self.dist_backend == "nccl" else "cpu")
        self.tdist.all_reduce(t, op=self.tdist.ReduceOp.MIN)
        return bool(t.item())

    def rounds_with_fallback(self, build, native):
        """build(native) -> dist.ShardedRounds on FRESH device buffers.  With native=True the exchange is the C ABI's own RCCL binding
        (dlopen'ed librccl, its own communicator, captured into the step's graph) — which no box this was developed on could run
        with more than one rank.  So the first step is made here, eagerly, under a guard: if the communicator does not come up or
        the first all-gather fails on ANY rank, every rank falls back to torch.distributed's collective (host-launched steps) and
        the line says so, instead of the run dying.  -> (rounds, native in force, steps already made, note or None)"""
        note = None
        if native:
            rounds, ok = None, True
            try:
                with _stdout_to_stderr():
                    rounds = build(True)
                rounds.step()
                self.torch.cuda.synchronize(self.dev)
            except Exception as e:                      # noqa: BLE001 — whatever went wrong, the fallback is the same
                ok = False; note = "native RCCL exchange unusable on this box (%r): torch.distributed exchange instead" % (e,)
            if self.agree(ok):
                return rounds, True, 1, None
            if rounds is not None and rounds.native is not None:
                try:
                    rounds.native.close()
                except Exception:
                    pass
            note = note or "native RCCL exchange failed on another rank: torch.distributed exchange instead"
            print("[bench] rank %d: %s" % (self.rank, note), file=sys.stderr)
        with _stdout_to_stderr():
            rounds = build(False)
        return rounds, False, 0, note

    def share(self, arr, S):
        """[scenes per GPU][...] of every rank -> [S][...] on every rank (scene generation is spread over the ranks)"""
        torch, tdist = self.torch, self.tdist
        if self.world == 1:
            return arr
        t = torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1).copy())
        if self.dist_backend == "nccl":
            t = t.to(self.dev)
            o_ = torch.empty(self.world * t.numel(), dtype=torch.uint8, device=self.dev)
            tdist.all_gather_into_tensor(o_, t)
            o_ = o_.cpu()
        else:
            pieces = [torch.empty_like(t) for _ in range(self.world)]
            tdist.all_gather(pieces, t)
            o_ = torch.cat(pieces)
        return o_.numpy().view(arr.dtype).reshape((S,) + arr.shape[1:])

    # ---- HIP events on the current stream ------------------------------------------------------------------------------
    def ev(self):
        if not self.ev_on:               # (while a step is being captured into a graph: timing events cannot live inside one)
            return None
        e = self.torch.cuda.Event(enable_timing=True); e.record(); return e

    def timed(self, lst):
        return _timed(self, lst)

    @staticmethod
    def mean_ms(pairs):
        pairs = [(a, b) for a, b in pairs if a is not None and b is not None]     # (pairs "recorded" while a graph was being captured are placeholders)
        return float(np.mean([a.elapsed_time(b) for a, b in pairs])) if pairs else 0.0

    # ---- capture + the timed region ------------------------------------------------------------------------------------
    def capture(self, fn, handles, allow=True):
        """fn() enqueues one step on the current stream -> a captured graph of it, or None (then the host launches).
        One step = a fixed sequence of launches on fixed buffers (at N > 1 including the RCCL all-gathers on their side
        stream): captured once, replayed in the timed region — no per-launch host work, no host jitter between kernels."""
        torch = self.torch
        if self.args.no_graph or not allow:
            return None
        try:
            for b in handles:
                b.enable_timing(False)
            torch.cuda.synchronize(self.dev)
            g_ = torch.cuda.CUDAGraph()
            self.ev_on = False
            with torch.cuda.graph(g_):
                fn()
            self.ev_on = True
            g_.replay(); g_.replay()
            torch.cuda.synchronize(self.dev)
            return g_
        except Exception as e:                       # (falls back to launching from the host)
            self.ev_on = True
            self.graph_notes.append("graph capture failed: %r" % (e,))
            torch.cuda.synchronize(self.dev)
            return None

    def run_leg(self, step_fn, handles, steps, warm, graph_ok=True, eager_after=40, clear=()):
        """warm untimed steps, then exactly `steps` steps between barriers (max over ranks), replaying one captured graph
        when possible.  Per-kernel HIP events (handle timing) cannot live inside a graph: with a graph they are taken from
        `eager_after` host-launched steps after the timed region.  -> (seconds, per-step GPU ms, graph used)"""
        for _ in range(warm):
            step_fn()
        self.barrier()
        g_ = self.capture(step_fn, handles, graph_ok)
        for b in handles:
            b.enable_timing(g_ is None); b.reset_timing()
        for lst in clear:
            lst.clear()
        self.barrier()
        evs = [self.ev()]
        t0 = time.perf_counter()
        for _ in range(steps):
            if g_ is not None:
                g_.replay()
            else:
                step_fn()
            evs.append(self.ev())
        self.barrier()
        self.last_wall = time.perf_counter() - t0
        dt = self.max_over_ranks(self.last_wall)
        step_ms = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(steps)])
        if g_ is not None and eager_after > 0:
            for b in handles:
                b.enable_timing(True); b.reset_timing()
            for lst in clear:
                lst.clear()
            for _ in range(min(steps, eager_after)):
                step_fn()
            self.barrier()
        return dt, step_ms, g_
