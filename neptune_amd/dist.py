"""Multi-GPU rounds: agents block-sharded by id across ranks, committed trajectories exchanged by
one all-gather of fixed-size records per round.

The reference runs one OS process per agent and broadcasts each committed trajectory on the
latched ROS topic /trajs (reference neptune/src/neptune_ros.cpp:172,179,434-480, received at
:379-430).  Here a rank owns n_local = N / world agents of every scene; after a round each rank
holds [S][n_local] new records (abi.nep_traj_rec, the DynTraj.msg mirror) and one all-gather over
RCCL/xGMI rebuilds the [S][N] snapshot every rank replans against next (bulk-synchronous Jacobi
rounds instead of the reference's asynchronous last-writer-wins).
"""
import numpy as np

from . import abi

REC_BYTES = abi.TRAJ_REC_DTYPE.itemsize


def shard(num_agents, world, rank):
    """Block partition by agent id: returns (first_local, n_local).  N must divide evenly so that
    the all-gather carries equal-size pieces."""
    if num_agents % world:
        raise ValueError("num_agents (%d) must be a multiple of the world size (%d)" % (num_agents, world))
    n_local = num_agents // world
    return rank * n_local, n_local


class RoundExchange:
    """all-gather of the committed-trajectory records.  Works on any torch device/backend
    (nccl == RCCL on ROCm; gloo in the CPU tests)."""

    def __init__(self, n_scenes, num_agents, world=1, rank=0, group=None, device="cpu"):
        import torch
        self.torch = torch
        self.S, self.N, self.world, self.rank, self.group = n_scenes, num_agents, world, rank, group
        self.first_local, self.n_local = shard(num_agents, world, rank)
        self.piece = n_scenes * self.n_local * REC_BYTES
        self.gathered = torch.empty(world * self.piece, dtype=torch.uint8, device=device) if world > 1 else None

    def gather(self, commit_local, committed_out, collective=None):
        """commit_local: uint8 [S][n_local][REC]; committed_out: uint8 [S][N][REC] (overwritten).  With one rank there is
        nothing to exchange: a device copy, unless collective=True asks for the (degenerate) all-gather all the same."""
        torch = self.torch
        S, N, nl, W = self.S, self.N, self.n_local, self.world
        import torch.distributed as dist
        if W == 1 and not (collective and dist.is_available() and dist.is_initialized()):
            committed_out.copy_(commit_local)
            return committed_out
        if self.gathered is None:   # single rank launched under torch.distributed.run: same collective path
            self.gathered = torch.empty(self.piece, dtype=torch.uint8, device=commit_local.device)
        if commit_local.is_cuda and dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(self.gathered, commit_local, group=self.group)
        elif commit_local.is_cuda:   # gloo with device tensors (single-GPU debugging): through host memory
            mine = commit_local.cpu().contiguous()
            pieces = [torch.empty_like(mine) for _ in range(W)]
            dist.all_gather(pieces, mine, group=self.group)
            self.gathered.copy_(torch.cat(pieces).to(self.gathered.device))
        else:
            pieces = list(self.gathered.view(W, self.piece).unbind(0))
            dist.all_gather(pieces, commit_local.contiguous(), group=self.group)
        # [W][S][n_local][REC] -> [S][W*n_local][REC]
        src = self.gathered.view(W, S, nl * REC_BYTES).permute(1, 0, 2)
        committed_out.view(S, W, nl * REC_BYTES).copy_(src)
        return committed_out


class HullExchange:
    """all-gather of per-rank hull blocks (nep_batch_hulls -> nep_batch_replan_hulls): what the
    separator consumes of the other agents' committed trajectories is their interval hulls, so each
    rank builds the hulls of its own agents only and the blocks travel instead of the records —
    the hull work is sharded with the agents.  The gathered buffer is used as it arrives: the kernels
    address it block by block (rank order = agent-id order), so no permute follows the collective."""

    def __init__(self, block_bytes, world=1, rank=0, group=None, device="cpu"):
        import torch
        self.torch = torch
        self.bb, self.world, self.rank, self.group = block_bytes, world, rank, group
        self.blocks = torch.zeros(world * block_bytes, dtype=torch.uint8, device=device)
        # this rank's block (write the hulls here): the gathered buffer itself when there is nothing
        # to gather, else a separate send buffer (no aliasing between a collective's input and output)
        self.local = self.blocks if world == 1 else torch.zeros(block_bytes, dtype=torch.uint8, device=device)

    class _Done:
        def wait(self):
            return True

    def gather_async(self):
        """Starts the collective and returns a handle whose wait() makes the current stream wait for it
        (RCCL runs on its own stream, so kernels enqueued meanwhile overlap with the exchange)."""
        import torch.distributed as dist
        if self.blocks.is_cuda and self.world > 1 and dist.get_backend(self.group) == "nccl":
            return dist.all_gather_into_tensor(self.blocks, self.local, group=self.group, async_op=True)
        self.gather()
        return HullExchange._Done()

    def gather(self):
        import torch.distributed as dist
        if self.world == 1 and not (dist.is_available() and dist.is_initialized()):
            return self.blocks
        if self.blocks.is_cuda and dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(self.blocks, self.local, group=self.group)
        else:   # gloo (CPU tests, single-GPU debugging): through host memory
            mine = self.local.cpu().contiguous()
            pieces = [self.torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(pieces, mine, group=self.group)
            self.blocks.copy_(self.torch.cat(pieces).to(self.blocks.device))
        return self.blocks


class NativeExchange:
    """The same two exchanges through the C ABI's own RCCL binding (nep_comm_* / nep_batch_exchange_*: what a C++ host
    without PyTorch calls).  The 128-byte communicator id is made by rank 0 and handed to the other ranks through
    `broadcast_id` (any transport: here torch.distributed's broadcast when a process group exists)."""

    def __init__(self, backend, world=1, rank=0, broadcast_id=None):
        import ctypes as C
        from ._lib import lib, check, BackendError
        self.C, self.lib, self.check = C, lib(), check
        self.be, self.world, self.rank = backend, world, rank
        uid = (C.c_uint8 * 128)()
        if rank == 0:
            check(self.lib.nep_comm_unique_id(uid))
        if world > 1:
            if broadcast_id is None:
                import torch
                import torch.distributed as dist
                t = torch.tensor(list(uid), dtype=torch.uint8)
                if dist.get_backend() == "nccl":
                    t = t.to(backend.device)
                dist.broadcast(t, src=0)
                uid = (C.c_uint8 * 128)(*t.cpu().tolist())
            else:
                uid = (C.c_uint8 * 128)(*broadcast_id(bytes(uid)))
        with backend.torch.cuda.device(backend.device):
            self._c = self.lib.nep_comm_create(uid, world, rank)
        if not self._c:
            raise BackendError(self.lib.nep_last_error().decode())

    def close(self):
        if getattr(self, "_c", None):
            self.lib.nep_comm_destroy(self._c)
            self._c = None

    __del__ = close

    def nranks(self):
        """ranks that joined the communicator (ncclCommCount) — nep_comm_nranks"""
        return int(self.check(self.lib.nep_comm_nranks(self._c)))

    def reserve(self, records_bytes=0, slots_bytes=0):
        """sizes the regrouping exchanges' staging buffers ahead of a graph capture (nep_comm_reserve)"""
        self.check(self.lib.nep_comm_reserve(self._c, int(records_bytes), int(slots_bytes)))

    def hulls(self, d_block, d_blocks, stream=None):
        st = stream if stream is not None else self.be.torch.cuda.current_stream(self.be.device)
        self.check(self.lib.nep_batch_exchange_hulls(self.be._h, self._c, d_block.data_ptr(), d_blocks.data_ptr(), st.cuda_stream))

    def slots(self, d_local, d_all, bytes_per_slot, stream=None):
        """any per-slot array [S][n_local][bytes] of every rank -> [S][N][bytes] (nep_batch_exchange_slots)"""
        st = stream if stream is not None else self.be.torch.cuda.current_stream(self.be.device)
        self.check(self.lib.nep_batch_exchange_slots(self.be._h, self._c, d_local.data_ptr(), d_all.data_ptr(), int(bytes_per_slot), st.cuda_stream))

    def records(self, d_commit_local, d_committed_all, stream=None):
        st = stream if stream is not None else self.be.torch.cuda.current_stream(self.be.device)
        self.check(self.lib.nep_batch_exchange_records(self.be._h, self._c, d_commit_local.data_ptr(), d_committed_all.data_ptr(), st.cuda_stream))


class ShardedRounds:
    """The multi-GPU round loop of bench.py (and of a deployment): the scenes are split into `chunks` groups, each with its
    own handle; for a chunk one round is  hulls of MY agents' committed trajectories -> all-gather of the hull blocks ->
    [front end ->] separating lines + QP against the gathered blocks.  One step = every chunk replans once, and the exchange
    of one chunk runs under another chunk's kernels.  Results do not depend on `chunks` (tested against chunks = 1).

    native=True (bench.py's default at N > 1): the all-gather goes through the C ABI's own RCCL binding
    (nep_batch_exchange_hulls) on a side stream that forks from and joins the compute stream inside the step — phase k of a
    step runs  exchange of chunk k+1 (mod C)  beside  replan of chunk k  — so a step is a fixed launch sequence with
    explicit stream dependencies and no host decision in it: it can be captured whole into one HIP graph (RCCL collectives
    are capturable) and replayed without Python between the kernels.
    native=False: torch.distributed's all_gather_into_tensor with async work handles (RCCL's own stream); right after a
    chunk's replan its next hulls are built and their all-gather is started."""

    def __init__(self, backends, d_local, d_guess, world=1, rank=0, group=None, native=False, fe=None, timer=None, d_ent=None, carry=None):
        """backends: one BatchBackend per chunk; d_local[k]: device bytes [Sc][n_local] committed records of my agents in
        chunk k; d_guess[k]: [Sc][n_local] guesses; fe: None or (fe_cfg, d_start[k], d_result[k]); timer: None or a callable
        name -> context manager (bench.py records HIP events around the phases); d_ent: None or per chunk the dense entangle
        case block [Sc][n_local][8][N] (int32) of nep_batch_replan_hulls (enable_entangle_check handles).
        carry: None — the next round's hulls are built from the commit slots as the QP kernel wrote them — or a tuple of byte
        ranges (lo, hi) of a record: after a chunk's replan those ranges of every commit slot are copied into d_local[k] and the
        hulls are built from d_local[k] — what a caller does whose records carry fields the back end does not write (the
        tethers' bend points of a config-5 scene: a commit slot holds the base only)."""
        import contextlib
        self.bes, self.d_local, self.d_guess, self.fe = backends, d_local, d_guess, fe
        self.d_ent, self.carry = d_ent, carry
        self.C = len(backends)
        dev = backends[0].device
        self.torch = backends[0].torch
        self.hx = [HullExchange(b.hull_block_bytes(), world, rank, group=group, device=dev) for b in backends]
        self.native = NativeExchange(backends[0], world, rank) if native else None
        self.pending = [None] * self.C
        self.timer = timer if timer is not None else (lambda name: contextlib.nullcontext())
        # A replan that fails publishes nothing: its d_commit slot keeps what it held (nep_batch_replan_hulls knows no previous
        # records).  The next round's hulls are built from d_commit, so it must hold the agents' committed records from the
        # start — an agent whose very first replan fails keeps flying (and being avoided on) the trajectory it had
        # (neptune_ros.cpp:651-663), instead of vanishing from the others' obstacle sets as a valid = 0 record.
        for b, dl in zip(backends, d_local):
            b.d_commit.copy_(dl.view_as(b.d_commit))
        self.side = self.torch.cuda.Stream(device=dev) if (native and self.torch.cuda.is_available()) else None
        self.primed = False

    def _start(self, k, src):
        b, hx = self.bes[k], self.hx[k]
        with self.timer("hull"):
            b.hulls(src, self.d_guess[k], hx.local)
        if self.native is not None:
            self.native.hulls(hx.local, hx.blocks)      # (in place when there is one rank: sendbuff == recvbuff + rank * count)
            self.pending[k] = HullExchange._Done()
        else:
            self.pending[k] = hx.gather_async()

    def _replan(self, k):
        b = self.bes[k]
        if self.fe is not None:
            cfg, d_start, d_res = self.fe
            with self.timer("frontend"):
                b.frontend_hulls(cfg, self.hx[k].blocks, d_start[k], self.d_guess[k], d_res[k])
        b.replan_hulls(self.hx[k].blocks, self.d_guess[k], d_ent=self.d_ent[k] if self.d_ent is not None else None)
        if self.carry is not None:
            src = b.d_commit.view(-1, REC_BYTES); dst = self.d_local[k].view(-1, REC_BYTES)
            for lo, hi in self.carry:
                dst[:, lo:hi].copy_(src[:, lo:hi])

    def _next_src(self, k):
        """the records chunk k's next hulls are built from"""
        return self.d_local[k] if self.carry is not None else self.bes[k].d_commit

    def prime(self):
        """native mode: the hull blocks chunk 0 replans against in the first step (the other chunks' are exchanged inside it)"""
        if self.native is not None and not self.primed:
            self._start(0, self.d_local[0])
            self.primed = True

    def step(self):
        if self.native is not None:
            torch = self.torch
            self.prime()
            main = torch.cuda.current_stream(self.bes[0].device)
            if self.C == 1:                               # nothing to overlap with: replan, then the exchange for the next step
                self._replan(0)
                self._start(0, self._next_src(0))
                return
            for k in range(self.C):
                j = (k + 1) % self.C
                self.side.wait_stream(main)               # fork: chunk j's commit records (previous step, or this one for j = 0) are complete
                with torch.cuda.stream(self.side):
                    self._start(j, self._next_src(j))
                self._replan(k)                           # reads blocks of chunk k, exchanged one phase ago
                main.wait_stream(self.side)               # join
            return
        for k in range(self.C):
            if self.pending[k] is None:
                self._start(k, self.d_local[k])
        for k in range(self.C):
            with self.timer("wait"):
                self.pending[k].wait()                   # what the stream still has to wait for
            self._replan(k)
            self._start(k, self._next_src(k))            # my agents' new committed trajectories


def stack_scenes(scenes):
    """[scene dicts] -> (committed [S][N], guesses [S][N]) numpy structured arrays."""
    com = np.stack([s["committed"] for s in scenes])
    gue = np.stack([s["guesses"] for s in scenes])
    return com, gue
