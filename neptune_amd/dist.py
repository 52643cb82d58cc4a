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

    def gather(self, commit_local, committed_out):
        """commit_local: uint8 [S][n_local][REC]; committed_out: uint8 [S][N][REC] (overwritten)."""
        torch = self.torch
        S, N, nl, W = self.S, self.N, self.n_local, self.world
        import torch.distributed as dist
        if W == 1 and not (dist.is_available() and dist.is_initialized()):
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


def stack_scenes(scenes):
    """[scene dicts] -> (committed [S][N], guesses [S][N]) numpy structured arrays."""
    com = np.stack([s["committed"] for s in scenes])
    gue = np.stack([s["guesses"] for s in scenes])
    return com, gue
