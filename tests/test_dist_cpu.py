"""CPU, world_size 2, gloo: the multi-GPU round logic (agent sharding + all-gather of committed
trajectory records) without a GPU.  The replan itself is replaced by a deterministic stand-in that
writes a record derived from the snapshot, so the test checks exactly what the exchange must
guarantee: after a round every rank holds the same [S][N] snapshot a single process would hold."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from neptune_amd import abi, dist as ndist, scene


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _fake_replan(committed, first_local, n_local, round_no):
    """Stand-in for the device replan: agent a's new record depends on its old record, on the mean
    of everyone's pos (so a stale snapshot is detected) and on the round."""
    S, N = committed.shape
    out = committed[:, first_local:first_local + n_local].copy()
    mean_pos = committed["pos"].mean(axis=1)            # [S][3]
    for s in range(S):
        for a in range(n_local):
            r = out[s, a]
            r["pos"] = committed[s, first_local + a]["pos"] * 0.5 + mean_pos[s] + round_no
            r["pwp"]["coeff"][:, :, 3] += 0.125 * (first_local + a + 1)
            r["pwp"]["times"] += 0.5
    return out


def _single_process(com0, rounds):
    com = com0.copy()
    S, N = com.shape
    for r in range(rounds):
        com = _fake_replan(com, 0, N, r)
    return com


def _worker(rank, world, port, S, N, rounds, com_bytes, result_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    com = np.frombuffer(com_bytes, dtype=abi.TRAJ_REC_DTYPE).reshape(S, N).copy()
    ex = ndist.RoundExchange(S, N, world, rank, device="cpu")
    committed = torch.from_numpy(com.view(np.uint8).reshape(-1).copy())
    for r in range(rounds):
        snap = committed.numpy().view(abi.TRAJ_REC_DTYPE).reshape(S, N)
        local = _fake_replan(snap, ex.first_local, ex.n_local, r)
        commit_local = torch.from_numpy(np.ascontiguousarray(local).view(np.uint8).reshape(-1).copy())
        ex.gather(commit_local, committed)
    result_q.put((rank, committed.numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_partition():
    assert ndist.shard(64, 8, 3) == (24, 8)
    assert ndist.shard(64, 1, 0) == (0, 64)
    covered = []
    for r in range(4):
        f, n = ndist.shard(256, 4, r)
        covered += list(range(f, f + n))
    assert covered == list(range(256))
    with pytest.raises(ValueError):
        ndist.shard(10, 4, 0)


def test_record_is_the_dyntraj_mirror():
    # id, is_agent, bbox, pos, pwp (times + coeff_x/y/z), bend points: mader_msgs/DynTraj.msg:1-9
    names = abi.TRAJ_REC_DTYPE.names
    for f in ("id", "is_agent", "bbox", "pos", "bend", "pwp"):
        assert f in names
    assert ndist.REC_BYTES == abi.TRAJ_REC_DTYPE.itemsize == 1872


def test_single_rank_gather_is_a_copy():
    sc = scene.make_scene(4, 0, seed=1)
    com = np.stack([sc["committed"], sc["committed"]])
    ex = ndist.RoundExchange(2, 4, 1, 0)
    src = torch.from_numpy(com.view(np.uint8).reshape(-1).copy())
    dst = torch.zeros_like(src)
    ex.gather(src, dst)
    assert torch.equal(src, dst)


@pytest.mark.timeout(300)
def test_two_rank_rounds_match_single_process():
    S, N, rounds, world = 3, 8, 3, 2
    scenes = [scene.make_scene(N, 2, seed=s) for s in range(S)]
    com0 = np.stack([s["committed"] for s in scenes])
    expect = _single_process(com0, rounds)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, S, N, rounds, com0.tobytes(), q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        out = np.frombuffer(got[r], dtype=abi.TRAJ_REC_DTYPE).reshape(S, N)
        assert out.tobytes() == expect.tobytes(), "rank %d snapshot differs from the single-process run" % r


def _hull_worker(rank, world, port, bb, rounds, result_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hx = ndist.HullExchange(bb, world, rank, device="cpu")
    out = []
    for r in range(rounds):
        hx.local.copy_(torch.arange(bb, dtype=torch.int64).mul(rank + 3 + r).remainder(251).to(torch.uint8))
        out.append(hx.gather().numpy().tobytes())
    result_q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_hull_block_exchange():
    """Sharded hulls: after the collective every rank holds every rank's block, in rank (= agent id)
    order, which is the layout nep_batch_replan_hulls addresses."""
    world, bb, rounds = 2, 4096, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hull_worker, args=(r, world, port, bb, rounds, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(rounds):
        want = b"".join((np.arange(bb, dtype=np.int64) * (k + 3 + r) % 251).astype(np.uint8).tobytes() for k in range(world))
        for k in range(world):
            assert got[k][r] == want


def test_single_rank_hull_exchange_is_in_place():
    hx = ndist.HullExchange(1024, 1, 0)
    hx.local.fill_(7)
    assert hx.gather() is hx.blocks and int(hx.blocks.sum()) == 7 * 1024
