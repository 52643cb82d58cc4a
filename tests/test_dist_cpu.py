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


# ---- ShardedRounds (bench.py's multi-GPU step, torch.distributed exchange) with a stand-in back end ------------------------------
class _StubBackend:
    """What ShardedRounds asks of a BatchBackend, on CPU tensors: `hulls` digests my agents' records into my block, `replan_hulls`
    writes new commit records from ALL blocks (so a stale or misplaced block changes the result) — and, like the QP kernel, it writes
    only part of a record: bytes [CARRY_LO, CARRY_HI) of a commit slot come out zero (the tether fields of a config-5 record)."""
    CARRY_LO, CARRY_HI = 96, 160
    FAIL_EVERY = int(os.environ.get("NEP_TEST_STUB_FAIL_EVERY", "0"))            # > 0: the replan of (scene, agent id) with (scene + id + round) % FAIL_EVERY == 0 FAILS: its commit slot keeps what it held

    def __init__(self, n_scenes, n_local, first_local, scene0=0):
        self.torch, self.device = torch, "cpu"
        self.S, self.nl, self.first, self.scene0, self.round = n_scenes, n_local, first_local, scene0, 0
        self.d_commit = torch.zeros(n_scenes * n_local * ndist.REC_BYTES, dtype=torch.uint8)

    def hull_block_bytes(self):
        return self.S * self.nl * 8

    def hulls(self, src, d_guess, out_local):
        rec = src.view(self.S, self.nl, ndist.REC_BYTES).to(torch.int64)
        w = torch.arange(1, ndist.REC_BYTES + 1, dtype=torch.int64)
        dig = (rec * w).sum(dim=2) + d_guess.view(self.S, self.nl).to(torch.int64)          # [S][nl]: every byte of the record counts
        out_local.copy_(dig.contiguous().view(torch.uint8).view(-1))

    def replan_hulls(self, blocks, d_guess, d_ent=None):
        world = blocks.numel() // self.hull_block_bytes()
        allb = blocks.view(torch.int64).view(world, self.S, self.nl)
        tot = allb.sum(dim=(0, 2))                                                         # [S]: everybody's digest of the scene
        old = self.d_commit.view(self.S, self.nl, ndist.REC_BYTES)
        ids = torch.arange(self.first, self.first + self.nl, dtype=torch.int64)
        new = (old.to(torch.int64) * 3 + tot[:, None, None] + ids[None, :, None] * 7 + torch.arange(ndist.REC_BYTES, dtype=torch.int64)[None, None, :]) % 251
        new[:, :, self.CARRY_LO:self.CARRY_HI] = 0
        if self.FAIL_EVERY > 0:              # a failed replan publishes nothing (neptune_ros.cpp:651-663): the slot keeps the record it held, carried bytes included
            sc = torch.arange(self.scene0, self.scene0 + self.S, dtype=torch.int64)[:, None]
            failed = ((sc + ids[None, :] + self.round) % self.FAIL_EVERY) == 0
            new = torch.where(failed[:, :, None], old.to(torch.int64), new)
        self.round += 1
        self.d_commit.copy_(new.to(torch.uint8).view(-1))


def _rounds_reference(local0, guess, S, N, steps):
    """one process, one chunk, every agent local: the result any sharding must reproduce"""
    be = _StubBackend(S, N, 0)
    d_local = [torch.from_numpy(local0.copy()).view(-1)]
    rounds = ndist.ShardedRounds([be], d_local, [torch.from_numpy(guess.copy()).view(-1)], 1, 0, native=False,
                                 carry=((0, _StubBackend.CARRY_LO), (_StubBackend.CARRY_HI, ndist.REC_BYTES)))
    for _ in range(steps):
        rounds.step()
    return d_local[0].view(S, N, ndist.REC_BYTES).numpy().copy()


def _rounds_worker(rank, world, port, S, N, C, steps, local_bytes, guess_bytes, result_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, nl = ndist.shard(N, world, rank)
    local0 = np.frombuffer(local_bytes, dtype=np.uint8).reshape(S, N, ndist.REC_BYTES)
    guess = np.frombuffer(guess_bytes, dtype=np.uint8).reshape(S, N)
    Sc = S // C
    bes = [_StubBackend(Sc, nl, first, scene0=k * Sc) for k in range(C)]
    d_local = [torch.from_numpy(np.ascontiguousarray(local0[k * Sc:(k + 1) * Sc, first:first + nl]).copy()).view(-1) for k in range(C)]
    d_guess = [torch.from_numpy(np.ascontiguousarray(guess[k * Sc:(k + 1) * Sc, first:first + nl]).copy()).view(-1) for k in range(C)]
    rounds = ndist.ShardedRounds(bes, d_local, d_guess, world, rank, native=False,
                                 carry=((0, _StubBackend.CARRY_LO), (_StubBackend.CARRY_HI, ndist.REC_BYTES)))
    for _ in range(steps):
        rounds.step()
    result_q.put((rank, first, nl, [d.numpy().tobytes() for d in d_local]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("chunks", [1, 2])
def test_two_rank_sharded_rounds_with_carried_fields_match_one_rank(chunks):
    """bench.py's N > 1 step (`--workload config5` included: the tethers' fields of a record are carried rank-locally, not written by
    the replan) over gloo with two ranks and one or two scene chunks against one rank with one chunk: the same records, byte for
    byte, after three steps — every block reaches every rank in rank order, a chunk's exchange is never a step stale, and the carried
    byte ranges survive the replan's partial writes."""
    S, N, steps, world = 4, 6, 3, 2
    rng = np.random.default_rng(7)
    local0 = rng.integers(0, 251, size=(S, N, ndist.REC_BYTES), dtype=np.uint8)
    guess = rng.integers(0, 100, size=(S, N), dtype=np.uint8)
    want = _rounds_reference(local0, guess, S, N, steps)
    assert (want[:, :, _StubBackend.CARRY_LO:_StubBackend.CARRY_HI] == local0[:, :, _StubBackend.CARRY_LO:_StubBackend.CARRY_HI]).all()      # (carried, not zeroed)
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_rounds_worker, args=(r, world, port, S, N, chunks, steps, local0.tobytes(), guess.tobytes(), q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    Sc = S // chunks
    for rank, first, nl, parts in got:
        for k, b in enumerate(parts):
            mine = np.frombuffer(b, dtype=np.uint8).reshape(Sc, nl, ndist.REC_BYTES)
            assert mine.tobytes() == np.ascontiguousarray(want[k * Sc:(k + 1) * Sc, first:first + nl]).tobytes(), "rank %d chunk %d differs from the one-rank run" % (rank, k)


def _run_sharded(S, N, C, steps, world, seed):
    rng = np.random.default_rng(seed)
    local0 = rng.integers(0, 251, size=(S, N, ndist.REC_BYTES), dtype=np.uint8)
    guess = rng.integers(0, 100, size=(S, N), dtype=np.uint8)
    want = _rounds_reference(local0, guess, S, N, steps)
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_rounds_worker, args=(r, world, port, S, N, C, steps, local0.tobytes(), guess.tobytes(), q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=400) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    Sc = S // C
    seen = set()
    for rank, first, nl, parts in got:
        seen.add(rank)
        for k, b in enumerate(parts):
            mine = np.frombuffer(b, dtype=np.uint8).reshape(Sc, nl, ndist.REC_BYTES)
            assert mine.tobytes() == np.ascontiguousarray(want[k * Sc:(k + 1) * Sc, first:first + nl]).tobytes(), "rank %d chunk %d differs from the one-rank run" % (rank, k)
    assert seen == set(range(world))
    return want, local0


@pytest.mark.timeout(600)
def test_eight_rank_sharded_rounds_at_the_config4_layout_match_one_rank():
    """BASELINE configs[3]'s layout — 64 agents block-sharded 8 per rank over EIGHT ranks, two scene chunks pipelined — over gloo on CPU
    (the 8-GPU RCCL run itself is the driver's; no such node is reachable from here): after three steps every rank's records equal
    the one-rank, one-chunk run byte for byte.  The exchange replaces neptune_ros.cpp:379-480."""
    assert ndist.shard(64, 8, 5) == (40, 8)
    _run_sharded(S=4, N=64, C=2, steps=3, world=8, seed=11)


@pytest.mark.timeout(300)
def test_sharded_rounds_with_failing_replans_keep_the_previous_records(monkeypatch):
    """Round-5 advisor finding (dist.py, carry mode): a slot whose replan FAILS keeps the record it held — ShardedRounds seeds every
    commit slot from d_local before the first round, so the copy of the non-carried byte ranges back into d_local never overwrites a
    valid record with an unwritten slot.  Stub replans fail for a third of the (scene, agent, round) triples, first round included: two
    ranks, two chunks == one rank, and no record is ever all zeros."""
    monkeypatch.setattr(_StubBackend, "FAIL_EVERY", 3)
    os.environ["NEP_TEST_STUB_FAIL_EVERY"] = "3"
    try:
        want, local0 = _run_sharded(S=4, N=6, C=2, steps=3, world=2, seed=13)
    finally:
        del os.environ["NEP_TEST_STUB_FAIL_EVERY"]
    assert (want.reshape(-1, ndist.REC_BYTES).astype(np.int64).sum(axis=1) > 0).all()
    # a slot that failed in every round still holds its initial record
    ids = np.arange(6)[None, :]; sc = np.arange(4)[:, None]
    always = np.ones((4, 6), dtype=bool)
    for r in range(3):
        always &= ((sc + ids + r) % 3) == 0
    assert not always.any() or (want[always] == local0[always]).all()
