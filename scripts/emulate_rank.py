"""One rank's share of a W-GPU bench step on ONE GPU (no collective): what bench.py --gpus W runs per rank, with the other
ranks' hull blocks made once before the timed loop.  S = scenes_per_gpu * W scenes, N / W local agents per scene, the scenes
split into `chunks` handles as dist.ShardedRounds does.  Prints per-kernel times of rank 0.
Usage: python scripts/emulate_rank.py W [scenes_per_gpu] [chunks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np


def main():
    import torch
    from neptune_amd import scene, dist as ndist
    from neptune_amd.backend import BatchBackend
    W = int(sys.argv[1]); spg = int(sys.argv[2]) if len(sys.argv) > 2 else 128; C = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    N, M = 64, 20
    # the scenes repeat a pool of 32 (made once): the timing does not depend on which seeds fill the batch
    pool = scene.make_scenes(N, M, range(32))
    S = spg * W; Sc = S // C; nl = N // W
    p = pool[0]["par"]
    steps = 20
    tot = {"hull": 0.0, "sep": 0.0, "qp": 0.0}; wall = 0.0
    handles = []
    for k in range(C):
        scs = [pool[(k * Sc + s) % 32] for s in range(Sc)]
        com, gue = ndist.stack_scenes(scs)
        ranks = []
        h0 = BatchBackend(p, scs[0]["statics"], first_local=0, n_local=nl, n_scenes=Sc)
        for s in range(1, Sc): h0.set_scene_statics(s, scs[s]["statics"])
        bb = h0.hull_block_bytes()
        blocks = torch.zeros(W * bb, dtype=torch.uint8, device=h0.device)
        # the other ranks' blocks: made by one more handle per rank, then dropped
        for r in range(1, W):
            hr = BatchBackend(p, scs[0]["statics"], first_local=r * nl, n_local=nl, n_scenes=Sc)
            hr.hulls(hr.to_device(np.ascontiguousarray(com[:, r * nl:(r + 1) * nl])), hr.to_device(np.ascontiguousarray(gue[:, r * nl:(r + 1) * nl])), blocks[r * bb:(r + 1) * bb])
            torch.cuda.synchronize(); hr.close()
        d_com = h0.to_device(np.ascontiguousarray(com[:, :nl])); d_gue = h0.to_device(np.ascontiguousarray(gue[:, :nl]))
        h0.enable_timing(True)
        handles.append((h0, d_com, d_gue, blocks, bb))
    for it in range(steps + 3):
        if it == 3:
            torch.cuda.synchronize(); t0 = time.time()
        for (h0, d_com, d_gue, blocks, bb) in handles:
            h0.hulls(d_com, d_gue, blocks[:bb])
            h0.replan_hulls(blocks, d_gue)
    torch.cuda.synchronize(); wall = (time.time() - t0) / steps
    for (h0, *_r) in handles:
        for k, w in (("hull", 0), ("sep", 1), ("qp", 2)):
            try:
                ms, n = h0.kernel_time_ms(w)
            except Exception:
                ms = float("nan")
            tot[k] += ms
    rp = S * nl
    print("W=%d: %d scenes x %d local agents = %d replans per rank and step in %d chunk(s): wall %.3f ms per step (host launches) -> %.2f M replans/s per GPU | kernels per step: %s"
          % (W, S, nl, rp, C, wall * 1e3, rp / wall / 1e6, {k: round(v, 4) for k, v in tot.items()}))


if __name__ == "__main__":
    main()
