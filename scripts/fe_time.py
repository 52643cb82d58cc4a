"""Development aid: the plain front end (entangle check off) on the bench's headline scenes — kernel time against the sum of the searches'
measured times / 1 024 workgroup slots, i.e. what the launch's tail and its ramp cost.   python scripts/fe_time.py [scenes=128] [rounds=4]"""
import os, sys, heapq
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from neptune_amd import scene, abi
from neptune_amd.backend import BatchBackend


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    N = 64
    made = scene.make_scenes(N, 20, range(S), workers=min(S, 32))
    p = made[0]["par"]
    be = BatchBackend(p, made[0]["statics"], n_scenes=S)
    for s in range(S):
        be.set_scene_statics(s, made[s]["statics"])
    d_c = be.to_device(np.stack([m["committed"] for m in made])); d_s = be.to_device(np.stack([scene.frontend_starts(m) for m in made]))
    d_g = torch.zeros(S * N * abi.GUESS_DTYPE.itemsize, dtype=torch.uint8, device=be.device)
    d_r = torch.zeros(S * N * abi.FE_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=be.device)
    cfg = scene.frontend_cfg(p, beam_width=32, pad_hold=1)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    for r in range(rounds):
        e0.record(); be.frontend(cfg, d_c, d_s, d_g, d_r); e1.record(); torch.cuda.synchronize()
        us = be.fe_search_us(); res = d_r.cpu().numpy().view(abi.FE_RESULT_DTYPE)
        def makespan(order, slots=1024):
            h = [0.0] * slots; heapq.heapify(h); end = 0.0
            for i in order:
                t = heapq.heappop(h) + us[i]; end = max(end, t); heapq.heappush(h, t)
            return end
        print("round %d: frontend of %d searches %.3f ms (with hulls and boxes); search us mean %.0f p50 %.0f p99 %.0f max %.0f; sum / 1024 = %.3f ms; list scheduling of these times: slot order %.3f, perfect LPT %.3f ms; depth mean %.2f" % (
            r, S * N, e0.elapsed_time(e1), us.mean(), np.percentile(us, 50), np.percentile(us, 99), us.max(), us.sum() / 1024e3, makespan(np.arange(len(us))) / 1e3, makespan(np.argsort(-us)) / 1e3, res["depth"].mean()), flush=True)


if __name__ == "__main__":
    main()
