"""Parity at scale: every replan of S bench scenes (front-end guesses by default) on the device against the CPU oracle,
one oracle process per host core.  Prints the distribution of the coefficient and cost differences."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from concurrent.futures import ProcessPoolExecutor

N, M = int(os.environ.get("NEP_AGENTS", 64)), int(os.environ.get("NEP_STATICS", 20))
SEED0 = int(os.environ.get("NEP_SEED0", 0))
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
use_fe = not os.environ.get("NEP_NO_FRONTEND")


def tighten(p):
    """NEP_VMAX / NEP_AMAX: tighter bounds than the scenes were made for (many active rows, relaxed and failed solves)"""
    if os.environ.get("NEP_VMAX"):
        p.v_max = float(os.environ["NEP_VMAX"])
    if os.environ.get("NEP_AMAX"):
        p.a_max = float(os.environ["NEP_AMAX"])
    return p


def oracle_one(job):
    seed, a, g = job
    from neptune_amd import scene
    from oracle import oracle
    sc = scene.make_scene(N, M, seed=SEED0 + seed)
    tighten(sc["par"])
    if os.environ.get("NEP_TOL"):           # experiment: the strict tests' tolerances "residual,gap" on both sides
        oracle.set_qp_tolerances(*[float(x) for x in os.environ["NEP_TOL"].split(",")])
    r = oracle.replan(sc["par"], a + 1, sc["committed"], g, sc["statics"])          # every scene has its own statics, on both sides
    strict = None
    if os.environ.get("NEP_SWEEP_STRICT"):
        # the same oracle pushed to its limit (gap tolerance 1e-13 and two refinement steps per Newton solve; read per solve
        # by the C library): tells which of the two sides a difference belongs to
        os.environ["ORC_EXP_GAPTOL"] = "1e-13"; os.environ["ORC_EXP_REFINE"] = "2"
        r2 = oracle.replan(sc["par"], a + 1, sc["committed"], g, sc["statics"])
        del os.environ["ORC_EXP_GAPTOL"], os.environ["ORC_EXP_REFINE"]
        strict = (r2["status"], r2["iters"], np.array(r2["coeff"]))
    return seed, a, r["status"], r["iters"], r["objective"], np.array(r["coeff"]), strict


def main():
    import torch
    from neptune_amd import abi, dist as ndist, scene
    from neptune_amd.backend import BatchBackend
    scs = [scene.make_scene(N, M, seed=SEED0 + s) for s in range(S)]
    p = tighten(scs[0]["par"])
    com, gue = ndist.stack_scenes(scs)
    be = BatchBackend(p, scs[0]["statics"], n_scenes=S)
    for s_ in range(1, S):
        be.set_scene_statics(s_, scs[s_]["statics"])
    d_com = be.to_device(com); d_gue = be.to_device(gue)
    if os.environ.get("NEP_TOL"):
        be.set_tolerances(*[float(x) for x in os.environ["NEP_TOL"].split(",")])
    if os.environ.get("NEP_CULL") is not None:
        be.set_line_cull(float(os.environ["NEP_CULL"]))
    if use_fe:
        d_start = be.to_device(np.stack([scene.frontend_starts(s) for s in scs]))
        be.frontend(scene.frontend_cfg(p, beam_width=32), d_com, d_start, d_gue, None)
        be.replan(None, d_gue)
    else:
        be.replan(d_com, d_gue)
    sol = be.solutions()
    g = d_gue.cpu().numpy().view(abi.GUESS_DTYPE).reshape(S, N)
    jobs = [(s, a, g[s, a].copy()) for s in range(S) for a in range(N) if int(g[s, a]["K"]) > 0]
    import multiprocessing as mp
    with ProcessPoolExecutor(max_workers=min(os.cpu_count() or 1, 128), mp_context=mp.get_context("spawn")) as ex:
        res = list(ex.map(oracle_one, jobs, chunksize=4))
    dco, dob, dpos, st_bad, worst = [], [], [], 0, None
    d_gs, d_os, tail = [], [], []
    for seed, a, status, iters, obj, coeff, strict in res:
        so = sol[seed * N + a]
        if int(so["stats"]["status"]) != status:
            st_bad += 1
            continue
        if status == 2:
            continue
        K = coeff.shape[1]
        d = float(np.abs(np.array(so["coeff"])[:, :K, :] - coeff).max())
        dco.append(d); dob.append(abs(float(so["stats"]["objective"]) - obj) / (1 + abs(obj)))
        dc = np.array(so["coeff"])[:, :K, :] - coeff                      # position difference along the segments (metres)
        dpos.append(max(float(np.abs(((dc[..., 0] * t + dc[..., 1]) * t + dc[..., 2]) * t + dc[..., 3]).max()) for t in (0.0, 0.125, 0.25, 0.375, 0.5)))
        if worst is None or d > worst[0]:
            worst = (d, seed, a, int(so["stats"]["iters"]), iters)
        if strict is not None and strict[0] == status:
            d_gs.append(float(np.abs(np.array(so["coeff"])[:, :K, :] - strict[2]).max())); d_os.append(float(np.abs(coeff - strict[2]).max()))
            if d > 1e-6:
                tail.append((seed, a, int(so["stats"]["status"]), int(so["stats"]["iters"]), iters, strict[1], d, d_gs[-1], d_os[-1]))
    dco = np.array(dco); dob = np.array(dob); dpos = np.array(dpos)
    if len(dco) == 0:
        print("replans compared 0 (status mismatches %d): nothing solved on either side" % st_bad)
        return
    print("statuses on the device: ok %d relaxed %d failed %d; iterations mean %.3f (line presolve radius %g)" % (tuple(np.bincount(sol["stats"]["status"].astype(int), minlength=3)[:3]) + (float(sol["stats"]["iters"].mean()), be.line_cull())))
    print("replans compared %d (status mismatches %d) | coeff diff: p50 %.2e p99 %.2e max %.2e, > 1e-6: %d, > 1e-5: %d | position diff along the trajectories: p99 %.2e max %.2e m | rel cost diff max %.2e | worst (diff, scene, agent, gpu iters, oracle iters) %s"
          % (len(dco), st_bad, np.percentile(dco, 50), np.percentile(dco, 99), dco.max(), int((dco > 1e-6).sum()), int((dco > 1e-5).sum()), np.percentile(dpos, 99), dpos.max(), dob.max(), worst))
    if d_gs:
        d_gs = np.array(d_gs); d_os = np.array(d_os)
        print("against the oracle at its limit (gap tolerance 1e-13, two refinement steps per Newton solve), %d replans: device p99 %.2e max %.2e | oracle (as shipped) p99 %.2e max %.2e"
              % (len(d_gs), np.percentile(d_gs, 99), d_gs.max(), np.percentile(d_os, 99), d_os.max()))
        for t in tail:
            print("   tail: scene %d agent %d status %d iterations device %d oracle %d strict %d | device-oracle %.2e device-strict %.2e oracle-strict %.2e" % t)


if __name__ == "__main__":
    main()
