"""Which config-5 replans go through the presolve's redo pass, and why (development aid)."""
import os, sys, dataclasses
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from neptune_amd import abi, dist as ndist, scene
from neptune_amd.backend import BatchBackend
from bench import _config5_scene
import multiprocessing as mp
from concurrent.futures import ProcessPoolExecutor


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    with ProcessPoolExecutor(max_workers=S, mp_context=mp.get_context("spawn")) as ex:
        made = list(ex.map(_config5_scene, [(256, 100, s) for s in range(S)]))
    scs = [m[0] for m in made]; case = np.stack([m[1] for m in made])
    p = dataclasses.replace(scs[0]["par"], enable_entangle=True)
    be = BatchBackend(p, scs[0]["statics"], n_scenes=S)
    for s in range(S):
        be.set_scene_statics(s, scs[s]["statics"])
    com, gue = ndist.stack_scenes(scs)
    d_c = be.to_device(com); d_g = be.to_device(gue); d_e = torch.from_numpy(np.ascontiguousarray(case).reshape(-1)).to(be.device)
    T = p.T_span
    M4 = scene.A_POS_INV * np.array([T ** 3, T ** 2, T, 1.0])[:, None]
    for rnd in range(3):
        be.replan(d_c, d_g, d_ent=d_e)
        n = be.redo_count()
        sol = be.solutions(); st = sol["stats"]
        # movement of every replan's control points
        co = np.array(sol["coeff"]); cg = np.array(gue.reshape(-1)["coeff"])
        mv = np.zeros(len(sol))
        for a in range(len(sol)):
            K = int(sol[a]["K"])
            if K < 1 or int(st[a]["status"]) == 2:
                continue
            dx = (co[a, 0, :K] - cg[a, 0, :K]) @ M4; dy = (co[a, 1, :K] - cg[a, 1, :K]) @ M4
            mv[a] = np.sqrt(dx * dx + dy * dy).max()
        full_rows = st["n_rows"] >= 48 * 8 + 4 * (st["n_lines"] - 0)
        print("round %d: redo count %d %s; replans with every row solved %d; control-point movement p50 %.3f p99 %.3f max %.3f m; > 4 m: %d" % (
            rnd, n, be.redo_reasons, int(full_rows.sum()), np.percentile(mv, 50), np.percentile(mv, 99), mv.max(), int((mv > 4.0).sum())))
        for a in np.nonzero(full_rows)[0][:8]:
            print("   slot %d status %d K %d lines %d rows %d iters %d moved %.3f" % (a, st[a]["status"], sol[a]["K"], st[a]["n_lines"], st[a]["n_rows"], st[a]["iters"], mv[a]))
            seg, nd = be.debug_lines(int(a), cap=30000)
            K = int(sol[a]["K"])
            cpx = co[a, 0, :K] @ M4; cpy = co[a, 1, :K] @ M4; gx = cg[a, 0, :K] @ M4; gy = cg[a, 1, :K] @ M4
            val = nd[:, 0:1] * cpx[seg] + nd[:, 1:2] * cpy[seg] + nd[:, 2:3] - 1.0
            gdist = -(nd[:, 0:1] * gx[seg] + nd[:, 1:2] * gy[seg] + nd[:, 2:3] - 1.0) / np.hypot(nd[:, 0], nd[:, 1])[:, None]
            bad = np.nonzero(val.max(axis=1) > -1e-9)[0]
            print("      lines %d; active/violated at the solution: %s; their distance from the guess: %s" % (len(seg), val.max(axis=1)[bad][:6], gdist.min(axis=1)[bad][:6]))
        cm = be.d_commit.view(-1, abi.TRAJ_REC_DTYPE.itemsize); v = d_c.view(-1, abi.TRAJ_REC_DTYPE.itemsize)
        v[:, 40:64].copy_(cm[:, 40:64]); v[:, 192:].copy_(cm[:, 192:])


if __name__ == "__main__":
    main()
