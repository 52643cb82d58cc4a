"""Development aid: the handle's default solve path (verified presolve + polish) against the every-row solve on the same replans — the
scenes' own guesses or front-end guesses — and, for every replan where the two differ by more than 1e-6 or in status, both against the
oracle: which side is off, its iterations, whether the polish pass listed / certified it.
python scripts/presolve_vs_full.py [scenes=16] [own|fe] [seed0=0] [agents=64] [statics=20]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from neptune_amd import scene, abi, dist as ndist
from neptune_amd.backend import BatchBackend


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    kind = sys.argv[2] if len(sys.argv) > 2 else "own"
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    N = int(sys.argv[4]) if len(sys.argv) > 4 else 64
    M = int(sys.argv[5]) if len(sys.argv) > 5 else 20
    scs = scene.make_scenes(N, M, range(seed0, seed0 + S), workers=min(S, 32))
    p = scs[0]["par"]
    if os.environ.get("NEP_AMAX"):
        p.a_max = float(os.environ["NEP_AMAX"])
    com, gue = ndist.stack_scenes(scs)
    bb = BatchBackend(p, scs[0]["statics"], n_scenes=S)
    for s in range(1, S):
        bb.set_scene_statics(s, scs[s]["statics"])
    d_com = bb.to_device(com); d_g = bb.to_device(gue)
    if kind == "fe":
        bb.frontend(scene.frontend_cfg(p, beam_width=32), d_com, bb.to_device(np.stack([scene.frontend_starts(s) for s in scs])), d_g, None)
    g = d_g.cpu().numpy().view(abi.GUESS_DTYPE).reshape(S, N)
    out = {}; flags = {}
    amax = os.environ.get('NEP_AMAX')
    for name, cull, pol in (("full", 0.0, 1), ("default", None, 1), ("default_nopolish", None, 0), ("full_nopolish", 0.0, 0)):
        bb.set_line_cull(4.0 if cull is None else cull); bb.set_polish(pol)
        for _ in range(2):
            bb.replan(d_com, d_g)
        out[name] = bb.solutions().copy(); flags[name] = bb.polish_flags()
        print(name, "cull", bb.line_cull(), "polish listed/certified", bb.polish_count(), "redo", bb.redo_count(),
              "status counts", np.bincount(out[name]["stats"]["status"].astype(int), minlength=3).tolist(), "iters mean %.3f" % out[name]["stats"]["iters"].mean())
    A, B = out["full"], out["default"]
    same = A["stats"]["status"] == B["stats"]["status"]
    ok = same & (A["stats"]["status"] != 2)
    d = np.abs(A["coeff"] - B["coeff"]).reshape(len(A), -1).max(axis=1)
    print("full vs default: status mismatches %d; coeff diff over equal non-failed: max %.3e, > 1e-7: %d, > 1e-6: %d of %d"
          % (int((~same).sum()), d[ok].max(), int((d[ok] > 1e-7).sum()), int((d[ok] > 1e-6).sum()), int(ok.sum())))
    bad = [i for i in range(len(A)) if (not same[i]) or (ok[i] and d[i] > 1e-6)]
    if os.environ.get("NEP_PVF_ORACLE"):
        # every slot of both paths against the oracle (one thread per core: ctypes releases the GIL)
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle
        oracle.lib()
        jobs = [(s_, a_) for s_ in range(S) for a_ in range(N) if int(g[s_, a_]["K"]) > 0]
        with ThreadPoolExecutor(min(128, os.cpu_count() or 1)) as ex:
            refs = list(ex.map(lambda j: oracle.replan(p, j[1] + 1, scs[j[0]]["committed"], g[j[0], j[1]], scs[j[0]]["statics"]), jobs))
        for name in ("full", "default"):
            dd = []; nm = 0
            for (s_, a_), r in zip(jobs, refs):
                so = out[name][s_ * N + a_]; K = int(g[s_, a_]["K"])
                if int(so["stats"]["status"]) != r["status"]:
                    nm += 1; bad.append(s_ * N + a_); continue
                if r["status"] != 2:
                    dd.append(float(np.abs(np.array(so["coeff"])[:, :K, :] - r["coeff"]).max()))
                    if dd[-1] > 1e-6:
                        bad.append(s_ * N + a_)
            dd = np.array(dd)
            print("%s vs oracle: %d replans, status mismatches %d, coeff diff p99 %.2e max %.2e, > 1e-6: %d" % (name, len(jobs), nm, np.percentile(dd, 99), dd.max(), int((dd > 1e-6).sum())))
        bad = sorted(set(bad))
    if bad:
        from oracle import oracle
        for i in bad[:40]:
            s, a = divmod(i, N)
            r = oracle.replan(p, a + 1, scs[s]["committed"], g[s, a], scs[s]["statics"])
            K = int(g[s, a]["K"])
            line = "slot %d (scene %d agent %d) K %d oracle status %d iters %d |" % (i, s, a, K, r["status"], r["iters"])
            for name in ("full", "default", "default_nopolish", "full_nopolish"):
                so = out[name][i]
                dd = float(np.abs(np.array(so["coeff"])[:, :K, :] - r["coeff"]).max()) if r["status"] != 2 else float("nan")
                line += " %s: st %d it %d/%d rows %d polish 0x%x d_oracle %.2e |" % (name, int(so["stats"]["status"]), int(so["stats"]["iters"]), int(so["stats"]["iters_first"]), int(so["stats"]["n_rows"]), int(flags[name][i]), dd)
            print(line)
    bb.close()


if __name__ == "__main__":
    main()
