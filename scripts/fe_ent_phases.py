"""Per-phase shader cycles of frontend_kernel<true> at config-5 size (256 agents + 100 obstacles, entangle check on): build with
make PROFILE=1 (NEP_QP_PROFILE=1 selects that library).  Thread 0's clock per phase, and inside the children pass its time in
the base-square test, the state copy, the propagation, and the per-parent mask fill."""
import os, sys, dataclasses
os.environ["NEP_QP_PROFILE"] = "1"
os.environ.setdefault("NEP_BACKEND_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "neptune_amd", "libneptune_backend_prof.so"))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from neptune_amd import scene, abi
from neptune_amd.backend import BatchBackend


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    scs = [scene.make_scene(256, 100, seed=s) for s in range(S)]
    if os.environ.get("NEP_SCRIPT_BENDS"):      # the bench's config-5 inputs: 2-4 bend points per tether (scene.synthetic_entangle)
        for k_, m_ in enumerate(scs): scene.synthetic_entangle(m_, seed=1000 + k_, frac=0.1)
    p = dataclasses.replace(scs[0]["par"], enable_entangle=True)
    be = BatchBackend(p, scs[0]["statics"], n_scenes=S)
    for s in range(S):
        be.set_scene_statics(s, scs[s]["statics"])
        reps, long_ = scene.static_reps(scs[s]["statics"]); be.set_static_reps(reps, long_, scene=s)
    com = np.stack([s["committed"] for s in scs])
    d_c = be.to_device(com); d_s = be.to_device(np.stack([scene.frontend_starts(s) for s in scs]))
    d_g = torch.zeros(S * 256 * abi.GUESS_DTYPE.itemsize, dtype=torch.uint8, device=be.device)
    d_r = torch.zeros(S * 256 * abi.FE_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=be.device)
    d_case = torch.zeros(S * 256 * abi.NEP_MAX_POL * 256, dtype=torch.int32, device=be.device)
    cfg = scene.frontend_cfg(p, beam_width=32, entangle=True)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    # NEP_SCRIPT_ROUNDS=n: n closed-loop rounds first (search -> replan -> safety pass with the entangle re-check -> commit), so that the
    # searches profiled are the ones the bench's config5.chain leg times in its steady state, not the first replan of a scene
    d_nx = torch.empty_like(d_c); d_ac = torch.zeros(S * 256, dtype=torch.int32, device=be.device)
    for _ in range(int(os.environ.get("NEP_SCRIPT_ROUNDS", "0"))):
        be.frontend_ent(cfg, d_c, d_s, d_g, d_r, d_case); be.replan(None, d_g, d_ent=d_case)
        be.safety_commit_ent(d_c, be.d_commit, d_g, d_nx, d_ac); d_c.copy_(d_nx)
    for _ in range(2):
        e0.record(); be.frontend_ent(cfg, d_c, d_s, d_g, d_r, d_case); e1.record()
    torch.cuda.synchronize()
    print("frontend_ent of %d searches: %.2f ms" % (S * 256, e0.elapsed_time(e1)))
    res = d_r.cpu().numpy().view(abi.FE_RESULT_DTYPE)
    names = ["loop top/barrier", "propagation pass (ENT)", "shortlist + masks", "children pass 2 (GJK)", "voxel dedup", "compact", "rank + install", "children pass 1"]
    rows = np.array([be.debug_phase_cycles(2 * s) + be.debug_phase_cycles(2 * s + 1) for s in range(S * 256)], dtype=np.float64)
    tot = rows[:, :8].sum(axis=1)
    print("mean over %d searches: total %.0f cycles, depths %.2f; children %.0f feasible %.0f" % (len(rows), tot.mean(), rows[:, 8].mean(), res["n_children"].mean(), res["n_feasible"].mean()))
    for k, n in enumerate(names):
        print("   %-24s %10.0f  %5.1f %%" % (n, rows[:, k].mean(), 100.0 * rows[:, k].sum() / tot.sum()))
    for k, n in enumerate(["thread 0: base squares", "thread 0: state copy", "thread 0: propagation", "thread 0: mask fill"]):
        print("   %-24s %10.0f  %5.1f %% of the total" % (n, rows[:, 12 + k].mean(), 100.0 * rows[:, 12 + k].sum() / tot.sum()))
    pn = ["agent loop", "static loop", "merge + counts", "update bends", "tether + rest"]
    npr = rows[:, 21].sum(); tp = rows[:, 16:21].sum()
    print("propagations per search %.0f (%.1f %% of them with a crossing in some step); cycles per propagation %.0f (summed over all threads)" % (rows[:, 21].mean(), 100.0 * rows[:, 22].sum() / max(npr, 1), tp / max(npr, 1)))
    print("   agents visited per propagation (all sampled steps): %.1f; cycles of the agent loop per visit: %.0f" % (rows[:, 31].sum() / max(npr, 1), rows[:, 16].sum() / max(rows[:, 31].sum(), 1)))
    for k, n in enumerate(pn):
        print("   %-24s %10.0f per propagation  %5.1f %%" % (n, rows[:, 16 + k].sum() / max(npr, 1), 100.0 * rows[:, 16 + k].sum() / max(tp, 1)))
    for k, n in enumerate(["merge: ids of the new list", "merge: scans", "merge: erase + re-anchor", "merge: append", "merge: counts"]):
        print("      %-28s %10.0f per propagation" % (n, rows[:, 26 + k].sum() / max(npr, 1)))
    print("propagations whose state differs from the parent's: %.2f %%; crossing-list length at the end of a propagation: mean %.2f, max over the searches: p50 %d p90 %d p99 %d max %d; searches flagged ent_overflow %d"
          % (100.0 * rows[:, 23].sum() / max(npr, 1), rows[:, 25].sum() / max(npr, 1), np.percentile(rows[:, 24], 50), np.percentile(rows[:, 24], 90), np.percentile(rows[:, 24], 99), rows[:, 24].max(), int(res["ent_overflow"].sum())))
    i = int(np.argmax(tot)); print("slowest search: slot %d total %.0f (%.1fx the mean)" % (i, tot[i], tot[i] / tot.mean()), rows[i, :8].astype(int).tolist(), rows[i, 12:].astype(int).tolist())


if __name__ == "__main__":
    main()
