"""What sets the QP kernel's duration in bench.py's `moving` leg (the closed loop on the device)?  Runs the loop for a number
of rounds and prints, for the last round, the workgroups' device times (nep_stats.solve_us) by status / terminal ball row /
iteration count, and the slowest ones (development aid).  Usage: python scripts/moving_tail.py [scenes] [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from neptune_amd import abi, dist as ndist, scene
from neptune_amd.backend import BatchBackend


def main():
    N, M, S = 64, 20, int(sys.argv[1]) if len(sys.argv) > 1 else 128
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    scs = scene.make_scenes(N, M, range(S))
    p = scs[0]["par"]
    com, gue = ndist.stack_scenes(scs)
    be = BatchBackend(p, scs[0]["statics"], n_scenes=S)
    for s in range(S):
        be.set_scene_statics(s, scs[s]["statics"])
    if os.environ.get("NEP_CULL"):
        be.set_line_cull(float(os.environ["NEP_CULL"]))
    cfg = scene.frontend_cfg(p, beam_width=32, pad_hold=int(os.environ.get("NEP_PAD_HOLD", "1")))
    starts = np.stack([scene.frontend_starts(s) for s in scs])
    d_st = be.to_device(starts); d_alt = torch.from_numpy(np.ascontiguousarray(starts["pos"].reshape(S * N, 3)).copy()).to(be.device)
    d_com = be.to_device(com); d_nxt = torch.empty_like(d_com); d_acc = torch.zeros(S * N, dtype=torch.int32, device=be.device)
    d_g = torch.zeros(S * N * abi.GUESS_DTYPE.itemsize, dtype=torch.uint8, device=be.device)
    d_res = torch.zeros(S * N * abi.FE_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=be.device)
    be.enable_timing(True)
    for r in range(rounds):
        be.reset_timing()
        be.frontend(cfg, d_com, d_st, d_g, d_res)
        be.replan(None, d_g)
        be.safety_commit(d_com, be.d_commit, d_g, d_nxt, d_acc)
        d_com.copy_(d_nxt)
        if r + 1 < rounds:
            be.next_starts(d_com, p.T_span, d_st, d_alt, 0.5)
    sol = be.solutions(timing=True); st = sol["stats"]; res = d_res.cpu().numpy().view(abi.FE_RESULT_DTYPE)
    us = st["solve_us"]; it = st["iters"].astype(int); itf = st["iters_first"].astype(int); status = st["status"].astype(int); qc = st["qc_active"].astype(int)
    print("round %d: qp kernel %.3f ms, separator %.3f ms; sum of workgroup times / 1024 = %.3f ms" % (rounds - 1, be.kernel_time_ms(2)[0], be.kernel_time_ms(1)[0], us.sum() / 1024 * 1e-3))
    print("solve_us: p50 %.0f p90 %.0f p99 %.0f max %.0f" % tuple(np.percentile(us, q) for q in (50, 90, 99, 100)))
    for name, m in (("ok, no ball row", (status == 0) & (qc == 0)), ("ok, ball row", (status == 0) & (qc == 1)), ("relaxed", status == 1), ("failed, K>0", (status == 2) & (sol["K"] > 0)), ("K = 0", sol["K"] == 0)):
        if m.any():
            print("  %-16s n %5d  us mean %7.1f p99 %7.1f max %7.1f  iters mean %5.2f  first-solve iters mean %5.2f max %d" % (name, m.sum(), us[m].mean(), np.percentile(us[m], 99), us[m].max(), it[m].mean(), itf[m].mean(), itf[m].max()))
    print("  K histogram", np.bincount(sol["K"].astype(int), minlength=9).tolist(), " front-end status", np.bincount(res["status"].astype(int), minlength=4).tolist())
    print("  share of the launch's workgroup time: ball rows %.1f %%, relaxed + failed %.1f %%" % (100 * us[(qc == 1)].sum() / us.sum(), 100 * us[status > 0].sum() / us.sum()))
    for s_ in np.argsort(-us)[:12]:
        print("   slot %5d  %7.1f us  status %d iters %2d first %2d  K %d lines %3d rows %4d qc %d  fe status %d feK %d dist %.2f" % (s_, us[s_], status[s_], it[s_], itf[s_], sol["K"][s_], st["n_lines"][s_], st["n_rows"][s_], qc[s_], res["status"][s_], res["K"][s_], res["dist_to_goal"][s_]))
    slow = us > 400
    print("  workgroups above 400 us: %d; by front-end status %s; by front-end K %s" % (slow.sum(), np.bincount(res["status"][slow].astype(int), minlength=4).tolist(), np.bincount(res["K"][slow].astype(int), minlength=9).tolist()))
    print("  all: front-end K histogram %s" % np.bincount(res["K"].astype(int), minlength=9).tolist())
    g = d_g.cpu().numpy().view(abi.GUESS_DTYPE); co = np.array(g["coeff"])
    T = p.T_span
    v_end = np.hypot(3 * co[:, 0, 7, 0] * T * T + 2 * co[:, 0, 7, 1] * T + co[:, 0, 7, 2], 3 * co[:, 1, 7, 0] * T * T + 2 * co[:, 1, 7, 1] * T + co[:, 1, 7, 2])
    v0 = np.hypot(co[:, 0, 0, 2], co[:, 1, 0, 2]); a0 = 2 * np.hypot(co[:, 0, 0, 1], co[:, 1, 0, 1])
    for nm, arr in (("end speed of the guess", v_end), ("start speed", v0), ("start accel", a0)):
        print("  %-24s slow: mean %.2f p10 %.2f p90 %.2f | all: mean %.2f p10 %.2f p90 %.2f" % (nm, arr[slow].mean(), np.percentile(arr[slow], 10), np.percentile(arr[slow], 90), arr.mean(), np.percentile(arr, 10), np.percentile(arr, 90)))



if __name__ == "__main__":
    main()
