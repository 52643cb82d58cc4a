"""Which replans make the QP kernel's tail?  Front-end guesses for the bench scenes, one replan, then the
iteration histogram and the slots with the most interior-point iterations (development aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from neptune_amd import abi, dist as ndist, scene
from neptune_amd.backend import BatchBackend

N, M, S = 64, 20, int(sys.argv[1]) if len(sys.argv) > 1 else 32
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 1
scs = [scene.make_scene(N, M, seed=s) for s in range(S)]
p = scs[0]["par"]
com, gue = ndist.stack_scenes(scs)
be = BatchBackend(p, scs[0]["statics"], n_scenes=S)
if os.environ.get("NEP_CULL"):
    be.set_line_cull(float(os.environ["NEP_CULL"]))
d_com = be.to_device(com); d_gue = be.to_device(gue)
cfg = scene.frontend_cfg(p, beam_width=32)
d_start = be.to_device(np.stack([scene.frontend_starts(s) for s in scs]))
d_res = torch.zeros(S * N * abi.FE_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=be.device)
ex = ndist.RoundExchange(S, N, 1, 0, device=be.device)
be.enable_timing(True)
for r in range(rounds):
    be.reset_timing()
    if os.environ.get("NEP_NO_FRONTEND"):
        be.replan(d_com, d_gue)
    else:
        be.frontend(cfg, d_com, d_start, d_gue, d_res)
        be.replan(None, d_gue)
    sol = be.solutions()
    print("round %d: qp kernel %.3f ms" % (r, be.kernel_time_ms(2)[0]))
    st = sol["stats"]
    it = st["iters"].astype(int); status = st["status"].astype(int)
    print("round %d: iters hist" % r, np.bincount(it, minlength=61)[:61].tolist())
    print("   status counts", np.bincount(status, minlength=3).tolist())
    order = np.concatenate([np.argsort(-it)[:6], np.nonzero(status == 1)[0][:6], np.argsort(-st["iters_first"].astype(int))[:4]])
    g = d_gue.cpu().numpy().view(abi.GUESS_DTYPE)
    for s_ in order:
        print("   slot %4d  status %d iters %2d first %2d  K %d lines %3d rows %4d qc %d obj %.6g" % (
            s_, status[s_], it[s_], st["iters_first"][s_], sol["K"][s_], st["n_lines"][s_], st["n_rows"][s_], st["qc_active"][s_], st["objective"][s_]))
    ex.gather(be.d_commit, d_com)
