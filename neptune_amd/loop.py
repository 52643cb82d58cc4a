"""A Neptune-like receding-horizon loop for a whole fleet on the batched device path — what
Neptune::replanFull (reference neptune/src/neptune.cpp:1302-1725) does per agent and per timer tick,
run as bulk-synchronous rounds: point A from the committed plan (include/neptune_plan.h) -> front-end
guess (include/neptune_frontend.h) -> separating lines + spline QP (include/neptune_backend.h) ->
post-solve safety check -> plan splice + trajectory composition -> publish.  The reference's goal
follower is its perfect tracker (scripts/perfect_tracker.py: the state is the commanded goal).

Host side only orchestrates: every numeric step is a C-ABI call (device kernels or the host library)."""
import numpy as np

from . import abi, plan, scene
from .backend import BatchBackend


class FleetLoop:
    def __init__(self, par, statics, starts, goals, beam_width=32, delta_t_states=6, replan_every=5, device=None):
        import torch
        self.torch = torch
        self.p, self.statics = par, statics
        self.N = N = par.num_agents
        self.goals = np.asarray(goals, dtype=np.float64).reshape(N, 3)
        self.be = BatchBackend(par, statics, n_scenes=1, device=device)
        self.be.set_safety_check_prev(True)    # nobody commits a trajectory that crosses what somebody else may keep flying
        self.be.set_line_cull(4.0)             # presolve: far separating lines are verified, not solved for (same optimum)
        self.fe = scene.frontend_cfg(par, beam_width=beam_width, pad_hold=1)
        self.dc, self.T = par.dc, par.T_span
        self.k_a = delta_t_states - 1          # index of point A in the plan (neptune.cpp:1376-1385 with deltaT_ states ahead)
        self.replan_every = replan_every       # control ticks between rounds (replan timer / dc)
        self.t = 0.0
        lo = (delta_t_states + 0.5) * par.dc   # pins deltaT_ (mu::saturate truncates its bounds to int)
        self.plans = [plan.CommittedPlan(par.dc, par.T_span, lo, lo, 0.0, 1.0, deltaT0=delta_t_states) for _ in range(N)]
        self.state = np.zeros((N, 12))
        for a in range(N):
            self.state[a, :3] = [starts[a][0], starts[a][1], par.goal_height]
            self.plans[a].reset(self.state[a])
        self.prev_pwp = [None] * N             # composed committed trajectory (pwp_prev_)
        self.done = np.zeros(N, dtype=bool)
        self.trace = None                      # set to a list to record (t, agent, outcome, K) of every replan
        self.stats = dict(rounds=0, replans=0, accepted=0, fe_no_solution=0, qp_failed=0, qp_relaxed=0, rejected_by_safety=0,
                          min_pair_dist=np.inf, min_static_dist=np.inf)
        self._static_pts = [np.asarray(s, dtype=np.float64) for s in scene_raw(statics, par)]

    # ---- records every agent publishes (publishOwnTraj) ----
    def _records(self, t_from):
        rec = np.zeros(self.N, dtype=abi.TRAJ_REC_DTYPE)
        for a in range(self.N):
            r = rec[a]
            r["id"] = a + 1; r["is_agent"] = 1; r["valid"] = 1; r["n_bend"] = 1
            r["bbox"] = 2 * self.p.drone_radius
            r["pos"] = self.state[a, :3]
            r["bend"][0] = self.p.pb[a]
            pw = self.prev_pwp[a]
            if pw is None:                      # not flying yet: a one-interval hover
                r["pwp"]["n_seg"] = 1
                r["pwp"]["times"][:2] = [t_from, t_from + 1000.0]
                r["pwp"]["coeff"][:, 0, 3] = self.state[a, :3]
            else:
                times, coeff = plan.pwp_arrays(pw)
                n = pw.n_seg
                r["pwp"]["n_seg"] = n
                r["pwp"]["times"][: n + 1] = times
                r["pwp"]["coeff"][:, :n, :] = coeff
        return rec

    def _tick(self):
        """one control period: every agent's tracker takes the next goal (Neptune::getNextGoal)"""
        for a in range(self.N):
            g, _last = self.plans[a].next_goal()
            self.state[a] = g
        self.t += self.dc
        xy = self.state[:, :2]
        d = np.sqrt(((xy[:, None, :] - xy[None, :, :]) ** 2).sum(-1)) + np.eye(self.N) * 1e9
        self.stats["min_pair_dist"] = min(self.stats["min_pair_dist"], float(d.min()))
        for pts in self._static_pts:
            c = pts.mean(axis=0); half = (pts.max(axis=0) - pts.min(axis=0)) / 2
            q = np.maximum(np.abs(xy - c) - half, 0.0)
            self.stats["min_static_dist"] = min(self.stats["min_static_dist"], float(np.sqrt((q ** 2).sum(-1)).min()))

    def round(self):
        """one bulk-synchronous replanning round for every agent that is not at its goal"""
        N, p, torch, be = self.N, self.p, self.torch, self.be
        t_now = self.t
        t_start = t_now + (self.k_a + 1) * self.dc      # plan[k] is k+1 control ticks ahead of the tracked state
        starts = np.zeros(N, dtype=abi.FE_START_DTYPE)
        k_end = np.zeros(N, dtype=np.int64)
        for a in range(N):
            pa = self.plans[a].select_a(self.state[a, :3], t_now)
            A = np.array([pa.A[i] for i in range(12)])
            k_end[a] = pa.k_index_end
            starts[a]["pos"] = A[0:3]; starts[a]["vel"] = A[3:6]; starts[a]["accel"] = A[6:9]
            starts[a]["goal"] = self.goals[a]
            starts[a]["t_start"] = t_start      # one clock per round; an agent whose plan is shorter rests at its end
        rec = self._records(t_now)
        d_com = be.to_device(rec); d_start = be.to_device(starts)
        d_guess = torch.zeros(N * abi.GUESS_DTYPE.itemsize, dtype=torch.uint8, device=be.device)
        d_fres = torch.zeros(N * abi.FE_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=be.device)
        be.frontend(self.fe, d_com, d_start, d_guess, d_fres)
        be.replan(None, d_guess)
        d_final = torch.empty_like(d_com); d_acc = torch.zeros(N, dtype=torch.int32, device=be.device)
        be.safety_commit(d_com, be.d_commit, d_guess, d_final, d_acc)
        sol = be.solutions(); states = be.states(); fres = d_fres.cpu().numpy().view(abi.FE_RESULT_DTYPE)
        acc = d_acc.cpu().numpy()
        self.stats["rounds"] += 1
        for a in range(N):
            if self.done[a]:
                continue
            self.stats["replans"] += 1
            K = int(sol[a]["K"]); status = int(sol[a]["stats"]["status"])
            outcome = ("fe_no_solution" if int(fres[a]["status"]) == 3 or K == 0 else
                       "qp_failed" if status == abi.NEP_FAILED else
                       "rejected_by_safety" if not acc[a] else "accepted")
            if self.trace is not None:
                self.trace.append((t_now, a, outcome, K, int(fres[a]["status"]), status))
            self.stats[outcome] += 1
            if outcome != "accepted":
                continue                        # replanFull() false: the agent keeps its plan (neptune.cpp:1519-1545, neptune_ros.cpp:651-663)
            self.stats["qp_relaxed"] += status == abi.NEP_RELAXED
            ns = int(sol[a]["n_states"])
            self.plans[a].splice(int(k_end[a]), states[a, :ns])
            new = plan.make_pwp(np.array(sol[a]["times"])[: K + 1], np.array(sol[a]["coeff"])[:, :K, :])
            if self.prev_pwp[a] is None:
                self.prev_pwp[a] = new
            else:
                # what the others must avoid is the path actually flown: nep_pwp_compose_exact.  (mu::composePieceWisePol,
                # nep_pwp_compose, describes the stretch up to point A with the wrong interval; include/neptune_plan.h)
                self.prev_pwp[a] = plan.compose_exact(t_now, self.prev_pwp[a], new)
        for _ in range(self.replan_every):
            self._tick()
        # arrived (sticky): inside the goal radius and practically at rest; such an agent stops replanning and its
        # committed trajectory keeps it where it is (DroneStatus GOAL_REACHED, neptune.cpp:1701-1711)
        slow = np.sqrt((self.state[:, 3:5] ** 2).sum(axis=1)) < 0.05
        self.done |= (np.hypot(*(self.state[:, :2] - self.goals[:, :2]).T) < p_goal_radius(self.fe)) & slow
        return self.done.all()

    def run(self, max_rounds=400):
        for _ in range(max_rounds):
            if self.round():
                break
        self.stats["sim_time"] = self.t
        self.stats["reached"] = int(self.done.sum())
        self.stats["dist_to_goal_mean"] = float(np.hypot(*(self.state[:, :2] - self.goals[:, :2]).T).mean())
        return self.stats

    def close(self):
        self.be.close()
        for pl in self.plans:
            pl.close()


def p_goal_radius(fe):
    return fe.goal_size


def scene_raw(statics, par):
    """un-inflated footprints of the static obstacles (for the simulation's distance log only)"""
    sd = 2 * par.drone_radius + 0.2
    out = []
    for s in statics:
        v = np.asarray(s, dtype=np.float64)
        c = v.mean(axis=0)
        out.append(c + (v - c) * np.maximum(1.0 - sd / np.maximum(np.abs(v - c), 1e-9), 0.0))
    return out
