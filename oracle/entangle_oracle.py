"""TEST INFRASTRUCTURE ONLY — CPU restatement of the tether entanglement-state propagation
(SURVEY §8f rank 4).  Nothing under neptune_amd/ may import this module.

Parity unpinned: the reference has no test or golden vector for these functions and does not build
here (Eigen, ROS).  The restatement follows the cited lines of neptune/src/entangle_utils.cpp and
neptune/src/kinodynamic_search.cpp; fixed-size Eigen products (P * T) are taken as left-to-right
sums, which the reference's compile flags (unknown) may reassociate.

Plain Python lists shaped like the reference's std::vector members; ids are 1-based as there."""
import copy
import math


class EntState:
    """eu::ent_state (entangle_utils.hpp:23-29)"""

    def __init__(self, n_active):
        self.alphas = []          # [(id, case)]
        self.betas = []
        self.bend = []            # bendPointsIdx
        self.active = [0] * n_active


class Setup:
    """What KinodynamicSearch holds (ctor :95-128, setUp :190-227, setTetherLength, setStaticObstRep)."""

    def __init__(self, num_agents, agent_id, num_pol, num_samples, T_span, cable_length, pb, static_rep, static_longest,
                 sampled, present, bendpts):
        self.N, self.id, self.num_pol, self.ns, self.T, self.cable = num_agents, agent_id, num_pol, num_samples, T_span, cable_length
        self.pb = [tuple(p) for p in pb]
        self.static_rep = static_rep          # [S][2 cols] of (x, y)
        self.static_longest = static_longest  # [S][2]
        self.sampled = sampled                # [N][num_pol][ns+1] of (x, y)
        self.present = present
        self.bendpts = bendpts                # [N] lists of (x, y)


def wedge(a, b, c):                                           # entangle_utils.cpp:16-19
    return (b[0] - a[0]) * (c[1] - a[1]) - (c[0] - a[0]) * (b[1] - a[1])


def wedge2(a, b, c):                                          # :21-28, also returns ab, ac
    ab = (b[0] - a[0], b[1] - a[1])
    ac = (c[0] - a[0], c[1] - a[1])
    return ab[0] * ac[1] - ac[0] * ab[1], ab, ac


def _ratio(u, v):
    if abs(u[1] * v[1]) > abs(u[0] * v[0]):
        return u[1] / v[1]
    return u[0] / v[0]


def hsig_to_add_agent(add, pk, pk1, pik, pik1, pb, bendpts, agent_id):   # :1129-1228
    have_base_addition = False
    n = len(bendpts)
    for i in range(n):
        if i != n - 1:
            c1, pik_pk, pbi_pk = wedge2(pk, bendpts[i + 1], bendpts[i])
            c2 = wedge(pk1, bendpts[i + 1], bendpts[i])
        else:
            c1, pik_pk, pbi_pk = wedge2(pk, pik, bendpts[i])
            c2 = wedge(pk1, pik1, bendpts[i])
        if i == n - 1:
            f1, pik_pb, pbi_pb = wedge2(pb, pik, bendpts[i])
            f2 = wedge(pb, pik1, bendpts[i])
            if f1 * f2 < 0:
                a = _ratio(pik_pb, pbi_pb)
                if a < 0:
                    pass
                elif a < 1:
                    add.append((agent_id, 1))
                elif i == 0:
                    add.append((agent_id, 0))
                have_base_addition = True
        if c1 * c2 < 0:
            a = _ratio(pik_pk, pbi_pk)
            if a < 0:
                add.append((agent_id, i + 2))
            elif a < 1 and i == n - 1:
                add.append((agent_id, 1))
            elif a >= 1 and i == 0:
                add.append((agent_id, 0))
    if have_base_addition:
        if len(add) >= 2 and add[-1] == add[-2]:
            del add[-2:]


def hsig_to_add_static(add, pk, pk1, static_rep, num_agents):  # :1231-1277
    for i, rep in enumerate(static_rep):
        pik, pbi = rep[1], rep[0]
        c1, pik_pk, pbi_pk = wedge2(pk, pik, pbi)
        c2 = wedge(pk1, pik, pbi)
        if c1 * c2 < 0:
            a = _ratio(pik_pk, pbi_pk)
            if a < 0:
                pass
            elif a < 1:
                add.append((num_agents + i + 1, 1))
            else:
                add.append((num_agents + i + 1, 0))


def get_bend_pt(st, pb, pb_self, static_rep, N):               # :1649-1679
    if not st.bend:
        return pb_self
    bid = st.alphas[st.bend[-1]]
    if 1 <= bid[0] <= N:
        return pb[bid[0] - 1]
    return static_rep[bid[0] - N - 1][bid[1]]


def beta_for_case(alpha, pk, pb, bp, static_rep, N):           # :1709-1722
    if alpha[0] <= N:
        return 0.0
    return wedge(pk, static_rep[alpha[0] - N - 1][alpha[1]], bp)


def breakcondition(to_add, in_list, N, idx, idx_last_bend):    # :1608-1647
    if to_add[0] <= N and to_add[1] >= 2:
        if idx <= idx_last_bend:
            return True
    elif to_add[0] <= N and to_add[1] < 2:
        pass
    elif to_add[0] > N:
        if in_list[0] > N or idx <= idx_last_bend:
            return True
    return False


def add_alpha_beta_to_list(add, st, pk, pb, pb_self, static_rep, N, bendpts):   # :1402-1534
    have_cancellation = True
    while have_cancellation:
        _b = st.bend[-1] if st.bend else -1
        restart = False
        i = 0
        while i < len(add) and not restart:
            j = len(st.alphas) - 1
            while j >= 0:
                t, l = add[i], st.alphas[j]
                if (l == t or
                        (t[0] <= N and l[0] == t[0] and t[1] >= len(bendpts[t[0] - 1]) + 1 and t[1] < l[1]) or
                        (t[0] <= N and l[0] == t[0] and l[1] >= 2 and t[1] >= 2 and abs(t[1] - l[1]) == 1 and j > _b)):
                    st.active[t[0] - 1] -= 1
                    del add[i]
                    del st.alphas[j]
                    del st.betas[j]
                    if j == _b:
                        st.bend.pop()
                        bp = get_bend_pt(st, pb, pb_self, static_rep, N)
                        for k in range(j, len(st.alphas)):
                            st.betas[k] = beta_for_case(st.alphas[k], pk, pb, bp, static_rep, N)
                    elif j < _b:
                        st.bend[-1] = _b - 1
                        for k in range(len(st.bend) - 2, -1, -1):
                            if st.bend[k] > j:
                                st.bend[k] -= 1
                            else:
                                break
                    restart = True
                    break
                if breakcondition(t, l, N, j, _b):
                    break
                j -= 1
            i += 1
        have_cancellation = restart
    if not add:
        return
    _pb = get_bend_pt(st, pb, pb_self, static_rep, N)
    for t in add:
        st.alphas.append(t)
        st.active[t[0] - 1] += 1
        st.betas.append(beta_for_case(t, pk, pb, _pb, static_rep, N))


def update_bend_pts(st, pk1, pb, pb_self, static_rep, N):      # :1536-1604
    bp = get_bend_pt(st, pb, pb_self, static_rep, N)
    idx_new_bend = -1
    idx_start = st.bend[-1] if st.bend else -1
    for i in range(idx_start + 1, len(st.alphas)):
        beta = beta_for_case(st.alphas[i], pk1, pb, bp, static_rep, N)
        if beta * st.betas[i] < -1e-7:
            idx_new_bend = i
    if idx_new_bend > -1:
        st.bend.append(idx_new_bend)
        bid = st.alphas[idx_new_bend]
        _bp = pb[bid[0] - 1] if bid[0] <= N else static_rep[bid[0] - N - 1][bid[1]]
        for i in range(idx_new_bend + 1, len(st.alphas)):
            st.betas[i] = beta_for_case(st.alphas[i], pk1, pb, _bp, static_rep, N)
        return
    while st.bend:
        if len(st.bend) == 1:
            bp_prev = pb_self
        else:
            bid = st.alphas[st.bend[-2]]
            bp_prev = pb[bid[0] - 1] if bid[0] <= N else static_rep[bid[0] - N - 1][bid[1]]
        beta = beta_for_case(st.alphas[st.bend[-1]], pk1, pb, bp_prev, static_rep, N)
        if beta * st.betas[st.bend[-1]] > 1e-7:
            for k in range(st.bend[-1] + 1, len(st.alphas)):
                st.betas[k] = beta_for_case(st.alphas[k], pk1, pb, bp_prev, static_rep, N)
            st.bend.pop()
        else:
            break


def tether_length(st, pb, pb_self, pk1, static_rep, static_longest, N):   # :1724-1743 (+ getBendPt2dwIdx :1681-1707)
    length = 0.0
    for idx in st.bend:
        bid = st.alphas[idx]
        if bid[0] <= N:
            bp, comp = pb[bid[0] - 1], 0.0
        else:
            bp, comp = static_rep[bid[0] - N - 1][bid[1]], static_longest[bid[0] - N - 1][bid[1]]
        length += _norm(bp, pb_self) + 2 * comp
        pb_self = bp
    return length + _norm(pk1, pb_self)


def _norm(a, b):
    return math.sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]))


def entangles_with_other_agents(su, st, coeff_x, coeff_y, end, index):
    """KinodynamicSearch::entanglesWithOtherAgents (kinodynamic_search.cpp:707-895).  Mutates st;
    returns (entangled, arc_length)."""
    basepoint = su.pb[su.id - 1]
    pk = (coeff_x[3], coeff_y[3])
    pk1 = pk
    arc = 0.0
    old = list(st.active)
    for j in range(1, su.ns + 1):
        add = []
        if j < su.ns:
            t = su.T * j / su.ns                                # sampled_time_vector_ (:120-125)
            tt = (t * t * t, t * t, t, 1.0)
            pk1 = (((coeff_x[0] * tt[0] + coeff_x[1] * tt[1]) + coeff_x[2] * tt[2]) + coeff_x[3] * tt[3],
                   ((coeff_y[0] * tt[0] + coeff_y[1] * tt[1]) + coeff_y[2] * tt[2]) + coeff_y[3] * tt[3])
        else:
            pk1 = (end[0], end[1])
        arc += _norm(pk1, pk)
        for i in range(su.N):
            if i == su.id - 1:
                continue
            if not su.present[i]:
                continue
            if index > su.num_pol:
                pik = su.sampled[i][su.num_pol - 1][su.ns]
                pik1 = pik
            else:
                pik = su.sampled[i][index - 1][j - 1]
                pik1 = su.sampled[i][index - 1][j]
            hsig_to_add_agent(add, pk, pk1, pik, pik1, basepoint, su.bendpts[i], i + 1)
        hsig_to_add_static(add, pk, pk1, su.static_rep, su.N)
        if len(st.alphas) + len(add) > su.N + len(su.static_rep):
            return True, arc
        add_alpha_beta_to_list(add, st, pk, su.pb, basepoint, su.static_rep, su.N, su.bendpts)
        for i in range(su.N):
            if old[i] < 2 and st.active[i] >= 2:
                return True, arc
            elif old[i] >= 2 and st.active[i] > old[i]:
                return True, arc
        update_bend_pts(st, pk1, su.pb, basepoint, su.static_rep, su.N)
        old = list(st.active)
        pk = pk1
    if tether_length(st, su.pb, basepoint, pk1, su.static_rep, su.static_longest, su.N) > su.cable:
        return True, arc
    return False, arc


def sample_points_of_intervals(times, coeff_x, coeff_y, t_start, t_end, num_pol, ns):
    """Neptune::SamplePointsOfIntervals (neptune.cpp:500-565) -> [num_pol][ns+1] of (x, y)."""
    deltaT = (t_end - t_start) / (1.0 * num_pol)
    n = len(coeff_x)
    out = []
    for i in range(num_pol):
        row = []
        for j in range(ns + 1):
            ts = t_start + deltaT * i + deltaT / ns * j
            low = next((k for k, tk in enumerate(times) if tk > ts), len(times))     # std::upper_bound
            if low != len(times):
                idx = min(max(low - 1, 0), n - 1)
                te = ts - times[idx]
                if te < 0:
                    te = 0
                elif te > deltaT:
                    te = deltaT
            else:
                idx = low - 1 - 1
                te = times[low - 1] - times[low - 2]
            tt = (te * te * te, te * te, te, 1.0)
            row.append((((coeff_x[idx][0] * tt[0] + coeff_x[idx][1] * tt[1]) + coeff_x[idx][2] * tt[2]) + coeff_x[idx][3] * tt[3],
                        ((coeff_y[idx][0] * tt[0] + coeff_y[idx][1] * tt[1]) + coeff_y[idx][2] * tt[2]) + coeff_y[idx][3] * tt[3]))
        out.append(row)
    return out


def propagate_guess(su, init, coeff_x, coeff_y):
    """States at the knots of a K-segment guess: what recoverEntStateVector (:582-603) returns for the
    node chain of that path.  Returns ([EntState] * (K+1), entangled_at)."""
    K = len(coeff_x)
    T = su.T
    st = copy.deepcopy(init)
    states = [copy.deepcopy(st)]
    hit = 0
    for s in range(1, K + 1):
        if not hit:
            cx, cy = coeff_x[s - 1], coeff_y[s - 1]
            if s < K:
                end = (coeff_x[s][3], coeff_y[s][3])
            else:
                end = (((cx[0] * (T * T * T) + cx[1] * (T * T)) + cx[2] * T) + cx[3],
                       ((cy[0] * (T * T * T) + cy[1] * (T * T)) + cy[2] * T) + cy[3])
            nxt = copy.deepcopy(st)
            ent, _ = entangles_with_other_agents(su, nxt, cx, cy, end, s)
            if ent:
                hit = s
            else:
                st = nxt
        states.append(copy.deepcopy(st))
    return states, hit


def case_ids(states, N):
    """solver_gurobi_poly.cpp:620-631 per (knot, agent): the case of the single active crossing."""
    out = [[0] * N for _ in range(8)]
    for i, st in enumerate(states[:8]):
        for j in range(N):
            if st.active[j] != 1:
                continue
            cid = 0
            for a in st.alphas:
                if a[0] == j + 1:
                    cid = a[1]
            out[i][j] = cid
    return out
