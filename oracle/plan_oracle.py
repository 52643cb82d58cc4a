"""TEST INFRASTRUCTURE ONLY — CPU restatement of the host logic around the back end.
Nothing under neptune_amd/ may import this module.

Parity unpinned: the reference holds no test or golden vector for these functions and neither ROS
(for the wire format) nor the reference itself builds in this image; the restatement follows the
cited lines, and the wire format follows the published ROS1 serialisation rules (little endian,
uint32 length prefixes for strings and variable arrays, bool = 1 byte, time = 2×uint32) applied to
mader_msgs/msg/{DynTraj,PieceWisePolTraj,CoeffPoly3}.msg.

Plain Python lists, shaped like the reference's std::vector / std::deque members."""
import math
import struct


class Pwp:
    """mt::PieceWisePol (mader_types.hpp:462-548): times[n+1], coeff_{x,y,z}[n] of [a,b,c,d]."""

    def __init__(self, times=(), cx=(), cy=(), cz=()):
        self.times = [float(t) for t in times]
        self.cx = [list(map(float, c)) for c in cx]
        self.cy = [list(map(float, c)) for c in cy]
        self.cz = [list(map(float, c)) for c in cz]

    def copy(self):
        return Pwp(self.times, self.cx, self.cy, self.cz)


def compose_piecewise_pol(t, dc, p1, p2):
    """mu::composePieceWisePol (utils.cpp:318-402).  p1/p2 are modified in place, as there."""
    if p1.times[-1] < t < p2.times[0]:                       # :321-325
        p2.times[0] = t
    if p1.times[-1] < p2.times[0]:                           # :327-330
        p2.times[0] = p1.times[-1]
    if t < p1.times[0]:                                      # :332-335
        p1.times[0] = t
    if abs(t - p2.times[0]) < 1e-5:                          # :337-340
        return p2.copy()
    if p1.times[-1] < p2.times[0] or t > p2.times[-1] or t < p1.times[0]:   # :342-355
        return Pwp()
    idx1 = [i for i, ti in enumerate(p1.times) if t < ti < p2.times[0]]      # :359-365
    idx2 = [i for i, ti in enumerate(p2.times) if ti > t]                    # :367-373
    p = Pwp()
    p.times.append(t)                                        # :376
    for i in idx1:                                           # :378-384
        p.times.append(p1.times[i])
        p.cx.append(list(p1.cx[i - 1])); p.cy.append(list(p1.cy[i - 1])); p.cz.append(list(p1.cz[i - 1]))
    for i in idx2:                                           # :386-399
        if i == 0:
            p.cx.append(list(p1.cx[-1])); p.cy.append(list(p1.cy[-1])); p.cz.append(list(p1.cz[-1]))
            p.times.append(p2.times[0])
            continue
        p.times.append(p2.times[i])
        p.cx.append(list(p2.cx[i - 1])); p.cy.append(list(p2.cy[i - 1])); p.cz.append(list(p2.cz[i - 1]))
    return p


def eval_source(p, t):
    """ground truth of a trajectory whose intervals run on their own local time; end point held after it"""
    n = len(p.cx)
    k = 0
    while k < n - 1 and p.times[k + 1] <= t:
        k += 1
    dt = min(t - p.times[k], p.times[k + 1] - p.times[k])
    return [((c[k][0] * dt + c[k][1]) * dt + c[k][2]) * dt + c[k][3] for c in (p.cx, p.cy, p.cz)]


# ---------------------------------------------------------------------------------------------
# mader_msgs/DynTraj on the ROS1 wire
# ---------------------------------------------------------------------------------------------
def dyntraj_encode(msg):
    """msg: dict(seq, stamp=(sec,nsec), frame_id: bytes, function: [bytes], bbox: [float],
    pos: (x,y,z), id, is_agent, bendpt: [(x,y,z)], pwp: Pwp) -> bytes (DynTraj.msg:1-9)."""
    b = bytearray()
    b += struct.pack("<III", msg["seq"], msg["stamp"][0], msg["stamp"][1])
    b += struct.pack("<I", len(msg["frame_id"])) + msg["frame_id"]
    b += struct.pack("<I", len(msg["function"]))
    for s in msg["function"]:
        b += struct.pack("<I", len(s)) + s
    b += struct.pack("<I", len(msg["bbox"]))
    for v in msg["bbox"]:
        b += struct.pack("<f", v)
    b += struct.pack("<3d", *msg["pos"])
    b += struct.pack("<i", msg["id"])
    b += struct.pack("<B", 1 if msg["is_agent"] else 0)
    b += struct.pack("<I", len(msg["bendpt"]))
    for v in msg["bendpt"]:
        b += struct.pack("<3d", *v)
    pwp = msg["pwp"]                                         # PieceWisePolTraj.msg:1-4
    b += struct.pack("<I", len(pwp.times))
    for v in pwp.times:
        b += struct.pack("<d", v)
    for arr in (pwp.cx, pwp.cy, pwp.cz):                     # CoeffPoly3.msg:1-4: a b c d
        b += struct.pack("<I", len(arr))
        for c in arr:
            b += struct.pack("<4d", *c)
    return bytes(b)


def dyntraj_decode(data):
    off = [0]

    def take(fmt):
        vals = struct.unpack_from(fmt, data, off[0])
        off[0] += struct.calcsize(fmt)
        return vals

    def take_bytes(n):
        if off[0] + n > len(data):
            raise struct.error("short")
        v = data[off[0]: off[0] + n]
        off[0] += n
        return v

    m = {}
    m["seq"], sec, nsec = take("<III")
    m["stamp"] = (sec, nsec)
    m["frame_id"] = take_bytes(take("<I")[0])
    m["function"] = [take_bytes(take("<I")[0]) for _ in range(take("<I")[0])]
    m["bbox"] = [take("<f")[0] for _ in range(take("<I")[0])]
    m["pos"] = take("<3d")
    m["id"] = take("<i")[0]
    m["is_agent"] = bool(take("<B")[0])
    m["bendpt"] = [take("<3d") for _ in range(take("<I")[0])]
    p = Pwp()
    p.times = [take("<d")[0] for _ in range(take("<I")[0])]
    p.cx = [list(take("<4d")) for _ in range(take("<I")[0])]
    p.cy = [list(take("<4d")) for _ in range(take("<I")[0])]
    p.cz = [list(take("<4d")) for _ in range(take("<I")[0])]
    if not (len(p.cx) == len(p.cy) == len(p.cz)):            # utils.cpp:231-236 aborts
        raise ValueError("coeff_x, coeff_y, coeff_z differ in length")
    m["pwp"] = p
    return m, off[0]


def publish_own_traj(pwp, state_pos, agent_id, drone_radius, bend_xy):
    """NeptuneRos::publishOwnTraj (neptune_ros.cpp:434-480) as a message dict; bend_xy = own base
    followed by the bend points in order."""
    return dict(seq=0, stamp=(0, 0), frame_id=b"", function=[b"", b"", b""],
                bbox=[2 * drone_radius] * 3, pos=tuple(state_pos), id=agent_id, is_agent=True,
                bendpt=[(x, y, 0.0) for x, y in bend_xy], pwp=pwp)


# ---------------------------------------------------------------------------------------------
# plan deque
# ---------------------------------------------------------------------------------------------
def _saturate(v, lo, hi):                                    # utils.cpp:744-767
    return lo if v < lo else (hi if v > hi else v)


class Plan:
    """mt::committedTrajectory plan_ (mader_types.hpp:674-738) + deltaT_ (neptune.hpp:133)."""

    def __init__(self, dc, T_span, lower, upper, runtime_opt, factor_alpha, deltaT0=75):
        self.dc, self.T_span, self.lower, self.upper = dc, T_span, lower, upper
        self.runtime_opt, self.factor_alpha = runtime_opt, factor_alpha
        self.deltaT = int(deltaT0)
        self.content = []

    def reset(self, state):
        self.content = [list(state)]

    def next_goal(self):                                     # neptune.cpp:860-891
        g = list(self.content[0])
        if len(self.content) > 1:
            self.content.pop(0)
            return g, False
        return g, True

    def select_a(self, state_pos, time_now):                 # neptune.cpp:1366-1423
        # deltaT_ is an int, so saturate(int&, int, int) is chosen and the bounds truncate
        self.deltaT = _saturate(self.deltaT, int(self.lower / self.dc), int(self.upper / self.dc))
        size = len(self.content)
        future_index = size - self.deltaT
        k_end = max(future_index, 0)
        if size < math.ceil(self.T_span / self.dc):
            k_end = 0
        k_index = size - 1 - k_end
        A = list(self.content[k_index])
        if future_index < 0:
            A[3:9] = [0.0] * 6
        head = self.content[0]
        if math.sqrt(sum((head[i] - state_pos[i]) ** 2 for i in range(3))) > 1.0:
            A[0:3] = list(state_pos)
        rs = k_index * self.dc - self.runtime_opt if k_end != 0 else self.upper
        rs = _saturate(rs, self.lower - self.runtime_opt, self.upper - self.runtime_opt)
        return dict(A=A, k_index=k_index, k_index_end=k_end, runtime_search=rs,
                    t_start=k_index * self.dc + time_now)

    def splice(self, k_end, traj_out):                       # neptune.cpp:1661-1687
        if len(self.content) - 1 - k_end < 0:
            return False
        del self.content[len(self.content) - k_end - 1:]
        self.content.extend(list(s) for s in traj_out)
        return True

    def update_delta(self, elapsed_ms):                      # neptune.cpp:1713-1720
        states_last_replan = math.ceil(elapsed_ms / (self.dc * 1000))
        self.deltaT = int(max(self.factor_alpha * states_last_replan, 1.0))


def next_start(times, coeff, valid, start, dt, alt_goal=None, r_switch=0.0):
    """Point A of the next bulk-synchronous round (checker of nep_batch_next_starts, include/neptune_frontend.h): the state
    of a committed trajectory — times [n+1], coeff [3][n][4] — at t = start['t_start'] + dt, evaluated like generatePwpOut's
    samples (solver_gurobi_poly.cpp:921-929); at rest at the end point beyond the last knot (Neptune::replanFull takes A
    on the committed plan, neptune.cpp:1366-1399).  start: dict pos/vel/accel/goal (lists of 3) + t_start; returns the new
    dict and the (possibly swapped) alternate goal."""
    out = {k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in start.items()}
    t = start["t_start"] + dt
    out["t_start"] = t
    n = len(times) - 1
    if valid and n >= 1:
        i = 0
        for k in range(1, n):
            if t >= times[k]:
                i = k
        past = t >= times[n]
        u = t - times[i]
        if u < 0.0:
            u = 0.0
        if past:
            u = times[n] - times[n - 1]
        for ax in range(3):
            c = coeff[ax][i]
            out["pos"][ax] = ((c[0] * (u * u * u) + c[1] * (u * u)) + c[2] * u) + c[3]
            out["vel"][ax] = 0.0 if past else (c[0] * (3 * u * u) + c[1] * (2 * u)) + c[2]
            out["accel"][ax] = 0.0 if past else c[0] * (6 * u) + c[1] * 2
    alt = None if alt_goal is None else list(alt_goal)
    if alt is not None:
        dx = out["pos"][0] - out["goal"][0]; dy = out["pos"][1] - out["goal"][1]
        if math.sqrt(dx * dx + dy * dy) < r_switch and math.sqrt(out["vel"][0] * out["vel"][0] + out["vel"][1] * out["vel"][1]) < 0.05:
            out["goal"], alt = alt, list(out["goal"])
    return out, alt
