"""What one replan moves and computes, and the committed counter summaries the line quotes."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def algorithmic_bytes(p, sc, hull_nv, n_states, ent_bytes=0.0):
    """fp64 compulsory traffic of one replan: guess + other agents' hull vertices
    + statics + bases (+ entangle inputs) in; coefficients, cost, status and sampled states out."""
    K = int(sc["guesses"][0]["K"])
    guess = 8 * (12 * K + (K + 1))
    N = p.num_agents
    hull = 16.0 * hull_nv[:, :K].sum() * (N - 1) / N          # per agent: every other agent's hulls
    statics = 16 * sum(len(s) for s in sc["statics"])
    bases = 16 * N
    out = 8 * (12 * K + 1) + 4 + 96 * n_states
    return guess + hull + statics + bases + ent_bytes + out


def algorithmic_flops(K, lines_mean, vertices_mean, iters_mean):
    """fp64 operations of one replan by SURVEY.md §8d's count: separator L (V+4) 3 2 I_lp with I_lp = 10; QP per
    interior-point iteration m n^2 + n^3/3 + 4 m n for the (x, y) system (n = 2K, m = 32K + 4L) and the z system
    (n = K, m = 16K)."""
    L = lines_mean
    sep = L * (vertices_mean + 4) * 3 * 2 * 10
    n_xy, m_xy, n_z, m_z = 2 * K, 32 * K + 4 * L, K, 16 * K
    per_iter = (m_xy * n_xy ** 2 + n_xy ** 3 / 3 + 4 * m_xy * n_xy) + (m_z * n_z ** 2 + n_z ** 3 / 3 + 4 * m_z * n_z)
    return sep + iters_mean * per_iter


def pmc_means(kernel, name="pmc_summary_latest.txt"):
    """mean per dispatch of every counter of `kernel` in a committed rocprofv3 --pmc summary (profiles/), or {}"""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return {}
    # (several instantiations of one template may be in the file — the presolve's redo pass launches qp_reg_kernel<false> over an empty
    # list next to the step's qp_reg_kernel<true>: the block with the most wave cycles is the kernel that did the work)
    cur, blocks = None, {}
    for line in open(path):
        if line.startswith("nep::"):
            cur = line.strip()
        elif cur is not None and cur.split("<")[0] == kernel and "mean" in line:
            parts = line.split()
            blocks.setdefault(cur, {})[parts[0]] = float(parts[2])
    if not blocks:
        return {}
    return max(blocks.values(), key=lambda v: v.get("SQ_WAVE_CYCLES", v.get("SQ_BUSY_CYCLES", 0.0)))


def measured_traffic(kernel, name="pmc_summary_latest.txt"):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc summary of this same command (profiles/):
    FETCH_SIZE and WRITE_SIZE are reported in KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (its
    calibration is for 16 B/lane streams; our 8 B/lane reads are uncalibrated, so this is an upper bound).  A replay of a
    committed file, not a measurement of this run (counters cannot be read from inside the process); None when no
    summary is committed."""
    vals = pmc_means(kernel, name)
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    return None


def executed_flops(kernel, name="pmc_summary_latest.txt"):
    """fp64 operations one launch of `kernel` EXECUTES, from the committed rocprofv3 --pmc summary of this command (its own pass:
    SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 count wave instructions, SQ_INSTS_VALU_MFMA_MOPS_F64 counts MFMA operations / 512):
    64 lanes x (ADD + MUL + TRANS + 2 FMA) + 512 x MFMA_MOPS.  Wave instructions are priced at 64 lanes whatever their EXEC mask, so
    this is the work ISSUED (an upper bound of the useful lane operations); compares, min / max, moves and conversions are not flops.
    A replay of a committed file like `traffic`; None when the summary has no such pass."""
    v = pmc_means(kernel, name)
    need = ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MFMA_MOPS_F64")
    if not all(k in v for k in need):
        return None
    valu = 64.0 * (v["SQ_INSTS_VALU_ADD_F64"] + v["SQ_INSTS_VALU_MUL_F64"] + v.get("SQ_INSTS_VALU_TRANS_F64", 0.0) + 2.0 * v["SQ_INSTS_VALU_FMA_F64"])
    mfma = 512.0 * v["SQ_INSTS_VALU_MFMA_MOPS_F64"]
    return {"flops_per_launch": valu + mfma, "valu_flops_per_launch": valu, "mfma_flops_per_launch": mfma,
            "source": "profiles/%s: 64 x (ADD + MUL + TRANS + 2 FMA)_F64 wave instructions + 512 x MFMA_MOPS_F64" % name}


def quantiles(a):
    a = np.asarray(a, dtype=np.float64)
    return {"p50": float(np.percentile(a, 50)), "p90": float(np.percentile(a, 90)), "p99": float(np.percentile(a, 99)), "max": float(a.max()), "mean": float(a.mean())}


def status_counts(sol):
    st = sol["stats"]["status"].astype(int)
    return {"status_ok": int((st == 0).sum()), "status_relaxed": int((st == 1).sum()), "status_failed": int((st == 2).sum())}


def active_summary(be_, tol=1e-6):
    """which replans are constrained at all: inequality rows with slack < tol at the optimum, over EVERY replan of the leg's last step
    (nep_batch_active_rows: box rows = position / velocity / acceleration bounds, line rows = separating lines)"""
    ar = be_.active_rows(tol)
    nb, nl = ar[:, 0], ar[:, 1]
    return {"sample": "every replan of the last step (%d)" % len(ar), "replans_with_active_rows_frac": float(((nb + nl) > 0).mean()),
            "replans_with_active_line_rows_frac": float((nl > 0).mean()), "replans_with_active_box_rows_frac": float((nb > 0).mean()),
            "active_box_rows_mean": float(nb.mean()), "active_line_rows_mean": float(nl.mean()), "tol_m": tol}


def solve_us_stats(be_):
    """per-replan device time of the interior-point workgroup (nep_stats.solve_us) of the last step"""
    us = be_.solutions(timing=True)["stats"]["solve_us"]
    return {"p50": float(np.percentile(us, 50)), "p90": float(np.percentile(us, 90)), "p99": float(np.percentile(us, 99)),
            "max": float(us.max()), "mean": float(us.mean()), "n": int(us.size)}


def step_quantiles(ms):
    return {"p50": float(np.percentile(ms, 50)), "p99": float(np.percentile(ms, 99)), "max": float(ms.max())}
