"""cpu_baseline: the CPU oracle (kind "port": the reference needs Gurobi / GLPK / CGAL, absent here) timed on the host cores on
a bounded sample of the same workload.  The only place of the bench that touches oracle/ (as the thing timed beside the
product, never as a fallback of it)."""
import os
import time

import numpy as np


def cpu_baseline(p, scenes, budget_s=12.0, ent=None):
    """scenes: scene dicts of the timed workload; ent: None or per-scene dense entangle case blocks [N][8][N] (config 5)"""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    oracle.lib()
    cores = os.cpu_count() or 1
    jobs = [(k, a) for k in range(len(scenes)) for a in range(p.num_agents)] * 64   # bounded by time below
    t0 = time.perf_counter()
    done = 0

    def one(job):
        k, a = job
        s = scenes[k]
        kw = {"case_id": ent[k][a]} if ent is not None else {}
        return oracle.replan(p, a + 1, s["committed"], s["guesses"][a], s["statics"], **kw)["status"]
    chunk = max(cores * 4, 64)
    with ThreadPoolExecutor(cores) as ex:     # ctypes releases the GIL: one solver thread per core
        for k in range(0, len(jobs), chunk):
            list(ex.map(one, jobs[k:k + chunk]))
            done += len(jobs[k:k + chunk])
            if time.perf_counter() - t0 > budget_s:
                break
    dt = time.perf_counter() - t0
    out = {"value": done / dt, "unit": "replans/s", "cores": cores, "kind": "port",
           "sample": "%d replans of the same scenes (seeds 0..), one oracle thread per core, %.1f s" % (done, dt)}
    # the reference's own solvers, where a box has them (neither is in this image: then the line says so)
    from oracle import reference_solvers as rs
    out["reference_solvers"] = rs.probe()
    if rs.glpk_lib() is not None:
        rng = np.random.default_rng(0)
        A = rng.uniform(-1, 1, (200, 8, 2)); B = rng.uniform(-1, 1, (200, 4, 2)) + np.array([3.0, 0.0])
        t1 = time.perf_counter()
        for a_, b_ in zip(A, B):
            rs.glpk_separator(a_, b_)
        out["reference_solvers"]["glpk_us_per_lp"] = (time.perf_counter() - t1) / 200 * 1e6
    return out
