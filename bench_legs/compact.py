"""The ONE line bench.py prints on stdout.  The driver keeps an 8 KB tail of the output and parses its last line: the line
carries the contract's fields, the roofline and CPU-baseline objects and a handful of scalar highlights, nothing else
(target <= 4 KB; round 4's 27 KB line could not be parsed).  Every leg's full record is in bench_detail.json."""
import json

LIMIT = 6000


def _r(x, nd=4):
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (nd + 2, x))
    return x


def _pick(d, keys, nd=4):
    return {k: _r(d[k], nd) for k in keys if d is not None and k in d}


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 1] + "…"


def compact_line(d, detail_path="bench_detail.json"):
    """d: the detail record of rank 0 -> dict for the last stdout line"""
    out = _pick(d, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"), 6)
    cfg = d.get("config", {})
    out["config"] = {"workload": _short(cfg.get("workload", ""), 240), **_pick(cfg, ("agents", "obstacles", "scenes_per_gpu", "scenes_in_flight", "scene_groups", "replans_per_step", "replans_per_gpu_per_step")),
                     "sharding": _short(cfg.get("sharding", ""), 200)}
    out.update(_pick(d, ("p50_solve_ms", "p99_solve_ms", "per_gpu_value")))
    out["kernel_ms"] = _pick(d.get("kernel_ms", {}), ("hull", "separator", "qp", "sequence", "exchange_wait"))
    sv = d.get("solver", {})
    out["solver"] = _pick(sv, ("status_ok", "status_relaxed", "status_failed", "ipm_iters_mean", "lines_mean", "rows_solved_mean", "lp_failed"))
    ar = sv.get("active_rows") or {}
    if "replans_with_active_line_rows_frac" in ar:
        out["solver"]["replans_with_active_line_rows_frac"] = _r(ar["replans_with_active_line_rows_frac"])
    rf = d.get("roofline", {})
    out["roofline"] = _pick(rf, ("bound", "kernel", "kernel_ms", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_replan", "replans_per_launch"))
    if rf.get("sequence"):
        out["roofline"]["sequence_frac"] = _r(rf["sequence"].get("frac"))
    f64 = d.get("roofline_fp64")
    if f64:
        out["roofline_fp64"] = _pick(f64, ("achieved", "peak", "unit", "frac", "executed_achieved", "executed_frac"))
    cb = d.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {**_pick(cb, ("value", "unit", "cores", "kind")), "sample": _short(cb.get("sample", ""), 120)}
    rs = d.get("reference_solvers")
    if rs:
        out["reference_solvers"] = {k: _short(rs[k], 40) for k in ("glpk", "gurobi", "eigen") if k in rs}
    rc = d.get("rccl") or {}
    out["rccl"] = {**_pick(rc, ("initialised", "nranks", "one_rank_all_gather_matches")), **({"exchange": _short(rc["exchange"], 80)} if rc.get("exchange") else {}),
                   **({"note": _short(rc["note"], 160)} if rc.get("note") else {})}
    if d.get("per_rank"):
        out["per_rank"] = [{"rank": r["rank"], **_pick(r["kernel_ms"], ("hull", "separator", "qp", "exchange_wait"), 3),
                            "step_ms_p50": _r(r["step_ms_p50"], 3), "wall_s": _r(r["wall_s"], 3)} for r in d["per_rank"]]
    if d.get("scene_digest"):
        out["scene_digest"] = d["scene_digest"][:4]
    # scalar highlights of the other legs (replans/s unless named otherwise)
    hl = {}
    for name in ("long_run", "one_stream", "full_rows", "chain", "moving", "crossing", "single_scene"):
        leg = d.get(name)
        if leg:
            hl[name] = _r(leg["value"], 5)
    for name in ("moving", "crossing"):
        if d.get(name) and "failed_frac" in d[name]:
            hl[name + "_failed_frac"] = _r(d[name]["failed_frac"], 3)
    c5 = d.get("config5")
    if c5:
        hl["config5"] = _r(c5["value"], 5)
        if c5.get("chain"):
            hl["config5_chain"] = _r(c5["chain"]["value"], 5)
    pa = d.get("per_agent_api")
    if pa:
        for k, v in pa.items():
            if k.startswith("config4") and isinstance(v, dict):
                hl["per_agent_api_p50_ms"] = _r(v["sequence_ms"]["p50"])
    if hl:
        out["highlights"] = hl
    out["launch"] = _short(d.get("launch", ""), 60)
    if d.get("graph_notes"):
        out["graph_notes"] = _short("; ".join(d["graph_notes"]), 200)
    out["detail"] = detail_path
    # never print a line the driver cannot read: shed the optional parts, largest first
    for drop in ("per_rank", "highlights", "reference_solvers", "scene_digest", "graph_notes"):
        if len(json.dumps(out)) <= LIMIT:
            break
        if drop == "per_rank" and "per_rank" in out:
            out["per_rank"] = [{"rank": r["rank"], "qp": r.get("qp"), "exchange_wait": r.get("exchange_wait")} for r in out["per_rank"]]
            if len(json.dumps(out)) <= LIMIT:
                break
        out.pop(drop, None)
    return out
