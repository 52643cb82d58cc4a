"""`python bench.py --gpus N` from a plain shell (the shape of the driver's command, no torch.distributed.run around it) must
launch its own N ranks and print ONE JSON line with n_gpus = N.  One GPU per box here, so the ranks share the device over
gloo (NEP_BENCH_ONE_DEVICE=1: a development aid; on a multi-GPU node the same command runs one rank per GPU over RCCL).
The exchange being replaced: reference neptune/src/neptune_ros.cpp:379-480."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "p50_solve_ms", "p99_solve_ms", "kernel_ms", "roofline", "rccl")


def _run(extra, env_extra, tmp_path=None, merged=False):
    """runs bench.py from a plain shell -> (the parsed LAST line, the detail record or None).  merged: stderr into stdout, as a driver
    that keeps one tail of both streams sees it — the JSON line must still be the last line."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(env_extra)
    detail = str(tmp_path / "detail.json") if tmp_path is not None else ""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + extra + ["--detail", detail]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT if merged else subprocess.PIPE, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout if merged else r.stderr)[-3000:]
    last = r.stdout.rstrip("\n").splitlines()[-1]
    assert last.startswith("{"), r.stdout[-2000:]
    if not merged:
        assert len([ln for ln in r.stdout.splitlines() if ln.startswith("{")]) == 1, r.stdout[-2000:]
    # the driver keeps an 8 KB tail and parses its last line (round 4's 27 KB line came back as parsed: null)
    assert len(last) < 6000, len(last)
    out = json.loads(last)
    for k in REQUIRED:
        assert k in out, k
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in out["roofline"], k
    return out, (json.load(open(detail)) if detail and os.path.exists(detail) else None)


@pytest.mark.gpu
def test_plain_shell_gpus_2_launches_its_own_ranks():
    out, _ = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--scenes", "4", "--agents", "16", "--obstacles", "8",
                   "--no-extra-legs", "--no-cpu-baseline"], {"NEP_BENCH_ONE_DEVICE": "1"})
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 3
    assert out["config"]["replans_per_gpu_per_step"] == 4 * 2 * 8          # scenes per GPU x world x local agents
    assert out["value"] > 0 and out["solver"]["status_failed"] == 0
    assert [r["rank"] for r in out["per_rank"]] == [0, 1]
    for r in out["per_rank"]:
        assert r["qp"] > 0


@pytest.mark.gpu
def test_the_drivers_command_prints_one_short_parsable_line(tmp_path):
    """`python bench.py --gpus 1 --steps 20 --warmup 5` — the driver's command, every leg on but short (fewer scenes in flight, a
    handful of steps per extra leg) — with stderr merged into stdout: the LAST line of everything the process prints must be the JSON
    line, below 6 000 bytes, with the contract's fields, `roofline` and `cpu_baseline`; every leg's full record is in the detail file.
    The timed call this measures: solver_gurobi_poly.cpp:823-826."""
    out, detail = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--scenes", "16", "--aux-steps", "20", "--config5-scenes", "2"], {},
                       tmp_path, merged=True)
    assert out["n_gpus"] == 1 and out["steps"] == 20 and out["warmup"] == 5 and out["dtype"] == "f64" and out["vs_baseline"] is None
    assert out["value"] > 0 and abs(out["value"] - 64 * 16 * 20 / (out["ms_per_step"] * 20e-3)) < 1e-3 * out["value"]
    assert out["config"]["agents"] == 64 and out["config"]["obstacles"] == 20 and "64 agents + 20 static obstacles" in out["config"]["workload"]
    assert 0 < out["roofline"]["frac"] < 1 and out["roofline"]["bound"] == "hbm" and out["roofline"]["peak"] == 8000.0
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["cores"] >= 1
    assert out["rccl"]["nranks"] == [1]
    for k in ("moving", "crossing", "chain", "full_rows", "config5", "config5_chain", "per_agent_api_p50_ms"):
        assert out["highlights"][k] > 0, k
    assert detail is not None
    for leg in ("long_run", "launch_order_off", "reference_tolerances", "full_rows", "chain", "moving", "crossing", "single_scene", "small_configs",
                "per_agent_api", "config5", "cpu_baseline", "roofline_fp64"):
        assert detail.get(leg), leg
    assert detail["value"] == pytest.approx(out["value"], rel=1e-5)
    assert detail["config5"]["chain"]["ent_overflow"] == 0
    # the headline times the handle's default solve path and says so; the every-row solve of the same replans agrees with it
    assert "verified line presolve 4 m, polish on" in out["config"]["workload"] and detail["solver"]["line_cull_radius_m"] == 4.0
    vd = detail["full_rows"]["vs_default"]
    assert vd["status_mismatches"] == 0 and vd["coeff_diff_max"] <= 1e-6, vd
    assert out["p50_solve_ms"] == pytest.approx(detail["per_agent_api"]["config4_64_agents_20_obstacles"]["sequence_ms"]["p50"]) and "solve_ms_definition" in detail


@pytest.mark.gpu
def test_config5_workload_two_ranks_equal_one_rank(tmp_path):
    """`--workload config5` (BASELINE configs[4]: entangle rows on, agents block-sharded by id, hull blocks with the tethers' samples and
    bend points all-gathered per round) launched from a plain shell with two ranks on the one device, against the same command with one
    rank: the replans of the scenes both runs hold are byte-identical after the warm-up rounds (sizes cut so that scene generation takes
    seconds; the exchange being replaced: neptune_ros.cpp:379-480)."""
    common = ["--workload", "config5", "--agents", "32", "--obstacles", "12", "--scenes", "2", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"]
    one, d1 = _run(["--gpus", "1"] + common, {}, tmp_path)
    two, d2 = _run(["--gpus", "2"] + common, {"NEP_BENCH_ONE_DEVICE": "1"}, tmp_path)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert two["config"]["replans_per_gpu_per_step"] == one["config"]["replans_per_gpu_per_step"] == 2 * 32
    assert two["config"]["replans_per_step"] == 2 * one["config"]["replans_per_step"]
    assert [r["rank"] for r in two["per_rank"]] == [0, 1]
    assert len(d1["scene_digest"]) == 2 and d2["scene_digest"][:2] == d1["scene_digest"]          # scenes 0, 1: in both runs
    assert d1["solver"]["lines_mean"] > 0 and one["value"] > 0 and two["value"] > 0


def test_gpus_n_without_a_gpu_fails_loudly():
    """no GPU (or fewer than N): a message, not a hang and not a silent single-rank number"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    env = dict(os.environ); env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extra-legs",
                        "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
