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


def _run(extra, env_extra):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(env_extra)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + extra
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_plain_shell_gpus_2_launches_its_own_ranks():
    out = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--scenes", "4", "--agents", "16", "--obstacles", "8",
                "--no-extra-legs", "--no-cpu-baseline"], {"NEP_BENCH_ONE_DEVICE": "1"})
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 3
    assert out["config"]["replans_per_gpu_per_step"] == 4 * 2 * 8          # scenes per GPU x world x local agents
    assert out["value"] > 0 and out["solver"]["status_failed"] == 0
    assert [r["rank"] for r in out["per_rank"]] == [0, 1]
    for r in out["per_rank"]:
        assert r["kernel_ms"]["qp"] > 0


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
