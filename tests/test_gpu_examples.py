"""The scripts under examples/ run end to end on the GPU (small sizes)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def test_batched_random_policy_example():
    import batched_random_policy
    assert batched_random_policy.main(envs=2048, steps=600) >= 2048  # every 500-step episode ended


def test_backtest_rollout_example():
    import backtest_rollout
    final = backtest_rollout.main(strategies=512, K=800)
    assert final.shape == (512,) and np.isfinite(final).all() and final.std() > 0


def test_single_env_dropin_example(capsys):
    import single_env_dropin
    steps, metrics = single_env_dropin.main(verbose=1)
    assert steps == 499  # max_episode_duration=500 lasts 499 steps, like the reference
    assert set(metrics) == {"Market Return", "Portfolio Return", "Position Changes", "Episode Length"}
    assert metrics["Episode Length"] == 500
    assert "Market Return" in capsys.readouterr().out


def test_vectorized_custom_reward_example(capsys):
    import vectorized_custom_reward
    mean = vectorized_custom_reward.main(envs=512, steps=150)
    assert np.isfinite(mean)
    out = capsys.readouterr().out
    assert "Position Changes" in out and "data_volume" in out


def test_c_abi_demo_runs(tmp_path):
    """examples/c_abi_demo.c — the hot path and the RCCL return gather driven from plain C."""
    import subprocess
    from test_host_cpu import _build_c_demo
    exe = _build_c_demo(tmp_path)
    r = subprocess.run([exe, "2048", "120"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "c_abi_demo ok: 2048 envs x 120 steps" in r.stdout and "episode ends seen" in r.stdout
