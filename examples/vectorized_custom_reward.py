#!/usr/bin/env python3
"""The reference's vectorised example (examples/example_vectorized_environment.py), on the batch:
the same constructor arguments — its own Python `reward_function(history)` included — given to
`BatchedTradingEnv` instead of `gym.make_vec("TradingEnv", num_envs=3, ...)`.

The reward function is written for ONE env's History; here it is called once per step for ALL
envs with a `BatchedHistory` whose entries are device arrays (one value per env), so the NumPy
formula runs unchanged and vectorised on the GPU.  A custom dynamic feature and a custom metric
(docs/source/customization.rst) ride along.

    python examples/vectorized_custom_reward.py [--envs 4096] [--steps 300]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from batched_random_policy import make_frame  # noqa: E402


def reward_function(history):  # verbatim from the reference's example
    return np.log(history["portfolio_valuation", -1] / history["portfolio_valuation", -2])  # log (p_t / p_t-1 )


def dynamic_feature_drawdown(history):
    """Valuation relative to the initial 1 000 (one value per env)."""
    return history["portfolio_valuation", -1] / 1000.0 - 1.0


def main(envs=4096, steps=300):
    import torch
    import gym_trading_env_amd as gte
    df = make_frame(T=5_000)
    env = gte.BatchedTradingEnv(
        df, num_envs=envs,
        name="BTCUSD", windows=5, positions=[-1, -0.5, 0, 0.5, 1, 1.5, 2], initial_position=0,
        trading_fees=0.01 / 100, borrow_interest_rate=0.0003 / 100,
        reward_function=reward_function, portfolio_initial_value=1000,
        dynamic_feature_functions=[gte.dynamic_feature_last_position_taken, dynamic_feature_drawdown],
        max_episode_duration=100, log_steps=128)
    env.add_metric("Position Changes", lambda history: np.sum(np.diff(history["position"]) != 0))
    observation, info = env.reset()
    total = torch.zeros(envs, dtype=torch.float64, device=observation.device)
    for k in range(steps):
        actions = torch.randint(0, 7, (envs,), dtype=torch.int32, device=observation.device)
        observation, reward, done, truncated, info = env.step(actions)
        total += reward
        if k == steps - 1 or bool((done | truncated).any()) and k > steps - 110:
            metrics = env.episode_metrics()
            if len(metrics["env_ids"]):
                e = int(metrics["env_ids"][0])
                print(f"step {k}: {len(metrics['env_ids'])} episodes ended; env {e}: Market Return "
                      f"{metrics['Market Return'][0]}, Portfolio Return {metrics['Portfolio Return'][0]}, "
                      f"Position Changes {metrics['Position Changes'][0]}")
                break
    print("info keys:", [k for k in info.keys() if not k.startswith("portfolio_distribution")])
    print("date of env 0:", info["date"][0], " data_volume:", float(info["data_volume"][0]))
    print("mean summed reward per env:", float(total.mean()))
    env.close()
    return float(total.mean())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=300)
    a = ap.parse_args()
    main(a.envs, a.steps)
