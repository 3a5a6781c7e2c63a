#!/usr/bin/env python3
"""The reference's single-environment loop, unchanged except for the import: constructor
arguments, reset/step tuple, History access, add_metric, episode metrics line.

    python examples/single_env_dropin.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from batched_random_policy import make_frame  # noqa: E402


def main(verbose=1):
    import gym_trading_env_amd as gte
    env = gte.TradingEnv(df=make_frame(T=3000, seed=1), positions=[-1, -0.5, 0, 0.5, 1, 1.5, 2],
                         windows=5, trading_fees=0.01 / 100, borrow_interest_rate=0.0003 / 100,
                         portfolio_initial_value=1000, max_episode_duration=500, verbose=verbose)
    env.add_metric("Position Changes",
                   lambda history: int(np.sum(np.diff(history["position"]) != 0)))
    env.add_metric("Episode Length", lambda history: len(history["position"]))
    np.random.seed(0)
    done = truncated = False
    observation, info = env.reset()
    steps = 0
    while not done and not truncated:
        action = int(np.random.randint(len(env.positions)))
        observation, reward, done, truncated, info = env.step(action)
        steps += 1
    metrics = env.get_metrics()
    env.close()
    return steps, metrics


if __name__ == "__main__":
    main()
