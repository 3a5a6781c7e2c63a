#!/usr/bin/env python3
"""65 536 environments on one MI355X with a random policy: what replaces a Python loop over
`gym.vector.SyncVectorEnv([make_trading_env] * N)` (reference docs/source/vectorize_env.rst).

    python examples/batched_random_policy.py [--envs 65536] [--steps 500]
"""
import argparse
import os
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_frame(T=20_000, seed=0):
    """A random-walk OHLCV frame with a few `feature_*` columns (the naming rule of
    TradingEnv._set_df, environments.py:130)."""
    rng = np.random.default_rng(seed)
    close = 100 * np.exp(np.cumsum(rng.normal(0, 1e-3, T)))
    df = pd.DataFrame({"open": close, "high": close * 1.001, "low": close * 0.999, "close": close,
                       "volume": rng.uniform(1, 2, T)},
                      index=pd.date_range("2020-01-01", periods=T, freq="h"))
    df["feature_return"] = df["close"].pct_change().fillna(0)
    df["feature_range"] = (df["high"] - df["low"]) / df["close"]
    for i in range(4):
        df[f"feature_noise_{i}"] = rng.normal(0, 1, T)
    return df


def main(envs=65_536, steps=500):
    import torch
    import gym_trading_env_amd as gte
    vec = gte.BatchedTradingEnv(make_frame(), num_envs=envs, positions=[-1, 0, 1], windows=20,
                                trading_fees=1e-4, borrow_interest_rate=3e-6,
                                max_episode_duration=500, autoreset="next_step", seed=1)
    obs, info = vec.reset()
    print("observation batch:", tuple(obs.shape), obs.dtype, obs.device)
    episodes = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        actions = torch.randint(0, 3, (envs,), dtype=torch.int32, device=obs.device)
        obs, reward, terminated, truncated, info = vec.step(actions)
        episodes += int((terminated | truncated).sum())
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{envs * steps / el:,.0f} env-steps/s incl. action sampling and the per-step "
          f"episode count; {episodes} episodes finished")
    print("mean portfolio valuation:", float(np.mean(info["portfolio_valuation"])))
    vec.close()
    return episodes


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65_536)
    ap.add_argument("--steps", type=int, default=500)
    a = ap.parse_args()
    main(a.envs, a.steps)
