#!/usr/bin/env python3
"""Backtest thousands of precomputed strategies in one launch per K steps (`rollout`):
every env follows its own moving-average crossover with different window lengths over the
same price series, started at the same row; only rewards, flags and valuations are kept.

    python examples/backtest_rollout.py [--strategies 4096]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from batched_random_policy import make_frame  # noqa: E402


def crossover_actions(close, fast, slow, first, K):
    """position index per (step, strategy): 2 (long) when SMA(fast) > SMA(slow) else 0 (short),
    for positions [-1, 0, 1]; decided on the row the env trades at."""
    c = np.concatenate([[0.0], np.cumsum(close)])
    rows = first + np.arange(K)[:, None]                          # [K, 1]
    sma = lambda w: (c[rows + 1] - c[rows + 1 - w]) / w           # [K, S]
    return np.where(sma(fast[None, :]) > sma(slow[None, :]), 2, 0).astype(np.int32)


def main(strategies=4096, K=2000):
    import torch
    import gym_trading_env_amd as gte
    df = make_frame(T=6000, seed=3)
    close = df["close"].to_numpy()
    rng = np.random.default_rng(0)
    fast = rng.integers(3, 40, strategies)
    slow = fast + rng.integers(5, 200, strategies)
    first = 300
    env = gte.BatchedTradingEnv(df, num_envs=strategies, positions=[-1, 0, 1], windows=None,
                                trading_fees=1e-4, borrow_interest_rate=3e-6,
                                initial_position=0, autoreset=None)
    env.reset(inject_idx=np.full(strategies, first, np.int32),
              inject_position_index=np.ones(strategies, np.int32))
    actions = torch.from_numpy(crossover_actions(close, fast, slow, first, K)).cuda()
    out = env.rollout(actions, valuation=True)        # one launch for all K steps
    final = out["valuation"][-1].cpu().numpy()
    best = int(np.argmax(final))
    print(f"{strategies} strategies x {K} steps; best: SMA({fast[best]}) / SMA({slow[best]}) "
          f"-> {final[best]:.1f} from 1000.0; median {np.median(final):.1f}")
    total_log_return = out["reward"].double().sum(0).cpu().numpy()
    assert np.allclose(np.log(final / 1000.0), total_log_return, rtol=1e-3, atol=1e-4)
    env.close()
    return final


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--strategies", type=int, default=4096)
    a = ap.parse_args()
    main(a.strategies)
