#!/usr/bin/env python3
"""steps/s of the N=1 drop-in TradingEnv (config 1 shape) on the GPU box: the price of one
launch + the device->host copies per step.  python tools/dropin_rate.py [--steps 3000]"""
import argparse
import os
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--windows", type=int, default=0)
    a = ap.parse_args()
    import gym_trading_env_amd as gte
    rng = np.random.default_rng(0)
    T = 33_092
    close = 100 * np.exp(np.cumsum(rng.normal(0, 1e-3, T)))
    df = pd.DataFrame({"close": close, "open": close, "high": close * 1.001, "low": close * 0.999,
                       "volume": rng.uniform(1, 2, T)})
    for i in range(5):
        df[f"feature_{i}"] = rng.normal(0, 1, T)
    env = gte.TradingEnv(df=df, positions=[0, 1], verbose=0, windows=a.windows or None)
    env.reset()
    acts = rng.integers(0, 2, a.steps).tolist()
    for x in acts[:200]:
        _, _, d, t, _ = env.step(x)
        if d or t:
            env.reset()
    t0 = time.perf_counter()
    for x in acts:
        _, _, d, t, _ = env.step(x)
        if d or t:
            env.reset()
    el = time.perf_counter() - t0
    print(f"N=1 drop-in TradingEnv: {a.steps / el:,.0f} steps/s ({el / a.steps * 1e6:.1f} us/step)")
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for x in acts[:1000]:
        env.step(x)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)


if __name__ == "__main__":
    main()
