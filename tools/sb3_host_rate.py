#!/usr/bin/env python3
"""Host-side cost of SB3TradingVecEnv.step() (the VecEnv face of the batch, sb3.py): wall time
per step minus the device work and the device->host copy it waits for, i.e. what the Python
adapter itself costs.  python tools/sb3_host_rate.py [--envs 4096]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=300)
    a = ap.parse_args()
    import gym_trading_env_amd as gte
    rng = np.random.default_rng(0)
    T = 20_000
    close = 100 * np.exp(np.cumsum(rng.normal(0, 1e-3, T)))
    feat = rng.normal(0, 1, (T, 14)).astype(np.float32)
    vec = gte.SB3TradingVecEnv((feat, close), a.envs, positions=[-1, 0, 1], trading_fees=1e-4,
                               max_episode_duration=200, copy=False)
    vec.reset()
    acts = rng.integers(0, 3, (64, a.envs))
    for i in range(210):  # past the first wave of episode ends
        vec.step(acts[i % 64])
    env = vec.env
    t_total = t_core = 0.0
    ended = 0
    for i in range(a.steps):
        t0 = time.perf_counter()
        obs, rew, dones, infos = vec.step(acts[i % 64])
        t1 = time.perf_counter()
        ended += int(dones.sum())
        # the same step without the adapter: launch + the one device->host transfer
        t2 = time.perf_counter()
        env.step(acts[(i + 1) % 64])
        t3 = time.perf_counter()
        t_total += t1 - t0
        t_core += t3 - t2
    v = infos[0]["portfolio_valuation"]
    print(f"{a.envs} envs: SB3 step() {t_total / a.steps * 1e3:.3f} ms, batch step() alone "
          f"{t_core / a.steps * 1e3:.3f} ms -> adapter {max(0.0, t_total - t_core) / a.steps * 1e3:.3f} ms "
          f"per step ({ended / a.steps:.1f} episode ends per step; infos[0]['portfolio_valuation'] = {v:.2f})")
    vec.close()


if __name__ == "__main__":
    main()
