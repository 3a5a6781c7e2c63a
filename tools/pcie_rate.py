#!/usr/bin/env python3
"""Compatibility-mode rate: the same step as bench.py but with the observations copied to
host memory every step (`output="numpy"`), i.e. PCIe-inclusive.  Never the headline value."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main(copy=True):
    import numpy as np
    from gym_trading_env_amd.batched import BatchedTradingEnv
    wl = bench.WORKLOADS["c3"]
    N = wl["envs"]
    feat, close = bench.synthetic_dataset(0, wl["T"], wl["n_static"])
    env = BatchedTradingEnv((feat, close), num_envs=N, seed=1, output="numpy", copy=copy,
                            **bench.env_kwargs(wl))
    env.reset()
    acts = np.random.default_rng(0).integers(0, 3, (16, N)).astype(np.int32)
    for i in range(3):
        env.step(acts[i])
    steps, t0 = 30, time.perf_counter()
    for i in range(steps):
        obs, reward, term, trunc, _ = env.step(acts[i % 16])
    el = time.perf_counter() - t0
    mb = obs.nbytes / 1e6
    print(f"output=numpy, copy={copy}: {el / steps * 1e3:.2f} ms/step, {N * steps / el / 1e6:.1f} M env-steps/s, "
          f"{mb:.0f} MB of observations per step over PCIe = {mb * steps / el / 1e3:.1f} GB/s")


    env.close()


if __name__ == "__main__":
    main(True)
    main(False)
