#!/usr/bin/env python3
"""Per-step cost of Python callables in the batch (vectorised over a BatchedHistory on the device):
python3 tools/callable_bench.py [envs]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def reward_function(history):  # the reference's vectorised example (examples/example_vectorized_environment.py)
    return np.log(history["portfolio_valuation", -1] / history["portfolio_valuation", -2])


def dyn_last_position(history):
    return history["position", -1]


def dyn_real_position(history):
    return history["real_position", -1]


def main():
    import torch
    from gym_trading_env_amd.batched import BatchedTradingEnv
    envs = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    wl = dict(bench.WORKLOADS["c3"], envs=envs)
    feat, close = bench.synthetic_dataset(0, wl["T"], wl["n_static"])
    acts = torch.randint(0, 3, (64, envs), dtype=torch.int32, device="cuda")
    base = bench.env_kwargs(wl)
    cases = [("built-in reward and features", {}),
             ("log_steps=2 only", dict(log_steps=2)),
             ("Python reward_function", dict(reward_function=reward_function)),
             ("Python reward + 2 Python dynamic features",
              dict(reward_function=reward_function,
                   dynamic_feature_functions=[dyn_last_position, dyn_real_position]))]
    for name, kw in cases:
        k = dict(base)
        k.update(kw)
        env = BatchedTradingEnv((feat, close), num_envs=envs, seed=1, output="torch", **k)
        env.reset()
        for i in range(60):
            env.step(acts[i % 64])
        torch.cuda.synchronize()
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            for i in range(200):
                env.step(acts[i % 64])
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 200 * 1e6)
        print(f"{envs} envs, {name:45s} " + "  ".join(f"{x:8.2f}" for x in ts) + " us/step", flush=True)
        env.close()


if __name__ == "__main__":
    main()
