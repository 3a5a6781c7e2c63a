#!/usr/bin/env python3
"""cProfile of BatchedTradingEnv.step with a Python reward_function (host side of the vectorised
callable path): python3 tools/callable_profile.py [envs] [ndyn]"""
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from callable_bench import reward_function, dyn_last_position, dyn_real_position  # noqa: E402


def main():
    import torch
    from gym_trading_env_amd.batched import BatchedTradingEnv
    envs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    ndyn = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    wl = dict(bench.WORKLOADS["c3"], envs=envs)
    feat, close = bench.synthetic_dataset(0, wl["T"], wl["n_static"])
    acts = torch.randint(0, 3, (64, envs), dtype=torch.int32, device="cuda")
    kw = dict(bench.env_kwargs(wl), reward_function=reward_function)
    if ndyn:
        kw["dynamic_feature_functions"] = [dyn_last_position, dyn_real_position]
    env = BatchedTradingEnv((feat, close), num_envs=envs, seed=1, output="torch", **kw)
    env.reset()
    for i in range(60):
        env.step(acts[i % 64])
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(500):
        env.step(acts[i % 64])
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)
    env.close()


if __name__ == "__main__":
    main()
