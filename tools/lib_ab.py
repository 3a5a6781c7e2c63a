#!/usr/bin/env python3
"""A/B of two BUILDS of libgte in one process on one box, interleaved rounds (box-to-box variance
is larger than most kernel changes): python3 tools/lib_ab.py [workload=c3] [rounds=3] libA.so libB.so ...
A library older than the current ctypes table may lack newer entry points: they are skipped."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    from gym_trading_env_amd import _abi
    from gym_trading_env_amd.batched import BatchedTradingEnv
    name, rounds = sys.argv[1], int(sys.argv[2])
    libs = [os.path.abspath(p) for p in sys.argv[3:]]
    wl = bench.WORKLOADS[name]
    N, D = wl["envs"], wl["n_datasets"]
    data = [bench.synthetic_dataset(d, wl["T"], wl["n_static"]) for d in range(D)]
    acts = torch.randint(0, 3, (64, N), dtype=torch.int32, device="cuda")
    full = dict(_abi.SYMBOLS)
    envs = []
    for p in libs:
        have = C.CDLL(p)
        _abi.SYMBOLS = {k: v for k, v in full.items() if hasattr(have, k)}
        env = BatchedTradingEnv(data if D > 1 else data[0], num_envs=N, seed=1, output="torch",
                                library_path=p, **bench.env_kwargs(wl))
        env.reset()
        bench.desynchronise(env, acts, wl["max_episode_duration"])
        envs.append(env)
    _abi.SYMBOLS = full
    print(f"# {name}: {N} envs, episodes out of phase; us per step, {rounds} interleaved rounds of 400 steps")
    res = [[] for _ in libs]
    for r in range(rounds):
        for k, env in enumerate(envs):
            for i in range(50):
                env.step(acts[i % 64])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(400):
                env.step(acts[i % 64])
            torch.cuda.synchronize()
            res[k].append((time.perf_counter() - t0) / 400 * 1e6)
    for p, v, env in zip(libs, res, envs):
        li = env.launch_info()
        print(f"{os.path.basename(p):24s} " + "  ".join(f"{x:7.2f}" for x in v) +
              f"   ({li['envs_per_wave']} envs/wave, {li['n_blocks']} workgroups, {li['resident_workgroups_per_cu']} resident/CU)")
    for env in envs:
        env.close()


if __name__ == "__main__":
    main()
