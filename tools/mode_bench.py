#!/usr/bin/env python3
"""Step time of the config-3 shape in the other auto-reset modes / with the optional features on:
python3 tools/mode_bench.py [envs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    from gym_trading_env_amd.batched import BatchedTradingEnv
    envs = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    wl = dict(bench.WORKLOADS["c3"], envs=envs)
    feat, close = bench.synthetic_dataset(0, wl["T"], wl["n_static"])
    acts = torch.randint(0, 3, (64, envs), dtype=torch.int32, device="cuda")
    base = bench.env_kwargs(wl)
    cases = [
        ("next_step (headline)", {}),
        ("next_step, shared-TU build of the kernel", dict(kernel_variant=64)),
        ("same_step", dict(autoreset="same_step")),
        ("same_step + final_obs", dict(autoreset="same_step", final_obs=True)),
        ("next_step + log_steps=2", dict(log_steps=2)),
        ("same_step + final_obs + log_steps=2", dict(autoreset="same_step", final_obs=True, log_steps=2)),
        ("disabled (caller resets)", dict(autoreset="disabled")),
    ]
    for name, kw in cases:
        k = dict(base)
        k.update(kw)
        env = BatchedTradingEnv((feat, close), num_envs=envs, seed=1, output="torch", **k)
        env.reset()
        if k.get("autoreset") != "disabled":
            bench.desynchronise(env, acts, wl["max_episode_duration"])
        for i in range(100):
            env.step(acts[i % 64])
        torch.cuda.synchronize()
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            for i in range(300):
                env.step(acts[i % 64])
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 300 * 1e6)
        li = env.launch_info()
        print(f"{name:40s} " + "  ".join(f"{x:7.2f}" for x in ts) + f" us/step   {li['envs_per_wave']} envs/wave, "
              f"{li['n_blocks']} workgroups, {li['obs_stores']}", flush=True)
        env.close()


if __name__ == "__main__":
    main()
