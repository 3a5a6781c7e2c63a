#!/usr/bin/env python3
"""Repeated envs-per-wave sweep of the step kernel (product library), episodes out of phase:
python3 tools/epw_repeat.py ENVS [workload] -> us per step, `reps` interleaved passes per value."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    from gym_trading_env_amd.batched import BatchedTradingEnv
    envs = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    name = sys.argv[2] if len(sys.argv) > 2 else "c3"
    wl = dict(bench.WORKLOADS[name], envs=envs)
    D = wl.get("n_datasets", 1)
    data = [bench.synthetic_dataset(d, wl["T"], wl["n_static"]) for d in range(D)]
    acts = torch.randint(0, 3, (64, envs), dtype=torch.int32, device="cuda")
    steps = 400 if envs > 100_000 else 1000
    res = {}
    for rep in range(3):
        for epw in [int(x) for x in os.environ.get("EPWS", "0 6 7 8 9 10 11 12 13 14 15 16").split()]:
            env = BatchedTradingEnv(data if D > 1 else data[0], num_envs=envs, seed=1, output="torch",
                                    envs_per_wave=epw, kernel_variant=int(os.environ.get("VARIANT", "0")),
                                    nontemporal_obs=int(os.environ.get("NT", "3")),
                                    **bench.env_kwargs(wl))
            env.reset()
            bench.desynchronise(env, acts, wl["max_episode_duration"])
            for i in range(50):
                env.step(acts[i % 64])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                env.step(acts[i % 64])
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / steps * 1e6
            li = env.launch_info()
            res.setdefault((epw, li["envs_per_wave"], li["n_blocks"]), []).append(us)
            env.close()
    for (epw, e, nb), v in res.items():
        print(f"{name} {envs} envs: epw {epw:2d} ({e:2d} per wave, {nb} workgroups) -> "
              + "  ".join(f"{x:7.2f}" for x in v) + f"   min {min(v):7.2f}", flush=True)


if __name__ == "__main__":
    main()
