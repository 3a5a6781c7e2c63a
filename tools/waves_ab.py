#!/usr/bin/env python3
"""Times the experimental workgroup shapes built by tools/waves_ab.sh on config 3 (episodes out of
phase, like bench.py): us per step by (waves per workgroup, loads in flight, envs per wave)."""
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    from gym_trading_env_amd import _abi
    from gym_trading_env_amd.batched import BatchedTradingEnv
    envs = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    wl = dict(bench.WORKLOADS["c3"], envs=envs)
    feat, close = bench.synthetic_dataset(0, wl["T"], wl["n_static"])
    acts = torch.randint(0, 3, (64, envs), dtype=torch.int32, device="cuda")
    libs = sorted(glob.glob(os.path.join(os.path.dirname(_abi.LIB_PATH), "libgte_w*u*.so")))
    for rep in range(2):
        for lib in libs:
            w = int(os.path.basename(lib)[8])
            for epw in sorted({0, 64 // w}):
                env = BatchedTradingEnv((feat, close), num_envs=envs, seed=1, output="torch",
                                        library_path=lib, envs_per_wave=epw, **bench.env_kwargs(wl))
                env.reset()
                bench.desynchronise(env, acts, wl["max_episode_duration"])
                for i in range(100):
                    env.step(acts[i % 64])
                torch.cuda.synchronize()
                chk = float(env._t["obs"].double().sum()) if rep == 0 else 0.0  # same for every build
                t0 = time.perf_counter()
                for i in range(1000):
                    env.step(acts[i % 64])
                torch.cuda.synchronize()
                us = (time.perf_counter() - t0) / 1000 * 1e6
                li = env.launch_info()
                print(f"{os.path.basename(lib):18s} epw {epw:2d} -> {us:7.2f} us/step  "
                      f"{li['envs_per_wave']} envs/wave x {li['threads_per_block'] // 64} waves, "
                      f"{li['n_blocks']} workgroups, {li.get('resident_workgroups_per_cu')} per CU  obs checksum {chk!r}", flush=True)
                env.close()


if __name__ == "__main__":
    main()
