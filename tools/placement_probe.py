#!/usr/bin/env python3
"""Does the step time depend on where the buffers land?  Builds the same env several times in one
process with GTE_DEBUG_ALLOC=1 (the library prints every device allocation to stderr) and prints
the time per step next to a digest of the addresses.  python3 tools/placement_probe.py [c5|c3] [envs]"""
import os
import sys
import time

os.environ["GTE_DEBUG_ALLOC"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    from gym_trading_env_amd.batched import BatchedTradingEnv
    name = sys.argv[1] if len(sys.argv) > 1 else "c5"
    wl = dict(bench.WORKLOADS[name])
    envs = int(sys.argv[2]) if len(sys.argv) > 2 else wl["envs"]
    D = wl.get("n_datasets", 1)
    data = [bench.synthetic_dataset(d, wl["T"], wl["n_static"]) for d in range(D)]
    acts = torch.randint(0, 3, (64, envs), dtype=torch.int32, device="cuda")
    for k in range(6):
        sys.stderr.write(f"ENV {k} begin\n"); sys.stderr.flush()
        env = BatchedTradingEnv(data if D > 1 else data[0], num_envs=envs, seed=1, output="torch",
                                **bench.env_kwargs(wl))
        env.reset()
        bench.desynchronise(env, acts, wl["max_episode_duration"])
        for i in range(50):
            env.step(acts[i % 64])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(1000):
            env.step(acts[i % 64])
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 1000 * 1e6
        sys.stderr.write(f"ENV {k} obs {env._t['obs'].data_ptr():#x} reward {env._t['reward'].data_ptr():#x} "
                         f"-> {us:.2f} us/step\n"); sys.stderr.flush()
        env.close()


if __name__ == "__main__":
    main()
