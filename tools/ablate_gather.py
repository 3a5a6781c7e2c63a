#!/usr/bin/env python3
"""Where the step kernel's gather time goes: the product with parts of the copy loop switched off by
`debug_flags` (timing only — results are wrong with them): 1 = no gather at all, 2 = no dynamic-column
patch, 8 = no window loads (constants stored).  One process, interleaved rounds.
python3 tools/ablate_gather.py [workload=c3]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    from gym_trading_env_amd.batched import BatchedTradingEnv
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    wl = bench.WORKLOADS[name]
    N, D = wl["envs"], wl["n_datasets"]
    data = [bench.synthetic_dataset(d, wl["T"], wl["n_static"]) for d in range(D)]
    acts = torch.randint(0, 3, (64, N), dtype=torch.int32, device="cuda")
    cases = [("product", {}), ("no dynamic-column patch (2)", dict(debug_flags=2)),
             ("no window loads (8)", dict(debug_flags=8)), ("neither (10)", dict(debug_flags=10)),
             ("no gather (1)", dict(debug_flags=1)),
             ("product, re-sort every 8 steps", dict(affinity_period=8)),
             ("product, re-sort every 128 steps", dict(affinity_period=128)),
             ("product, identity order", dict(affinity_period=-1))]
    envs = []
    for label, kw in cases:
        env = BatchedTradingEnv(data if D > 1 else data[0], num_envs=N, seed=1, output="torch",
                                **dict(bench.env_kwargs(wl), **kw))
        env.reset()
        bench.desynchronise(env, acts, wl["max_episode_duration"])
        envs.append(env)
    res = [[] for _ in cases]
    for r in range(3):
        for k, env in enumerate(envs):
            for i in range(40):
                env.step(acts[i % 64])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(300):
                env.step(acts[i % 64])
            torch.cuda.synchronize()
            res[k].append((time.perf_counter() - t0) / 300 * 1e6)
    print(f"# {name}: {N} envs, episodes out of phase, us per step (3 interleaved rounds of 300 steps)")
    for (label, _), v in zip(cases, res):
        print(f"{label:36s} " + "  ".join(f"{x:7.2f}" for x in v))
    for env in envs:
        env.close()


if __name__ == "__main__":
    main()
