#!/usr/bin/env python3
"""Per-step time of gte_rollout (K steps fused into one launch) next to single steps, one GPU.

    python tools/rollout_bench.py [--workload c3] [--k 8 32 128] [--reps 5]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--k", type=int, nargs="+", default=[8, 32, 128])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--envs", type=int, default=0)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--affinity", type=int, default=0, help="L2-affinity period (0 default, -1 off)")
    ap.add_argument("--lib", default=None, help="another build of libgte.so (A/B)")
    ap.add_argument("--desync", action="store_true",
                    help="spread the episode phases first (bench.desynchronise): ~N/500 envs end in "
                         "every step instead of all of them every 500 steps")
    a = ap.parse_args()
    import torch
    from gym_trading_env_amd.batched import BatchedTradingEnv
    wl = bench.WORKLOADS[a.workload]
    N = a.envs or wl["envs"]
    D = wl["n_datasets"]  # config 5: 128 resident datasets, a dataset switch at every episode
    data = [bench.synthetic_dataset(d, wl["T"], wl["n_static"]) for d in range(D)]
    dev = torch.device("cuda", 0)
    env = BatchedTradingEnv(data if D > 1 else data[0], num_envs=N, seed=1, output="torch",
                            kernel_variant=a.variant, affinity_period=a.affinity, library_path=a.lib,
                            **bench.env_kwargs(wl))
    env.reset()
    W = wl["windows"] or 1
    b_alg = bench.algorithmic_bytes(W, wl["n_static"] + 2, wl["n_static"], 2)
    Kmax = max(a.k)
    acts = torch.randint(0, 3, (Kmax, N), dtype=torch.int32, device=dev)
    if a.desync:
        bench.desynchronise(env, acts, wl["max_episode_duration"])
    for i in range(100):
        env.step(acts[i % Kmax])

    def report(tag, us):
        print(f"{tag:44s} {us:8.2f} us/step  {N / us:9.1f} M env-steps/s  "
              f"{N * b_alg / us / 1e3:8.1f} GB/s alg", flush=True)

    env.timer_start()
    for i in range(1000):
        env.step(acts[i % Kmax])
    report("single steps", env.timer_stop() * 1e3 / 1000)
    for K in a.k:
        for keep in (True, False):
            if keep and K * N * W * (wl["n_static"] + 2) * 4 > 40e9:
                continue
            env.rollout(acts[:K], keep_obs=keep)  # warm-up (allocations)
            torch.cuda.synchronize()
            best = []
            for _ in range(a.reps):
                # result tensors are allocated by torch before the timed launch
                env.timer_start()
                env.rollout(acts[:K], keep_obs=keep)
                best.append(env.timer_stop() * 1e3 / K)
            report(f"rollout K={K} " + ("obs of every step" if keep else "last obs only"),
                   sorted(best)[len(best) // 2])
    env.close()


if __name__ == "__main__":
    main()
