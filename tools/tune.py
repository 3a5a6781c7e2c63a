#!/usr/bin/env python3
"""In-process A/B of step-kernel variants on one GPU (interleaved rounds, HIP events).

    python tools/tune.py [--workload c3] [--steps 300] [--rounds 5] variant ...
variant = kernel_variant:epw:nt[:debug_flags[:affinity_period]], e.g. 0:16:1 1:16:1 0:16:1:1 0:16:1:0:-1
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--envs", type=int, default=0)
    ap.add_argument("--duration", type=int, default=0)
    ap.add_argument("--region-affine", type=int, default=0,
                    help="K>0: inject start rows so that workgroup b (64 envs) only reads table "
                         "region b %% K (potential of an XCD/L2-affine env order; K=8 XCDs)")
    ap.add_argument("--lib", default=None, help="alternative libgte.so to load (A/B of builds)")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    import torch
    from gym_trading_env_amd.batched import BatchedTradingEnv
    wl = dict(bench.WORKLOADS[a.workload])
    if a.duration:
        wl["max_episode_duration"] = a.duration if a.duration > 0 else "max"
    N = a.envs or wl["envs"]
    feat, close = bench.synthetic_dataset(0, wl["T"], wl["n_static"])
    dev = torch.device("cuda", 0)
    acts = torch.randint(0, 3, (64, N), dtype=torch.int32, device=dev)
    envs = {}
    for v in a.variants:
        kv, epw, nt, *rest = v.split(":")
        dbg = rest[:1]
        aff = int(rest[1]) if len(rest) > 1 else 0
        envs[v] = BatchedTradingEnv((feat, close), num_envs=N, seed=1, output="torch",
                                    kernel_variant=int(kv), envs_per_wave=int(epw),
                                    nontemporal_obs=int(nt),
                                    debug_flags=int(dbg[0]) if dbg else 0, affinity_period=aff,
                                    library_path=a.lib,
                                    **bench.env_kwargs(wl))
        if a.region_affine:
            K = a.region_affine
            T, W, dur = wl["T"], wl["windows"] or 1, wl["max_episode_duration"]
            lo, hi = W - 1, T - dur - (W - 1)
            rng = np.random.default_rng(0)
            region = (np.arange(N) // 64) % K
            span = (hi - lo) // K
            idx0 = (lo + region * span + rng.integers(0, span, N)).astype(np.int32)
            envs[v].reset(inject_idx=idx0)
        else:
            envs[v].reset()
        for i in range(50):
            envs[v].step(acts[i % 64])
    torch.cuda.synchronize()
    res = {v: [] for v in envs}
    for r in range(a.rounds):
        for v, e in envs.items():
            e.timer_start()
            for i in range(a.steps):
                e.step(acts[i % 64])
            res[v].append(e.timer_stop() * 1e3 / a.steps)
    W = wl["windows"] or 1
    b_alg = bench.algorithmic_bytes(W, wl["n_static"] + 2, wl["n_static"], 2)
    for v, t in res.items():
        med, mn = float(np.median(t)), float(np.min(t))
        print(f"{v:14s} {envs[v].launch_info()} us/step median {med:7.2f} min {mn:7.2f}  "
              f"-> {N / med:8.1f} M env-steps/s, {b_alg * N / med / 1e3:7.1f} GB/s alg "
              f"({b_alg * N / med / 1e3 / 80:.1f}% of 8 TB/s)", flush=True)


if __name__ == "__main__":
    main()
