#!/usr/bin/env python3
"""A/B of gte_rollout (observations of every step) between env configurations in ONE process,
interleaved, every repetition printed:  python tools/rollout_ab.py [--k 128] name=kw,kw ...
e.g.  on=affinity_period:0 off=affinity_period:-1"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--k", type=int, default=128)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--warm", type=int, default=50, help="single steps before the rollouts")
    ap.add_argument("--desync", action="store_true", help="spread the episode phases first")
    ap.add_argument("--reuse", action="store_true", help="write every rollout into the same output tensors")
    ap.add_argument("--share", action="store_true", help="--reuse with ONE set of output tensors for all variants")
    ap.add_argument("--prealloc", action="store_true",
                    help="--share with the output tensors allocated BEFORE anything else in the process")
    ap.add_argument("--no-obs", action="store_true", help="state-only rollouts (keep_obs=False)")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    import torch
    from gym_trading_env_amd.batched import BatchedTradingEnv
    wl = bench.WORKLOADS[a.workload]
    N = wl["envs"]
    D = wl["n_datasets"]
    data = [bench.synthetic_dataset(d, wl["T"], wl["n_static"]) for d in range(D)]
    dev = torch.device("cuda", 0)
    acts = torch.randint(0, 3, (a.k, N), dtype=torch.int32, device=dev)
    pre = None
    if a.prealloc:
        a.share = True
        W, F = wl["windows"] or 1, wl["n_static"] + 2
        pre = {"reward": torch.empty((a.k, N), dtype=torch.float32, device=dev),
               "terminated": torch.empty((a.k, N), dtype=torch.bool, device=dev),
               "truncated": torch.empty((a.k, N), dtype=torch.bool, device=dev),
               "obs": torch.empty((a.k, N, W, F), dtype=torch.float32, device=dev)}
    envs = {}
    for v in a.variants:
        name, _, kws = v.partition("=")
        kw = {k: int(x) for k, x in (t.split(":") for t in kws.split(",") if t)}
        e = BatchedTradingEnv(data if D > 1 else data[0], num_envs=N, seed=1, output="torch", **kw,
                              **bench.env_kwargs(wl))
        e.reset()
        if a.desync:
            bench.desynchronise(e, acts, wl["max_episode_duration"])
        for i in range(a.warm):
            e.step(acts[i % a.k])
        envs[name] = e
    out = torch.empty((a.k, N) + envs[a.variants[0].split("=")[0]].obs_shape, dtype=torch.float32, device=dev)
    print("obs buffer at 0x%x" % out.data_ptr())
    del out
    outs = {"shared": pre} if pre is not None else {}
    for rep in range(a.reps):
        for name, e in envs.items():
            e.timer_start()
            key = "shared" if a.share else name
            o = e.rollout(acts, keep_obs=not a.no_obs, out=outs.get(key) if (a.reuse or a.share) else None)
            outs[key] = o
            print(f"rep {rep} {name:10s} {e.timer_stop() * 1e3 / a.k:7.2f} us/step", flush=True)
    for e in envs.values():
        e.close()


if __name__ == "__main__":
    main()
