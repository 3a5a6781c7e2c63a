#!/usr/bin/env python3
"""Step time against the ADDRESS of the observation buffer, everything else unchanged: one env,
its observation output re-bound (gte_bind_outputs) to different offsets inside one arena whose
base is 256 MiB aligned.  python3 tools/obs_alignment_probe.py [c3|c5] [envs]"""
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
    name = sys.argv[1] if len(sys.argv) > 1 else "c5"
    wl = dict(bench.WORKLOADS[name])
    envs = int(sys.argv[2]) if len(sys.argv) > 2 else wl["envs"]
    D = wl.get("n_datasets", 1)
    data = [bench.synthetic_dataset(d, wl["T"], wl["n_static"]) for d in range(D)]
    acts = torch.randint(0, 3, (64, envs), dtype=torch.int32, device="cuda")
    env = BatchedTradingEnv(data if D > 1 else data[0], num_envs=envs, seed=1, output="torch",
                            **bench.env_kwargs(wl))
    env.reset()
    bench.desynchronise(env, acts, wl["max_episode_duration"])
    nbytes = env._t["obs"].numel() * 4
    A = 256 << 20
    arena = torch.zeros(nbytes + 2 * A, dtype=torch.uint8, device="cuda")
    base = (-arena.data_ptr()) % A
    print(f"{name} {envs} envs, obs {nbytes / 1e6:.0f} MB; own buffer at {env._t['obs'].data_ptr():#x}")
    MB = 1 << 20
    offsets = [0, 4096, 64 * 1024, 1 * MB, 2 * MB, 6 * MB, 14 * MB, 16 * MB, 32 * MB, 64 * MB, 96 * MB, 128 * MB,
               130 * MB, 0]
    for off in offsets:
        view = arena[base + off: base + off + nbytes].view(torch.float32).view(env._t["obs"].shape)
        env._t["obs"] = view
        b = _abi.GteOutputs()
        for k, t in env._t.items():
            setattr(b, k, t.data_ptr())
        _abi.check(env._lib, env._lib.gte_bind_outputs(env._h, C.byref(b)))
        for i in range(100):
            env.step(acts[i % 64])
        torch.cuda.synchronize()
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            for i in range(500):
                env.step(acts[i % 64])
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 500 * 1e6)
        print(f"  obs at 256 MiB boundary + {off / MB:9.4f} MiB ({view.data_ptr():#x}) -> "
              + "  ".join(f"{x:6.2f}" for x in ts) + " us/step", flush=True)
    env.close()


if __name__ == "__main__":
    main()
