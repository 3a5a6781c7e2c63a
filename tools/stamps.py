#!/usr/bin/env python3
"""Where a workgroup of the step kernel spends its time: runs the DIAGNOSTIC build
libgte_stamps.so (make -C gym-trading-env_amd/csrc libgte_stamps.so), whose step kernel
records s_memrealtime (100 MHz) at 8 points per workgroup, and prints the median timeline."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

LABELS = ["kernel entry", "env ids + own rings in LDS", "A: record + action arrived",
          "A: trade done, next price asked for", "A: state machine done",
          "A: record/ring/job stores done", "barrier passed", "gather done"]


def main():
    """python3 tools/stamps.py [workload=c3] [affinity_period=0 (auto) | -1 (identity order)]"""
    import torch
    from gym_trading_env_amd import _abi
    from gym_trading_env_amd.batched import BatchedTradingEnv
    lib_path = os.path.join(os.path.dirname(_abi.LIB_PATH), "libgte_stamps.so")
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    affinity = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    wl = bench.WORKLOADS[name]
    N = wl["envs"]
    D = wl["n_datasets"]
    data = [bench.synthetic_dataset(d, wl["T"], wl["n_static"]) for d in range(D)]
    env = BatchedTradingEnv(data if D > 1 else data[0], num_envs=N, seed=1, output="torch", library_path=lib_path,
                            kernel_variant=64, affinity_period=affinity, **bench.env_kwargs(wl))
    print(f"# workload {name}, {N} envs, affinity_period {affinity}")
    lib = env._lib
    lib.gte_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    blocks = env.launch_info()["n_blocks"]
    buf = torch.zeros((blocks, 8), dtype=torch.int64, device="cuda")
    acts = torch.randint(0, 3, (64, N), dtype=torch.int32, device="cuda")
    env.reset()
    if len(sys.argv) > 3 and sys.argv[3] == "desync":  # (round 2's timelines: episodes in phase)
        bench.desynchronise(env, acts, wl["max_episode_duration"])
    for i in range(200):
        env.step(acts[i % 64])
    assert lib.gte_debug_set_stamps(env._h, C.c_void_p(buf.data_ptr())) == 0
    rows = []
    for i in range(20):
        env.step(acts[i % 64])
        torch.cuda.synchronize()
        t = buf.cpu().numpy().astype(np.float64)
        rows.append((t - t[:, :1].min()) * 10.0)  # ns since the first workgroup entered
    t = np.stack(rows)  # [steps, blocks, 8]
    print(f"{blocks} workgroups, ns since the first workgroup's entry (median / p10 / p90 over workgroups and 20 steps)")
    for k, lab in enumerate(LABELS):
        v = t[:, :, k].ravel()
        print(f"  {k} {lab:42s} {np.median(v):8.0f} {np.percentile(v, 10):8.0f} {np.percentile(v, 90):8.0f}")
    d = np.diff(t, axis=2)
    print("segment durations (median ns):", [int(np.median(d[:, :, k])) for k in range(7)])
    env.close()


if __name__ == "__main__":
    main()
