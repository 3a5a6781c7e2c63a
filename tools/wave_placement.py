#!/usr/bin/env python3
"""Which SIMD runs phase A: the DIAGNOSTIC build libgte_stamps.so compiled with
-DGTE_STAMPS_HWID records HW_ID of wave 0 (the phase-A wave) of every workgroup of the step
kernel.  Prints how the phase-A waves of one CU spread over its 4 SIMDs."""
import collections
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    from gym_trading_env_amd import _abi
    from gym_trading_env_amd.batched import BatchedTradingEnv
    lib_path = os.path.join(os.path.dirname(_abi.LIB_PATH), sys.argv[1] if len(sys.argv) > 1 else "libgte_stamps.so")
    print(f"# {os.path.basename(lib_path)}")
    wl = bench.WORKLOADS["c3"]
    N = wl["envs"]
    feat, close = bench.synthetic_dataset(0, wl["T"], wl["n_static"])
    env = BatchedTradingEnv((feat, close), num_envs=N, seed=1, output="torch", library_path=lib_path,
                            kernel_variant=64, **bench.env_kwargs(wl))
    lib = env._lib
    lib.gte_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    blocks = env.launch_info()["n_blocks"]
    buf = torch.zeros((blocks, 8), dtype=torch.int64, device="cuda")
    acts = torch.randint(0, 3, (64, N), dtype=torch.int32, device="cuda")
    env.reset()
    for i in range(50):
        env.step(acts[i % 64])
    assert lib.gte_debug_set_stamps(env._h, C.c_void_p(buf.data_ptr())) == 0
    for rep in range(3):
        env.step(acts[rep])
        torch.cuda.synchronize()
        t = buf.cpu().numpy()
        hw = t[:, 1] & 0xFFFFFFFF
        xcc = (t[:, 1] >> 32) & 0xF
        simd = (hw >> 4) & 3
        cu = (hw >> 8) & 15
        sh = (hw >> 12) & 1
        se = (hw >> 13) & 7
        entry = (t[:, 0] - t[:, 0].min()) * 10.0
        per_cu = collections.defaultdict(list)
        for b in range(blocks):
            per_cu[(int(xcc[b]), int(se[b]), int(sh[b]), int(cu[b]))].append(int(simd[b]))
        worst = collections.Counter()
        for k, v in per_cu.items():
            worst[max(collections.Counter(v).values())] += 1
        print(f"step {rep}: {blocks} workgroups on {len(per_cu)} CUs; phase-A waves per SIMD overall "
              f"{np.bincount(simd, minlength=4).tolist()}; workgroups per CU "
              f"{sorted(collections.Counter(len(v) for v in per_cu.values()).items())}; "
              f"CUs by max phase-A waves on one SIMD {sorted(worst.items())}")
        if rep == 0:
            for k in sorted(per_cu)[:6]:
                print("   ", k, per_cu[k])
            print("    first 24 workgroups: xcc", xcc[:24].tolist(), "simd", simd[:24].tolist())
            print("    entry time (ns) of workgroup 0, 8, 256, 1024, last:",
                  [int(entry[i]) for i in (0, 8, 256, min(1024, blocks - 1), blocks - 1)])
    env.close()


if __name__ == "__main__":
    main()
