#!/usr/bin/env python3
"""Print the launch geometry the library picks at the headline shape for a few envs_per_wave
settings (0 = automatic) — incl. the resident workgroups per CU the runtime reports for the hot
kernel.  python tools/occupancy_probe.py [--lib path/to/libgte.so]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--envs", type=int, nargs="+", default=[65536, 32768, 81920, 131072])
    a = ap.parse_args()
    from gym_trading_env_amd.batched import BatchedTradingEnv
    wl = bench.WORKLOADS["c3"]
    feat, close = bench.synthetic_dataset(0, wl["T"], wl["n_static"])
    for n in a.envs:
        e = BatchedTradingEnv((feat, close), num_envs=n, library_path=a.lib, **bench.env_kwargs(wl))
        print(n, e.launch_info(), flush=True)
        e.close()


if __name__ == "__main__":
    main()
