#!/usr/bin/env python3
"""One-off soak (not collected by pytest: no test_ prefix): the random-configuration parity test
of test_gpu_parity.py over many more seeds and, every fourth seed, a batch big enough to reach
the multi-workgroup launch geometries (up to 140 000 envs).
    python3 tests/soak_random_configurations.py [first_seed] [n_seeds]
Uses the oracle as the checker (tests/ may)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle  # noqa: E402
from test_gpu_parity import _compare_with_oracle, _synthetic  # noqa: E402


def one(seed):
    rng = np.random.default_rng(50_000 + seed)
    big = seed % 4 == 3
    T = int(rng.integers(80, 500)) if not big else int(rng.integers(2_000, 20_000))
    n_static = int(rng.integers(1, 40))
    if big and rng.random() < 0.6:
        n_static = int(rng.choice([2, 6, 14, 30]))  # F_obs % 4 == 0 with 2 dynamic features: the hot shape
    sigma = float(rng.choice([1e-3, 1e-2, 5e-2, 0.12]))
    ds = [_synthetic(7000 + seed, T, n_static, sigma=sigma, drift=-sigma / 4)]
    P = int(rng.integers(2, 9))
    positions = sorted(set(np.round(rng.uniform(-2.5, 3.5, P), 2).tolist() + [0.0]))
    windows = None if rng.random() < 0.25 else int(rng.integers(1, 30))
    first = 0 if windows is None else windows - 1
    room = T - 2 * first
    if room < 12:
        windows, first, room = 3, 2, T - 4
    max_dur = "max" if rng.random() < 0.3 else int(rng.integers(3, max(4, min(room // 2, 60))))
    nd = int(rng.integers(0, 5)) if not (big and n_static in (2, 6, 14, 30)) else 2
    kinds = [str(rng.choice(["last_position_taken", "real_position"])) for _ in range(nd)]
    n_envs = int(rng.integers(1, 700)) if not big else int(rng.choice([4_096, 16_384, 33_000, 65_536, 100_003, 140_000]))
    if big and windows is not None and n_envs * windows * (n_static + nd) * 4 > 1.2e9:
        n_envs = 16_384
    kw = dict(positions=positions, windows=windows, dynamic_feature_functions=kinds,
              trading_fees=float(rng.choice([0.0, 1e-4, 1e-3, 1e-2])),
              borrow_interest_rate=float(rng.choice([0.0, 3e-6, 1e-4, 1e-3])),
              portfolio_initial_value=float(rng.choice([1000.0, 1.0, 1e6])),
              initial_position="random" if rng.random() < 0.7 else positions[int(rng.integers(len(positions)))],
              max_episode_duration=max_dur,
              autoreset=[None, "next_step", "same_step"][int(rng.integers(3))],
              dyn_persist=bool(rng.random() < 0.2) and not big)
    t0 = time.time()
    n = _compare_with_oracle(oracle, ds, n_envs=n_envs, steps=24 if big else 40, seed=seed,
                             check_every=6 if big else 4, **kw)
    print(f"seed {seed:4d}: {n_envs:6d} envs, T {T:5d}, F_s {n_static:2d}, W {windows}, nd {nd}, "
          f"{kw['autoreset']}, dur {max_dur}: ok ({n} episode ends, {time.time() - t0:.1f} s)", flush=True)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    oracle.build()
    for seed in range(first, first + count):
        one(seed)
    print("soak passed:", count, "configurations")


if __name__ == "__main__":
    main()
