#!/usr/bin/env python3
"""One-off soak (not collected by pytest): test_rollout_random_shapes_equal_single_steps over many
more seeds, every fourth one with a batch of tens of thousands of envs (work-queue launch with
several groups per workgroup, both store policies).
    python3 tests/soak_rollouts.py [first_seed] [n_seeds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_rollout import _check, _data, _twins  # noqa: E402


def one(seed):
    import torch
    rng = np.random.default_rng(90_000 + seed)
    big = seed % 4 == 3
    nd = int(rng.integers(1, 5))
    fv = int(rng.integers(1, 41)) if not big else int(rng.integers(1, 12))  # vectors per row
    Fs = 4 * fv - nd
    if Fs < 0:
        fv += 1
        Fs = 4 * fv - nd
    W = int(rng.choice([2, 3, 5, 8, 13, 20, 40])) if not big else int(rng.choice([2, 5, 8, 20]))
    D = int(rng.choice([1, 1, 3]))
    T = int(rng.integers(3 * W + 40, 3 * W + 400)) if not big else int(rng.integers(2000, 9000))
    sigma = float(rng.choice([2e-3, 2e-2, 6e-2]))
    data = [_data(19000 + 10 * seed + d, T + 7 * d, Fs, sigma=sigma)[:2] for d in range(D)]
    kinds = [str(rng.choice(["last_position_taken", "real_position"])) for _ in range(nd)]
    autoreset = [None, "next_step", "same_step"][int(rng.integers(3))]
    N = int(rng.integers(1, 2500)) if not big else int(rng.choice([9_001, 20_000, 40_000, 70_001]))
    kw = dict(positions=sorted(set(np.round(rng.uniform(-2, 3, 4), 1).tolist() + [0.0])), windows=W,
              dynamic_feature_functions=kinds, trading_fees=float(rng.choice([0, 1e-4, 1e-2])),
              borrow_interest_rate=float(rng.choice([0, 3e-6, 1e-3])),
              max_episode_duration=int(rng.integers(4, 30)), autoreset=autoreset,
              episodes_between_dataset_switch=int(rng.integers(1, 3)), seed=seed)
    t0 = time.time()
    a, b = _twins(data if D > 1 else data[0], N, **kw)
    P = len(kw["positions"])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    kmax = 30 if not big else 10
    ends = 0
    for K, keep in ((int(rng.integers(2, kmax)), True), (int(rng.integers(2, kmax)), False), (1, True)):
        acts = torch.randint(-1, P, (K, N), dtype=torch.int32, device="cuda", generator=gen)
        ends += _check(a, b, acts, keep, f"seed {seed} W={W} Fobs={Fs + nd} nd={nd} N={N} K={K} {autoreset}")
        one_ = torch.randint(-1, P, (N,), dtype=torch.int32, device="cuda", generator=gen)
        for x, y in zip(a.step(one_)[:4], b.step(one_)[:4]):
            np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy())
    a.close()
    b.close()
    print(f"seed {seed:4d}: N {N:6d}, W {W:2d}, F_obs {Fs + nd:3d}, nd {nd}, D {D}, {autoreset}: ok "
          f"({ends} flags raised, {time.time() - t0:.1f} s)", flush=True)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    for seed in range(first, first + count):
        one(seed)
    print("rollout soak passed:", count, "shapes")


if __name__ == "__main__":
    main()
