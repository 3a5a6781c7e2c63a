#!/usr/bin/env python3
"""Host time of `episode_metrics()` with one custom metric at the headline shape (65 536 envs,
episodes out of phase: ~131 envs finish per step), and of its parts: the ONE transfer of the
finished envs' episodes (`gte_read_log_envs`), building the History objects, the user's metric.
python3 tools/metrics_bench.py [envs] [log_steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    from gym_trading_env_amd.batched import BatchedTradingEnv
    envs = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    wl = dict(bench.WORKLOADS["c3"], envs=envs)
    feat, close = bench.synthetic_dataset(0, wl["T"], wl["n_static"])
    acts = torch.randint(0, 3, (64, envs), dtype=torch.int32, device="cuda")
    env = BatchedTradingEnv((feat, close), num_envs=envs, seed=1, output="torch", log_steps=L,
                            **bench.env_kwargs(wl))
    env.add_metric("Position Changes", lambda h: int(np.sum(np.diff(h["position"]) != 0)))
    env.reset()
    bench.desynchronise(env, acts, wl["max_episode_duration"])
    for i in range(wl["max_episode_duration"] + 20):  # every env has a whole episode in the log
        env.step(acts[i % 64])
    t_read, t_hist, t_all, n_fin, rows = [], [], [], [], []
    for i in range(30):
        env.step(acts[i % 64])
        torch.cuda.synchronize()
        ids = env.terminal_ids()
        t0 = time.perf_counter()
        b = env.read_log_envs(ids)
        t1 = time.perf_counter()
        hs = env.histories(ids)
        t2 = time.perf_counter()
        m = env.episode_metrics()
        t3 = time.perf_counter()
        assert len(m["Position Changes"]) == len(ids) == len(hs)
        t_read.append(t1 - t0); t_hist.append(t2 - t1); t_all.append(t3 - t2)
        n_fin.append(len(ids)); rows.append(int(b["n_rows"].sum()))
    ms = lambda v: f"{1e3 * float(np.median(v)):7.3f} ms (min {1e3 * min(v):.3f})"
    print(f"# {envs} envs, log_steps {L}: {np.mean(n_fin):.0f} envs finish per step, {np.mean(rows):.0f} logged rows "
          f"of theirs per step ({np.mean(rows) * 73 / 1e6:.2f} MB packed)")
    print(f"gte_read_log_envs (one kernel + one transfer, all 12 columns)   {ms(t_read)}")
    print(f"histories(): that + {np.mean(n_fin):.0f} History objects (24 columns each)       {ms(t_hist)}")
    print(f"episode_metrics() with one custom metric (read + build + metric) {ms(t_all)}")
    env.close()


if __name__ == "__main__":
    main()
