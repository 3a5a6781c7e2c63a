#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched step() hot path on MI355X.

One "step" = one gte_step launch over every environment of the rank's shard
(TradingEnv.step for N envs, reference environments.py:233-272), observations
left on the device.  Workload (BASELINE.json configs[2]): 65 536 envs
per GPU, synthetic random-walk OHLCV T = 100 000, 30 static + 2 dynamic features,
window 20 (obs 20x32 f32), positions [-1, 0, 1], fees 1e-4, borrow interest 3e-6
(margin path live), random starts with max_episode_duration = 500, next-step
auto-reset, uniform random actions pre-generated on the device.

Prints ONE JSON line (rank 0).  `roofline` prices the step kernel against the
8 TB/s HBM peak with the ALGORITHMIC bytes of SURVEY §8d (5 250 B per env-step at
this shape); `cpu_baseline` is the oracle's C restatement (oracle/, a "port")
timed on this box's host cores on a bounded sample of the same workload.

Multi-GPU (torchrun, one rank per GPU): envs shard with no data-path collective
except the RCCL all-gather of the returns (reward, terminated, truncated: 6 bytes
per env and step), which is what the north star names.  Every step's returns cross
xGMI inside the timed region, a --gather-every (32) step block at a time on RCCL's
stream while the following steps run (DESIGN.md §6 has the per-step alternatives
measured); --gather-obs adds the observation all-gather (xGMI-bound, SURVEY §7
hard part 7).  Weak scaling: 65 536 envs/GPU.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # BASELINE.json configs[2] (headline), [1] and the per-GPU share of [4]
    "c3": dict(n_static=30, windows=20, T=100_000, max_episode_duration=500, envs=65_536,
               n_datasets=1),
    "c2": dict(n_static=14, windows=None, T=100_000, max_episode_duration=500, envs=4_096,
               n_datasets=1),
    # config 5: 1 024 symbols x 256 envs over 8 GPUs = 128 resident datasets and 32 768
    # envs per GPU, per-env dataset indirection, switch at every episode
    "c5": dict(n_static=30, windows=20, T=100_000, max_episode_duration=500, envs=32_768,
               n_datasets=128),
}


def synthetic_dataset(dataset_id: int, T: int, n_static: int):
    """close = 100*exp(cumsum(N(0, 1e-3))), features N(0,1) f32."""
    rng = np.random.default_rng(1234 + dataset_id)
    close = 100.0 * np.exp(np.cumsum(rng.normal(0.0, 1e-3, T)))
    feat = rng.normal(0.0, 1.0, (T, n_static)).astype(np.float32)
    return feat, close


def algorithmic_bytes(W: int, F_obs: int, F_s: int, n_dyn: int) -> int:
    """SURVEY §8d B_alg per env-step."""
    return 4 * W * F_obs + 4 * W * F_s + 8 + 104 + 4 * W * n_dyn + 4 * n_dyn + 4 + 6


def env_kwargs(wl):
    return dict(positions=[-1, 0, 1], windows=wl["windows"], trading_fees=1e-4,
                borrow_interest_rate=3e-6, portfolio_initial_value=1000,
                initial_position="random", max_episode_duration=wl["max_episode_duration"],
                autoreset="next_step")


def host_cores() -> int:
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota
    (a 1-GPU box exposes all host CPUs in the mask but only a 16-core share)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return int(os.environ.get("GTE_BENCH_CORES", min(n, 16)))


def cpu_baseline(wl, seconds_target: float = 12.0):
    """Time the ORACLE (oracle/gte_oracle.c, OpenMP over envs) on the host cores on a
    bounded sample of the same workload.  Reported baseline only; never the product."""
    from gym_trading_env_amd.config import make_config
    from oracle import oracle
    oracle.build()
    cores = host_cores()
    n_envs, n_dyn = 65_536, 2
    feat, close = synthetic_dataset(0, wl["T"], wl["n_static"])
    full = np.zeros((wl["T"], wl["n_static"] + n_dyn), np.float32)
    full[:, :wl["n_static"]] = feat
    cfg = make_config(n_envs=n_envs, n_static=wl["n_static"], seed=7, **env_kwargs(wl))
    env = oracle.OracleEnv(cfg, [(full, close)])
    env.reset()
    rng = np.random.default_rng(99)
    acts = rng.integers(0, 3, (32, n_envs)).astype(np.int32)
    for i in range(3):
        env.step(acts[i], threads=cores)
    steps, t0 = 0, time.perf_counter()
    while True:
        env.step(acts[steps % 32], threads=cores)
        steps += 1
        el = time.perf_counter() - t0
        if el > seconds_target or steps >= 20000:
            break
    env.close()
    return {"value": n_envs * steps / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n_envs} envs x {steps} steps of the same workload, oracle C restatement "
                      f"with OpenMP on {cores} threads, {el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=0, help="envs per GPU (default: the workload's)")
    ap.add_argument("--epw", type=int, default=0, help="envs per wavefront (0 = auto)")
    ap.add_argument("--nt", type=int, default=-1,
                    help="observation store policy: 0 plain, 1 nt, 2 sc1 (default: the library's)")
    ap.add_argument("--variant", type=int, default=0, help="kernel_variant bits (A/B timing)")
    ap.add_argument("--affinity", type=int, default=0,
                    help="L2-affinity re-sort period in steps (0 = default 128, -1 = off)")
    ap.add_argument("--gather-obs", action="store_true", help="also all-gather observations (N>1)")
    ap.add_argument("--gather-every", type=int, default=32,
                    help="N>1: all-gather the returns in blocks of this many steps (every step's "
                         "reward/flags still cross xGMI inside the timed region; 1 = per step)")
    ap.add_argument("--gather-depth", type=int, default=2,
                    help="N>1: blocks in rotation (>=2: a block's all-gather overlaps the steps "
                         "that fill the next one on RCCL's stream; 1 = synchronous per-step gather)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a
    # version banner at communicator creation), so everything but that line goes to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the env has no CPU fallback")
    # rehearsal knobs (a 1-GPU box can run 2 ranks on its one device over gloo; RCCL refuses
    # two ranks per device): GTE_BENCH_BACKEND=gloo GTE_BENCH_SHARE_DEVICE=1
    backend = os.environ.get("GTE_BENCH_BACKEND", "nccl")
    if os.environ.get("GTE_BENCH_SHARE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # GTE_BENCH_FORCE_DIST=1: rehearse the N>1 code path with a 1-rank group on one GPU
    force_dist = world == 1 and bool(os.environ.get("GTE_BENCH_FORCE_DIST"))
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
    use_dist = world > 1 or force_dist
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)  # RCCL over xGMI
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from gym_trading_env_amd.batched import BatchedTradingEnv
    depth = max(1, args.gather_depth)
    block = max(1, args.gather_every) if depth > 1 else 1

    wl = WORKLOADS[args.workload]
    N = args.envs or wl["envs"]
    n_dyn = 2
    W = wl["windows"] or 1
    F_obs = wl["n_static"] + n_dyn
    D = wl["n_datasets"]
    # configs 2-4: ONE dataset replicated on every rank; config 5: symbols partitioned by rank
    data = [synthetic_dataset((rank * D + d) if D > 1 else 0, wl["T"], wl["n_static"])
            for d in range(D)]
    tuning = dict(envs_per_wave=args.epw, kernel_variant=args.variant,
                  affinity_period=args.affinity)
    if args.nt >= 0:
        tuning["nontemporal_obs"] = args.nt
    env = BatchedTradingEnv(data if D > 1 else data[0], num_envs=N, seed=20240607,
                            env_id_base=rank * N, device=local_rank, output="torch",
                            return_slots=depth * block if use_dist else 1,
                            **tuning, **env_kwargs(wl))
    gen = torch.Generator(device=dev)
    gen.manual_seed(99 + rank)
    n_rows = 64
    actions = torch.randint(0, 3, (n_rows, N), dtype=torch.int32, device=dev, generator=gen)
    env.reset()

    # the return of a sharded run: RCCL all-gather of the packed (reward f32 | terminated u8 |
    # truncated u8) records, 6 bytes per env and step, which the kernel writes directly in
    # that layout; by default a 32-step block at a time, overlapping the following steps
    returns = pipe = None
    if use_dist:
        from gym_trading_env_amd.distributed import ReturnGather, ReturnPipeline
        returns = ReturnGather(N, dev, obs_shape=env.obs_shape if args.gather_obs else None,
                               depth=depth, block=block)
        if depth > 1:
            pipe = ReturnPipeline(env, returns, block, depth)

    def drain():
        if pipe is not None:
            pipe.flush()  # incl. a block the step count left unfinished

    def one_step(i):
        if pipe is not None:
            pipe.before_step()  # rows about to be rewritten must have been gathered
        obs, reward, term, trunc, _ = env.step(actions[i % n_rows])
        if pipe is not None:
            pipe.after_step()   # starts the block's all-gather on block boundaries
        elif returns is not None:
            returns.gather(env.packed_returns)
        if returns is not None and args.gather_obs:
            returns.gather_obs(obs)

    for i in range(args.warmup):
        one_step(i)
    drain()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    env.timer_start()  # HIP events on the stream the kernel is launched on
    for i in range(args.steps):
        one_step(args.warmup + i)
    drain()  # every all-gather belongs to the timed region
    ev_ms = env.timer_stop()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el, ev_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el, ev_ms = float(t[0]), float(t[1])

    # episodes really end inside the timed region (auto-reset is part of the step)
    episodes = int(env.state("episode").sum()) - N  # beyond the initial reset
    info = env.launch_info()
    b_alg = algorithmic_bytes(W, F_obs, wl["n_static"], n_dyn)
    kernel_us = ev_ms * 1e3 / args.steps
    achieved = b_alg * N / (kernel_us * 1e-6) / 1e9
    traffic = None
    tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tp) and N == wl["envs"]:  # the counters were collected at the default size
        try:
            traffic = json.load(open(tp)).get(f"{args.workload}_bytes_per_launch")
        except Exception:
            traffic = None
    if rank == 0:
        out = {
            "metric": "env-steps/sec", "value": world * N * args.steps / el, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {N} envs/GPU x obs ({W},{F_obs}) f32, "
                                   f"{D} dataset(s)/GPU of T={wl['T']}, positions [-1,0,1], fees 1e-4, borrow 3e-6, "
                                   f"max_episode_duration {wl['max_episode_duration']}, next-step autoreset",
                       "envs_per_gpu": N, "global_envs": world * N,
                       "parallelism": f"env-shard x{world}" + (
                           "" if world == 1 else " + RCCL all-gather(reward,flags"
                           + (",obs)" if args.gather_obs else ")")
                           + (f" in {block}-step blocks overlapping the following steps"
                              if depth > 1 else ", synchronous per step")),
                       "launch": info,
                       "episodes_finished": episodes},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_env_step": b_alg,
                         "kernel_us_per_launch": kernel_us},
        }
        if traffic:
            # `achieved` prices ALGORITHMIC bytes (SURVEY §8d credits no reuse of the feature
            # table between envs or steps), so with the table served from L2 it can pass the
            # HBM peak; the counter-measured bytes per launch over the same kernel time:
            out["roofline"]["traffic_rate"] = traffic / (kernel_us * 1e-6) / 1e9
            out["roofline"]["traffic_frac"] = out["roofline"]["traffic_rate"] / HBM_PEAK_GBS
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl)
            # the reference's OWN Python step() cannot run on this box (its files do not
            # travel); quote the timing taken in the build container (tools/time_reference.py)
            rp = os.path.join(ROOT, "profiles", "reference_cpu_timing.json")
            if os.path.exists(rp):
                try:
                    ref = json.load(open(rp))
                    rpy = {"where": ref["where"], "cpu": ref["cpu"], "cores": ref["cores"],
                           "env_steps_per_s": ref["shapes"].get(args.workload)}
                    # the build's own interpreter-bound per-env loop (oracle/py_loop.py) was timed
                    # there next to the reference and is timed here now, on one core: the ratio
                    # rescales the reference's own figure to this box (an estimate, labelled so)
                    here = ref.get("python_loop_c3_steps_per_s_1_process")
                    one = (ref["shapes"].get(args.workload) or {}).get("steps_per_s_1_process")
                    if here and one:
                        from oracle.py_loop import time_loop
                        rate, n, el = time_loop(seconds=2.0)
                        cores = out["cpu_baseline"]["cores"]
                        rpy["python_loop"] = {"env_steps_per_s_this_box_1_core": rate,
                                              "env_steps_per_s_there_1_core": here,
                                              "sample": f"{n} steps, {el:.1f} s, config-3-shaped env"}
                        rpy["estimated_on_this_box"] = {
                            "env_steps_per_s_1_core": one * rate / here, "cores": cores,
                            "env_steps_per_s_all_cores": one * rate / here * cores,
                            "how": "reference timing there x (python_loop here / python_loop there), "
                                   "one process per core"}
                    out["cpu_baseline"]["reference_python"] = rpy
                except Exception:
                    pass
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    env.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
