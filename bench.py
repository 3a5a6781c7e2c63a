#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched step() hot path on MI355X.

One "step" = one gte_step launch over every environment of the rank's shard
(TradingEnv.step for N envs, reference environments.py:233-272), observations
left on the device.  Workload (BASELINE.json configs[2]): 65 536 envs
per GPU, synthetic random-walk OHLCV T = 100 000, 30 static + 2 dynamic features,
window 20 (obs 20x32 f32), positions [-1, 0, 1], fees 1e-4, borrow interest 3e-6
(margin path live), random starts with max_episode_duration = 500, next-step
auto-reset, uniform random actions pre-generated on the device.

Episodes are DE-SYNCHRONISED before anything is timed (an untimed prologue resets
env group e % 500 at prologue step e % 500), so every step of every window —
the driver's 20-step one included — carries its share of episode ends and
auto-resets (about N/500 per step) instead of a reset storm every 500 steps.

Prints ONE JSON line (rank 0).  `roofline`:
  achieved   ALGORITHMIC bytes of SURVEY §8d (5 250 B per env-step at this shape, no reuse
             of the feature table credited) / kernel time — kept as the survey defines it;
             with the table served from L2 it exceeds what the memory system really moves;
  traffic    bytes per launch from the PMC counters (FETCH_SIZE x 1024 x 2 — the gfx950
             half-count of wide reads — plus WRITE_SIZE x 1024), collected by THIS run in
             separate `rocprofv3 --pmc` child passes of the same workload (falls back to
             profiles/hbm_traffic.json, and says so, when rocprofv3 is unavailable);
  frac       traffic / kernel time / 8 TB/s — the headline fraction, physical (<= 1);
  frac_compulsory   the bytes no implementation can avoid (SURVEY §8d lower-bound
             variant: obs store + one new table row + state + ring + returns = 2 970 B
             per env-step) / kernel time / 8 TB/s;
  regime     where the observation buffer lives: at 65 536 envs its 168 MB fit the 256 MiB
             Infinity Cache (the counters then measure fabric traffic, not DRAM traffic);
  hbm_regime the same kernel at 262 144 envs (671 MB of observations per launch, streamed
             to HBM with non-temporal stores), with its own counters.
`cpu_baseline` is the oracle's C restatement (oracle/, a "port") timed on this box's
host cores on a bounded sample of the same workload.

Multi-GPU (torchrun, one rank per GPU): envs shard with no data-path collective
except the RCCL all-gather of the returns (reward, terminated, truncated: 6 bytes
per env and step), which is what the north star names.  `value` is measured with per-step
returns through the C ABI's own RCCL communicator (gte_allgather_returns): synchronously after
every step (`--gather-mode step`) or overlapping the next step (`step-overlap`), whichever a short
probe finds faster on the node (`auto`, the default); `block` moves a --gather-every (32) step
block at a time on RCCL's stream while the following steps run and is reported
beside it (config.other_gather_modes);
--gather-obs adds the observation all-gather (xGMI-bound).  Weak scaling: 65 536 envs/GPU.

`python3 bench.py --gpus N` from a bare shell starts its own N ranks (fresh child processes under
torch.distributed.run, before this process touches the GPU) and relays rank 0's line; under an
external torchrun it runs as one rank.  Workloads: c3 (headline), c2, c4 (BASELINE config 4:
32 768 envs per GPU, returns AND observations all-gathered; also reported without the observation
gather), c5 (128 symbols per GPU, partitioned by rank).
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
INFINITY_CACHE_BYTES = 256 << 20
HBM_REGIME_ENVS = 262_144  # 671 MB of observations per launch: far past the Infinity Cache

WORKLOADS = {
    # BASELINE.json configs[2] (headline), [1] and the per-GPU share of [4]
    "c3": dict(n_static=30, windows=20, T=100_000, max_episode_duration=500, envs=65_536,
               n_datasets=1),
    "c2": dict(n_static=14, windows=None, T=100_000, max_episode_duration=500, envs=4_096,
               n_datasets=1),
    # config 4: 262 144 envs sharded over 8 GPUs = 32 768 envs per GPU (kept per GPU at any --gpus:
    # weak scaling), the single dataset replicated, RCCL all-gather of the returns AND of the
    # observations (what BASELINE config 4 names; --no-gather-obs measures it without)
    "c4": dict(n_static=30, windows=20, T=100_000, max_episode_duration=500, envs=32_768,
               n_datasets=1, gather_obs=True),
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
    """SURVEY §8d B_alg per env-step (no reuse of the feature table credited)."""
    return 4 * W * F_obs + 4 * W * F_s + 8 + 104 + 4 * W * n_dyn + 4 * n_dyn + 4 + 6


def compulsory_bytes(W: int, F_obs: int, F_s: int, n_dyn: int) -> int:
    """SURVEY §8d lower-bound variant: perfect window reuse, i.e. the observation store, ONE
    new table row, the price, the state record in and out, the dynamic ring and the returns."""
    return 4 * W * F_obs + 4 * F_s + 8 + 104 + 4 * W * n_dyn + 4 * n_dyn + 4 + 6


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


# ---------------------------------------------------------------------------------------
# the workload, shared by the timed run, the HBM-regime leg and the PMC child passes

def make_env(args, wl, N, rank, local_rank, return_slots=1):
    from gym_trading_env_amd.batched import BatchedTradingEnv
    D = wl["n_datasets"]
    # configs 2-4: ONE dataset replicated on every rank; config 5: symbols partitioned by rank
    data = [synthetic_dataset((rank * D + d) if D > 1 else 0, wl["T"], wl["n_static"])
            for d in range(D)]
    tuning = dict(envs_per_wave=args.epw, kernel_variant=args.variant,
                  affinity_period=args.affinity, debug_flags=getattr(args, "debug_flags", 0))
    if args.nt >= 0:
        tuning["nontemporal_obs"] = args.nt
    return BatchedTradingEnv(data if D > 1 else data[0], num_envs=N, seed=20240607,
                             env_id_base=rank * N, device=local_rank, output="torch",
                             return_slots=return_slots, **tuning, **env_kwargs(wl))


def desynchronise(env, actions, cycle: int):
    """Untimed prologue: spread the episode phases evenly.  A next-step auto-reset env repeats
    every `cycle` = max_episode_duration step() calls (duration-1 steps + the reset call); group
    g = env % cycle is reset (TradingEnv.reset on those envs, a masked gte_reset) right before
    prologue step g, so afterwards about N/cycle envs end — and as many auto-reset — in EVERY
    step.  Only reference-semantics calls are used."""
    N = env.num_envs
    group = np.arange(N) % cycle
    n_rows = actions.shape[0]
    for g in range(cycle):
        if g:  # group 0 keeps the phase of the initial reset()
            env.reset(mask=(group == g).astype(np.uint8))
        env._launch_step(actions[g % n_rows])


def run_steps(env, actions, first, count):
    n_rows = actions.shape[0]
    for i in range(first, first + count):
        env.step(actions[i % n_rows])


def pmc_traffic(args, N, timeout_s=150):
    """Bytes per step-kernel launch from the hardware counters, collected NOW: one
    `rocprofv3 --pmc <counter>` child pass per counter (FETCH_SIZE and WRITE_SIZE need
    separate passes; no trace domain next to --pmc), each running `python3 bench.py
    --pmc-child` = this very workload, de-synchronisation included.  -> dict or None."""
    rocprof = shutil.which("rocprofv3")
    if rocprof is None:
        return None
    out = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="gte_pmc_", dir="/tmp")
        cmd = [rocprof, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
               sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", args.workload,
               "--envs", str(N), "--steps", "12", "--warmup", "4", "--epw", str(args.epw),
               "--nt", str(args.nt), "--variant", str(args.variant), "--affinity", str(args.affinity)]
        if args.sync_episodes:
            cmd.append("--sync-episodes")
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
            vals, name = [], None
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == counter and "gte_kernel<0" in r["Kernel_Name"]:
                        vals.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
                        name = r["Kernel_Name"].split("(")[0].strip()
            vals = [v for _, v in sorted(vals)][-12:]  # the post-warm-up launches
            if not vals:
                return None
            out[counter] = sum(vals) / len(vals)
            out["kernel"] = name
        except Exception as e:  # noqa: BLE001 - any failure means "no live counters"
            print(f"[bench] live PMC pass {counter} failed: {e!r}", file=sys.stderr)
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch = out["FETCH_SIZE"] * 1024 * 2  # KB -> B; gfx950 tallies 128-B read requests at 64 B
    write = out["WRITE_SIZE"] * 1024
    return {"bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
            "kernel": out["kernel"],
            "source": "live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of this "
                      "workload in this bench run (mean of the last 12 step-kernel dispatches; "
                      "KB x1024, FETCH_SIZE x2 per MI355X_MICROARCH.md §HBM)"}


def recorded_traffic(workload, N):
    """Fallback: the counters committed under profiles/ for this workload and size."""
    tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        rec = json.load(open(tp)).get(f"{workload}_{N}")
        if rec:
            return {"bytes_per_launch": rec["bytes_per_launch"], "fetch_bytes": rec["fetch_bytes"],
                    "write_bytes": rec["write_bytes"], "kernel": rec.get("kernel"),
                    "source": f"recorded: profiles/hbm_traffic.json[{workload}_{N}] "
                              f"({rec.get('round', '?')}), not collected by this run"}
    except Exception:
        pass
    return None


def roofline_block(N, kernel_us, shape, traffic):
    """`achieved` is a PHYSICAL rate: the PMC traffic of one launch / its duration when counters
    exist, else the compulsory bytes / duration (`achieved_is` says which); `frac` = achieved /
    peak always.  The SURVEY §8d algorithmic figure (no reuse of the table credited: with the
    table served from L2 it exceeds what the memory system moves, and the peak) is kept as
    `achieved_algorithmic` / `frac_algorithmic` only.  `frac_compulsory` — useful bytes only — and
    `traffic_over_compulsory` — wasted re-reads — are reported for every workload: a kernel that
    moves 1.9x its compulsory bytes at a high rate (config 5) shows a high `frac` and a low
    `frac_compulsory`."""
    W, F_obs, F_s, n_dyn = shape
    b_alg, b_min = algorithmic_bytes(*shape), compulsory_bytes(*shape)
    secs = kernel_us * 1e-6
    alg_rate = b_alg * N / secs / 1e9
    min_rate = b_min * N / secs / 1e9
    obs_bytes = 4 * W * F_obs * N
    in_mall = obs_bytes <= (190 << 20)
    r = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": None, "traffic": None,
         "frac_compulsory": min_rate / HBM_PEAK_GBS, "traffic_over_compulsory": None,
         "achieved_is": None,
         "achieved_algorithmic": alg_rate, "frac_algorithmic": alg_rate / HBM_PEAK_GBS,
         "algorithmic_bytes_per_env_step": b_alg, "compulsory_bytes_per_env_step": b_min,
         "kernel_us_per_launch": kernel_us,
         "regime": ("infinity-cache: %.0f MB of observations per launch stay in the 256 MiB MALL "
                    "(sc1 stores); FETCH/WRITE_SIZE count MALL hits, so `achieved` is a FABRIC "
                    "(Infinity Cache) rate measured against the HBM peak, not a DRAM rate"
                    % (obs_bytes / 1e6) if in_mall else
                    "hbm: %.0f MB of observations per launch stream to DRAM (non-temporal stores); "
                    "`achieved` is a DRAM rate against the HBM peak" % (obs_bytes / 1e6))}
    if traffic:
        rate = traffic["bytes_per_launch"] / secs / 1e9
        r.update(achieved=rate, frac=rate / HBM_PEAK_GBS, traffic=traffic["bytes_per_launch"],
                 achieved_is="PMC traffic (FETCH_SIZE x2 + WRITE_SIZE) per launch / kernel time",
                 traffic_fetch_bytes=traffic["fetch_bytes"], traffic_write_bytes=traffic["write_bytes"],
                 traffic_over_compulsory=traffic["bytes_per_launch"] / (b_min * N),
                 traffic_source=traffic["source"], kernel=traffic.get("kernel"))
    else:  # no counters: the compulsory bytes are the only physical count available
        r.update(achieved=min_rate, frac=min_rate / HBM_PEAK_GBS,
                 achieved_is="compulsory bytes per launch / kernel time (no PMC counters available)")
    return r


# ---------------------------------------------------------------------------------------
# --gpus N from a bare shell: start the ranks as fresh children

def self_launch(argv, n_gpus):
    """`python3 bench.py --gpus N` without a launcher around it (WORLD_SIZE unset): start the N
    ranks as CHILD processes — `python -m torch.distributed.run --nproc-per-node N bench.py ...`,
    before this process has imported torch or touched the GPU (never an exec) — relay rank 0's
    ONE JSON line, and exit non-zero if any rank failed.  Under an external torchrun (the
    driver's way) WORLD_SIZE is set and this is not used."""
    import socket
    with socket.socket() as so:  # a free rendezvous port
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL across processes: dmabuf IPC only
    print(f"[bench] --gpus {n_gpus} without a launcher: starting {' '.join(cmd)}", file=sys.stderr)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:  # everything but the JSON line goes to stderr, like in a rank
        if out.lstrip().startswith('{"metric"'):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc != 0 or line is None:
        print(f"[bench] the {n_gpus}-rank run failed (exit code {rc}, JSON line "
              f"{'missing' if line is None else 'present'})", file=sys.stderr)
        raise SystemExit(rc or 1)
    print(line, flush=True)
    raise SystemExit(0)


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
    ap.add_argument("--debug-flags", type=int, default=0, help="gte_config.debug_flags (timing probes: results are wrong)")
    ap.add_argument("--affinity", type=int, default=0,
                    help="L2-affinity re-sort period in steps (0 = default 128, -1 = off)")
    ap.add_argument("--sync-episodes", action="store_true",
                    help="skip the de-synchronising prologue: all envs start together, resets "
                         "arrive as a storm every max_episode_duration steps (round-1 behaviour)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not spawn the rocprofv3 --pmc child passes (traffic then comes from "
                         "profiles/hbm_traffic.json, or is null)")
    ap.add_argument("--no-hbm-regime", action="store_true",
                    help="skip the 262 144-env leg (observations streamed to HBM)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--gather-obs", action="store_true",
                    help="also all-gather the observations after every step (N>1; xGMI-bound: 2 560 B "
                         "per env).  Workload c4 does by default — BASELINE config 4 names it — and "
                         "also reports the same run WITHOUT it (config.without_obs_gather)")
    ap.add_argument("--no-gather-obs", action="store_true",
                    help="c4: measure `value` without the observation all-gather (the run with it is "
                         "then reported under config.with_obs_gather)")
    ap.add_argument("--gather-mode", default="auto",
                    choices=["auto", "step", "step-overlap", "block", "torch-step"],
                    help="N>1, the mode `value` is measured in: 'step' = synchronous per-step "
                         "all-gather through libgte's own RCCL communicator (gte_allgather_returns, "
                         "what the north star describes); 'step-overlap' = the same on the library's "
                         "communication stream, overlapping the next step (returns one step late); "
                         "'block' = asynchronous all-gather of --gather-every step blocks "
                         "(torch.distributed); 'torch-step' = synchronous per-step through "
                         "torch.distributed; 'auto' (default) = the faster of the two PER-STEP forms "
                         "'step' and 'step-overlap', decided by a short untimed probe on this node "
                         "(synchronous gathers pay RCCL's cross-GPU latency in every step, overlapped "
                         "ones two cross-stream events).  The other modes are measured too and "
                         "reported under config.other_gather_modes")
    ap.add_argument("--one-gather-mode", action="store_true",
                    help="N>1: measure only --gather-mode")
    ap.add_argument("--gather-every", type=int, default=32,
                    help="block mode: all-gather the returns in blocks of this many steps (every "
                         "step's reward/flags still cross xGMI inside the timed region)")
    ap.add_argument("--gather-depth", type=int, default=2,
                    help="block mode: blocks in rotation (a block's all-gather overlaps the steps "
                         "that fill the next one on RCCL's stream)")
    ap.add_argument("--graph-steps", type=int, default=0, metavar="G",
                    help="single GPU: capture G (even) consecutive steps into ONE HIP graph "
                         "(BatchedTradingEnv.capture_steps) and time replays of it — one host call "
                         "per G steps — instead of G launches; for launch-bound batches (c2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-python-loop", type=float, default=0.0, metavar="SECONDS",
                    help="cpu_baseline calibration only, no GPU: time the interpreter-bound per-env loop "
                         "(oracle/py_loop.py) on one core for SECONDS and print env-steps/s "
                         "(tools/time_reference.py records it next to the reference's own timing)")
    args = ap.parse_args()
    if args.cpu_python_loop > 0:  # part of the cpu_baseline leg: the only place that touches oracle/
        from oracle.py_loop import time_loop
        print(time_loop(seconds=args.cpu_python_loop)[0])
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.pmc_child:
        self_launch(sys.argv[1:], args.gpus)  # children first: this process never touches the GPU
    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a
    # version banner at communicator creation), so everything but that line goes to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    wl = WORKLOADS[args.workload]
    N = args.envs or wl["envs"]
    n_dyn = 2
    W = wl["windows"] or 1
    F_obs = wl["n_static"] + n_dyn
    shape = (W, F_obs, wl["n_static"], n_dyn)
    cycle = wl["max_episode_duration"]
    hbm_leg = (world == 1 and not args.pmc_child and not args.no_hbm_regime
               and args.workload == "c3" and N == wl["envs"])

    # Counter passes first, while this process has not touched the GPU: each is a child
    # `rocprofv3 --pmc X -- python3 bench.py --pmc-child ...` (the program itself after `--`).
    live = {}
    profiled = ("rocprof" in os.environ.get("LD_PRELOAD", "").lower()
                or any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ))
    if profiled:  # this process is itself being profiled: no nested profiler runs
        print("[bench] running under a profiler: traffic comes from profiles/hbm_traffic.json", file=sys.stderr)
    if world == 1 and not args.pmc_child and not args.no_pmc and not profiled:
        for n in [N] + ([HBM_REGIME_ENVS] if hbm_leg else []):
            t = pmc_traffic(args, n)
            if not t:
                break  # rocprofv3 missing or failing: do not spend the time again
            live[n] = t

    import torch
    import torch.distributed as dist

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
    use_dist = (world > 1 or force_dist) and not args.pmc_child
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)  # RCCL over xGMI
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start bench.py bare (it launches "
                         f"its own ranks) or under torchrun with --nproc-per-node {args.gpus}")
    # the observation all-gather: asked for, or part of the workload (c4)
    want_obs = use_dist and (args.gather_obs or (bool(wl.get("gather_obs")) and not args.no_gather_obs))
    both_obs = use_dist and bool(wl.get("gather_obs")) and not args.gather_obs  # c4 reports both

    modes = []
    if use_dist:
        first = "step" if args.gather_mode == "auto" else args.gather_mode
        modes = [first] + ([] if args.one_gather_mode and args.gather_mode != "auto" else
                           [m for m in ("step", "step-overlap", "block") if m != first])
        if backend != "nccl":  # the library's communicator is RCCL: gloo rehearsals use torch's
            modes = [m for m in modes if m in ("block", "torch-step")] or ["block"]
    depth = max(2, args.gather_depth)
    block = max(1, args.gather_every)

    env = make_env(args, wl, N, rank, local_rank, return_slots=depth * block if use_dist else 1)
    gen = torch.Generator(device=dev)
    gen.manual_seed(99 + rank)
    n_rows = 64
    actions = torch.randint(0, 3, (n_rows, N), dtype=torch.int32, device=dev, generator=gen)
    env.reset()
    if not args.sync_episodes:
        desynchronise(env, actions, cycle)

    if args.pmc_child:  # counted by rocprofv3 from outside: just run the launches
        run_steps(env, actions, 0, args.warmup + args.steps)
        env.synchronize()
        env.close()
        return

    # the return of a sharded run: RCCL all-gather of the packed (reward f32 | terminated u8 |
    # truncated u8) records, 6 bytes per env and step, which the kernel writes directly in
    # that layout
    comm = returns = pipe = None
    if use_dist:
        from gym_trading_env_amd.distributed import ReturnGather, ReturnPipeline
        if any(m in ("step", "step-overlap") for m in modes):
            from gym_trading_env_amd.distributed import NativeReturnGather
            try:
                comm = NativeReturnGather(env, with_obs=want_obs or both_obs, mode=1)  # libgte's own communicator
                ok = torch.ones(1, device=dev)
            except Exception as e:  # noqa: BLE001 - RCCL not loadable / communicator refused
                print(f"[bench] rank {rank}: libgte's RCCL communicator unavailable ({e!r}); "
                      "gathering through torch.distributed instead", file=sys.stderr)
                comm, ok = None, torch.zeros(1, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # every rank takes the same path
            if float(ok) == 0.0:
                if comm is not None:
                    comm.close()
                    comm = None
                modes = [("torch-step" if m == "step" else m) for m in modes if m != "step-overlap"]
        returns = ReturnGather(N, dev, obs_shape=env.obs_shape if (want_obs or both_obs) else None,
                               depth=depth, block=block)
        returns1 = ReturnGather(N, dev)  # torch-step: plain synchronous gather
        pipe = ReturnPipeline(env, returns, block, depth)

    def timed(mode, first_step, with_obs=None):
        """W warm-up + K timed steps with the returns gathered the `mode` way -> (wall s, event ms)."""
        with_obs = want_obs if with_obs is None else with_obs
        def one_step(i):
            if mode == "block":
                pipe.before_step()  # rows about to be rewritten must have been gathered
            elif mode == "step-overlap":
                comm.wait(1)        # the gather of step t-2 has released what step t rewrites
            obs = env.step(actions[i % n_rows])[0]
            if mode == "block":
                pipe.after_step()   # starts the block's all-gather on block boundaries
            elif mode == "step":
                comm.mode = 0       # ncclAllGather on the env's stream, right behind the step kernel
                comm.gather()
            elif mode == "step-overlap":
                comm.mode = 1       # on the communication stream, beside the next step
                comm.gather()
            elif mode == "torch-step":
                returns1.gather(env.packed_returns)
            if mode is not None and with_obs:
                (comm.gather_obs() if mode in ("step", "step-overlap") else returns.gather_obs(obs))

        def drain():
            if mode == "block":
                pipe.flush()  # incl. a block the step count left unfinished
            elif mode == "step-overlap":
                comm.wait(0)

        for i in range(args.warmup):
            one_step(first_step + i)
        drain()
        torch.cuda.synchronize(dev)
        graph, G = None, args.graph_steps
        if G and mode is None:  # recorded here, right before the timed region (nothing executes)
            base = first_step + args.warmup
            graph = env.capture_steps(lambda i: env.step(actions[(base + i) % n_rows]), G)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        env.timer_start()  # HIP events on the stream the kernel is launched on
        done = 0
        if graph is not None:
            for _ in range(args.steps // G):
                graph.replay()
            done = args.steps // G * G
        for i in range(done, args.steps):
            one_step(first_step + args.warmup + i)
        drain()  # every all-gather belongs to the timed region
        env.timer_mark()  # end event recorded behind the last launch; read after the wall clock
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        ev_ms = env.timer_stop()
        if world > 1:
            t = torch.tensor([el, ev_ms], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el, ev_ms = float(t[0]), float(t[1])
        return el, ev_ms

    if args.gather_mode == "auto" and "step" in modes and "step-overlap" in modes:
        # which per-step form is faster here?  (untimed probe: a few steps of each)
        keep = (args.warmup, args.steps)
        args.warmup, args.steps = 3, 12
        probe = {m: timed(m, 0)[0] for m in ("step", "step-overlap")}
        args.warmup, args.steps = keep
        best = min(probe, key=probe.get)
        print(f"[bench] per-step gather probe: {probe} -> {best}", file=sys.stderr)
        modes = [best] + [m for m in modes if m != best]
        if args.one_gather_mode:
            modes = modes[:1]
    episodes0 = int(env.state("episode").sum())
    el, ev_ms = timed(modes[0] if modes else None, 0)
    episodes = int(env.state("episode").sum()) - episodes0
    mode = modes[0] if modes else None
    other_modes = {}
    for k, m in enumerate(modes[1:]):
        e2, _ = timed(m, (k + 1) * (args.warmup + args.steps))
        other_modes[m] = {"ms_per_step": e2 * 1e3 / args.steps, "value": world * N * args.steps / e2}
    other_obs = None
    if both_obs and mode is not None:  # c4: the same run with the observation gather switched
        e3, _ = timed(mode, (len(modes) + 1) * (args.warmup + args.steps), with_obs=not want_obs)
        other_obs = {"ms_per_step": e3 * 1e3 / args.steps, "value": world * N * args.steps / e3}

    # (episodes really end inside the timed region: auto-reset is part of the step)
    info = env.launch_info()
    kernel_us = ev_ms * 1e3 / args.steps
    if comm is not None:
        comm.close()
    env.close()

    hbm = None
    if hbm_leg and rank == 0:
        # the same kernel where the observations cannot stay on chip: 262 144 envs
        big = make_env(args, wl, HBM_REGIME_ENVS, 0, local_rank)
        acts = torch.randint(0, 3, (8, HBM_REGIME_ENVS), dtype=torch.int32, device=dev, generator=gen)
        big.reset()
        if not args.sync_episodes:
            desynchronise(big, acts, cycle)
        steps_b = max(20, min(args.steps, 200))
        run_steps(big, acts, 0, 10)
        big.timer_start()
        run_steps(big, acts, 10, steps_b)
        us_b = big.timer_stop() * 1e3 / steps_b
        hbm = roofline_block(HBM_REGIME_ENVS, us_b, shape,
                             live.get(HBM_REGIME_ENVS) or recorded_traffic(args.workload, HBM_REGIME_ENVS))
        hbm.update(envs=HBM_REGIME_ENVS, steps=steps_b, env_steps_per_s=HBM_REGIME_ENVS / (us_b * 1e-6),
                   launch=big.launch_info())
        big.close()

    if rank == 0:
        par = f"env-shard x{world}"
        how = {"block": f" in {block}-step blocks overlapping the following steps (torch.distributed): "
                        f"global returns arrive up to {block} steps late",
               "step": ", synchronous after every step on the env's stream, libgte's own RCCL "
                       "communicator (C ABI gte_allgather_returns): per-step global returns",
               "step-overlap": ", after every step on libgte's communication stream, overlapping the "
                               "next step (gte_allgather_returns mode 1): global returns one step late",
               "torch-step": ", synchronous per step (torch.distributed)"}
        if use_dist:
            par += " + RCCL all-gather(reward,flags" + (",obs)" if want_obs else ")") + how[mode]
        D = wl["n_datasets"]
        what = f"{args.workload}: {N} envs/GPU x obs ({W},{F_obs}) f32, "
        if args.workload == "c4":
            what = (f"c4: {N} envs/GPU (262 144 envs sharded over 8 GPUs; {world * N} global here) x obs "
                    f"({W},{F_obs}) f32, returns " + ("AND observations all-gathered over xGMI"
                                                     if want_obs else "all-gathered, observations kept local")
                    + " after every step, ")
        out = {
            "metric": "env-steps/sec", "value": world * N * args.steps / el, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": what +
                                   f"{wl['n_datasets']} dataset(s)/GPU of T={wl['T']}, positions [-1,0,1], "
                                   f"fees 1e-4, borrow 3e-6, max_episode_duration {cycle}, next-step autoreset",
                       "envs_per_gpu": N, "global_envs": world * N, "parallelism": par,
                       "episode_phase": ("synchronised (all envs reset together every %d steps)" % cycle
                                         if args.sync_episodes else
                                         "staggered by an untimed prologue: ~N/%d episode ends and "
                                         "auto-resets in every step" % cycle),
                       "launch": info, "episodes_finished": episodes,
                       "submission": (f"HIP graph: {args.graph_steps} consecutive steps captured once "
                                      f"(hipStreamBeginCapture through torch.cuda.graph), one replay per "
                                      f"{args.graph_steps} steps" if args.graph_steps and mode is None
                                      else "one gte_step call (one launch) per step"),
                       "datasets": ("one dataset, replicated on every rank" if D == 1 else
                                    f"{world * D} symbols partitioned by rank: rank r keeps symbols "
                                    f"{D}r .. {D}r+{D - 1} resident and its {N} envs only ever visit those "
                                    f"(per-env dataset_index, switch at every episode); no remote reads")},
            "roofline": roofline_block(N, kernel_us, shape,
                                       live.get(N) or recorded_traffic(args.workload, N)),
        }
        if other_modes:  # the same run with the returns gathered the other ways
            out["config"]["other_gather_modes"] = {
                m: dict(v, parallelism=f"env-shard x{world} + RCCL all-gather(reward,flags"
                        + (",obs)" if want_obs else ")") + how[m]) for m, v in other_modes.items()}
        if other_obs:
            out["config"]["without_obs_gather" if want_obs else "with_obs_gather"] = dict(
                other_obs, parallelism=f"env-shard x{world} + RCCL all-gather(reward,flags"
                + (")" if want_obs else ",obs)") + how[mode])
        if hbm:
            out["roofline"]["hbm_regime"] = hbm
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
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
