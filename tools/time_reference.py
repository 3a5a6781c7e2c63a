#!/usr/bin/env python3
"""Time the REFERENCE's own Python step() in the build container (it cannot travel to the
GPU box): configs C1/C2/C3 shapes, one process and one process per core.  Writes
profiles/reference_cpu_timing.json, which bench.py quotes beside its on-box baseline.
Uses the same gymnasium stand-in as tests/golden/make_golden.py (no arithmetic in it)."""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def _env(shape):
    import make_golden as mg  # installs the stand-in and imports the reference
    if shape == "c1":
        df, _, _ = mg.btc_dataset(33092)
        return mg.TradingEnv(df=df, positions=[0, 1], verbose=0)
    n_static, windows = (14, None) if shape == "c2" else (30, 20)
    feat, close = mg.random_walk(1234, 100_000, n_static, sigma=1e-3)
    df = mg.make_df(feat, close)
    return mg.TradingEnv(df=df, positions=[-1, 0, 1], windows=windows, trading_fees=1e-4,
                         borrow_interest_rate=3e-6, max_episode_duration=500, verbose=0)


def _run(args):
    shape, seconds = args
    env = _env(shape)
    P = len(env.positions)
    rng = np.random.default_rng(os.getpid())
    np.random.seed(os.getpid() % 2**31)
    env.reset()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(200):
            _, _, done, trunc, _ = env.step(int(rng.integers(0, P)))
            n += 1
            if done or trunc:
                env.reset()
    return n / (time.perf_counter() - t0)


def main():
    cores = len(os.sched_getaffinity(0))
    out = {"where": "build container (no GPU)", "cores": cores,
           "cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"),
           "python": sys.version.split()[0], "numpy": np.__version__,
           "note": "reference imported unchanged from /root/reference/src; uniform random actions; "
                   "episodes reset inside the timed loop", "shapes": {}}
    for shape in ("c1", "c2", "c3"):
        one = _run((shape, 4.0))
        with mp.Pool(cores) as pool:
            many = pool.map(_run, [(shape, 4.0)] * cores)
        out["shapes"][shape] = {"steps_per_s_1_process": one, f"steps_per_s_{cores}_processes": sum(many)}
        print(shape, out["shapes"][shape], flush=True)
    # the build's own interpreter-bound loop (oracle/py_loop.py), timed HERE right after the
    # reference: bench.py times the same loop on the GPU box and uses the ratio to rescale
    # the reference's numbers to that box's cores
    # (through bench.py's cpu_baseline leg: nothing outside tests/, smoke() and that leg touches oracle/)
    import subprocess
    rates = [float(subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"),
                                            "--cpu-python-loop", "3.0"]).decode().split()[-1])
             for _ in range(3)]
    out["python_loop_c3_steps_per_s_1_process"] = float(np.median(rates))
    print("py_loop c3", rates, flush=True)
    with open(os.path.join(ROOT, "profiles", "reference_cpu_timing.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
