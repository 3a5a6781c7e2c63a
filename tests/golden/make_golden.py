#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself.

Runs only in the build container, where /root/reference is mounted; the GPU
box and the test-suite never execute this file, they read the .npz it wrote.
The reference (ten2net/Gym-Trading-Env, pure Python) is imported UNCHANGED
from /root/reference/src.  Its only missing third-party import is
`gymnasium`, which contributes no arithmetic to step()/reset(): the reference
uses it for the `gym.Env` base class, two space descriptors and the registry
(environments.py:1-2,26,112-123; __init__.py:1-14).  A ~25-line in-memory
module supplies exactly those names.

What is committed is DATA: the inputs (feature table, close prices, actions,
the reference's own random draws at each reset) and the outputs the reference
produced for them, per call.  No reference source is copied.

Trace format (one .npz per scenario, E reference env objects x K calls):
  cfg_json                  kwargs for gym_trading_env_amd.config.make_config
  feat_<d> f32 [T, F_s], close_<d> f64 [T]          the datasets
  op      u8  [K, E]   0 = env.reset() was called, 1 = env.step(action)
  action  i32 [K, E]   position index, -1 = None (hold); ignored when op == 0
  idx, step, pos_index, dataset i32 [K, E]           state after the call
  position, real_position, asset, fiat, interest_asset, interest_fiat,
  portfolio_valuation, reward f64 [K, E]
  done, truncated u8 [K, E]
  obs     f32 [K, E, (W,) F_obs]                     observation returned
  seed_base, fresh_env_each_episode: np.random.seed(seed_base + 7919*e + episode) was
  called before every reset of env e; whether a new env object was built per episode
Call k of env e follows Gymnasium's NEXT_STEP convention: the call after a
terminal step is the reset (reward 0, flags false), so a batched env with
next-step auto-reset replays the whole trace with one reset() + K-1 step()s.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
import types

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/src"
REF_CSV = "/root/reference/examples/data/BTC_USD-Hourly.csv"


def install_gymnasium_stand_in():
    """The four names the reference takes from gymnasium, and nothing else."""
    if "gymnasium" in sys.modules:
        return
    gym = types.ModuleType("gymnasium")

    class Env:
        metadata = {}

        def reset(self, seed=None, options=None):
            if seed is not None:
                self.np_random = np.random.default_rng(seed)

    class Discrete:
        def __init__(self, n):
            self.n = n

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    spaces = types.ModuleType("gymnasium.spaces")
    spaces.Discrete, spaces.Box = Discrete, Box
    envs = types.ModuleType("gymnasium.envs")
    registration = types.ModuleType("gymnasium.envs.registration")
    registration.register = lambda **kw: None
    envs.registration = registration
    gym.Env, gym.spaces, gym.envs = Env, spaces, envs
    sys.modules.update({"gymnasium": gym, "gymnasium.spaces": spaces,
                        "gymnasium.envs": envs,
                        "gymnasium.envs.registration": registration})


install_gymnasium_stand_in()
sys.path.insert(0, REF_SRC)
from gym_trading_env.environments import MultiDatasetTradingEnv, TradingEnv  # noqa: E402

FIELDS_F64 = ("position", "real_position", "asset", "fiat", "interest_asset",
              "interest_fiat", "portfolio_valuation", "reward")
FIELDS_I32 = ("idx", "step", "pos_index", "dataset")


def make_df(feat, close, high=None, low=None):
    T = len(close)
    df = pd.DataFrame({"open": close, "high": close * 1.001 if high is None else high,
                       "low": close * 0.999 if low is None else low,
                       "close": close, "volume": np.ones(T)},
                      index=pd.date_range("2020-01-01", periods=T, freq="h"))
    for j in range(feat.shape[1]):
        df[f"feature_{j}"] = feat[:, j]
    return df


def random_walk(seed, T, n_static, sigma=1e-3, drift=0.0):
    """SURVEY §8d synthetic shape: close = 100*exp(cumsum(N(drift, sigma)))."""
    rng = np.random.default_rng(seed)
    close = 100.0 * np.exp(np.cumsum(rng.normal(drift, sigma, T)))
    feat = rng.normal(0.0, 1.0, (T, n_static)).astype(np.float32)
    return feat, close


def snapshot(env, rec, k, e, positions, ds_names):
    pf = env._portfolio
    price = env._get_price()
    rec["idx"][k, e] = env._idx
    rec["step"][k, e] = env._step
    rec["pos_index"][k, e] = positions.index(env._position)
    # env.name stays "Stock" until the first switch (environments.py:390 then :95),
    # so the dataset is identified by its length (the fixture's datasets differ in T)
    rec["dataset"][k, e] = ds_names.index(len(env.df)) if ds_names else 0
    rec["position"][k, e] = env._position
    rec["real_position"][k, e] = env.historical_info["real_position", -1]
    rec["asset"][k, e] = pf.asset
    rec["fiat"][k, e] = pf.fiat
    rec["interest_asset"][k, e] = pf.interest_asset
    rec["interest_fiat"][k, e] = pf.interest_fiat
    rec["portfolio_valuation"][k, e] = env.historical_info["portfolio_valuation", -1]
    assert price == env.historical_info["data_close", -1]


def run_trace(make_env, positions, n_envs, n_calls, action_rng, p_none=0.1,
              fresh_env_each_episode=False, autoreset=True, ds_names=None,
              seed_base=1000, actions=None, p_order=0.0):
    """Drive n_envs reference env objects for n_calls calls each."""
    rec = {f: np.zeros((n_calls, n_envs), np.float64) for f in FIELDS_F64}
    rec.update({f: np.zeros((n_calls, n_envs), np.int32) for f in FIELDS_I32})
    rec["op"] = np.ones((n_calls, n_envs), np.uint8)
    rec["action"] = np.zeros((n_calls, n_envs), np.int32)
    rec["done"] = np.zeros((n_calls, n_envs), np.uint8)
    rec["truncated"] = np.zeros((n_calls, n_envs), np.uint8)
    if p_order > 0:  # env.add_limit_order(position, limit, persistent=True) BEFORE call k
        rec["lo_pos"] = np.full((n_calls, n_envs), -1, np.int32)
        rec["lo_limit"] = np.zeros((n_calls, n_envs), np.float64)
    obs_rec = None
    P = len(positions)
    for e in range(n_envs):
        env = make_env(e)
        ended = True
        episode = 0
        for k in range(n_calls):
            if k == 0 or (ended and autoreset):
                if fresh_env_each_episode and k > 0:
                    env = make_env(e)
                # the reference draws from the GLOBAL legacy NumPy RNG
                # (environments.py:167,174,385); `seed=` is inert for it
                np.random.seed(seed_base + 7919 * e + episode)
                obs, info = env.reset()
                episode += 1
                rec["op"][k, e] = 0
                rec["action"][k, e] = -1
                reward, done, trunc = 0.0, False, False
                assert info["idx"] == env._idx
            else:
                if actions is not None:
                    a = int(actions[k, e])
                else:
                    a = -1 if action_rng.random() < p_none else int(action_rng.integers(0, P))
                rec["action"][k, e] = a
                if p_order > 0 and action_rng.random() < p_order:
                    pi = int(action_rng.integers(0, P))
                    limit = float(env._get_price() * (1 + action_rng.normal(0, 0.01)))
                    # non-persistent orders crash the reference when they fill
                    # (RuntimeError: dictionary changed size during iteration, :223)
                    env.add_limit_order(positions[pi], limit, persistent=True)
                    rec["lo_pos"][k, e] = pi
                    rec["lo_limit"][k, e] = limit
                try:
                    obs, reward, done, trunc, info = env.step(None if a < 0 else a)
                except IndexError:
                    raise SystemExit(f"scenario steps env {e} past its data at call {k}")
            ended = bool(done or trunc)
            snapshot(env, rec, k, e, positions, ds_names)
            rec["reward"][k, e] = float(reward)
            rec["done"][k, e] = done
            rec["truncated"][k, e] = trunc
            obs = np.array(obs, dtype=np.float32)
            if obs_rec is None:
                obs_rec = np.zeros((n_calls, n_envs) + obs.shape, np.float32)
            obs_rec[k, e] = obs
    rec["obs"] = obs_rec
    # what a replay through the single-env API needs to reproduce the global-RNG draws
    rec["seed_base"] = np.array(seed_base)
    rec["fresh_env_each_episode"] = np.array(int(fresh_env_each_episode))
    return rec


def save(name, cfg, datasets, rec, note):
    out = {"cfg_json": np.array(json.dumps(cfg)), "note": np.array(note)}
    for d, ds in enumerate(datasets):
        out[f"feat_{d}"] = np.asarray(ds[0], np.float32)
        out[f"close_{d}"] = np.asarray(ds[1], np.float64)
        if len(ds) == 4:
            out[f"high_{d}"] = np.asarray(ds[2], np.float64)
            out[f"low_{d}"] = np.asarray(ds[3], np.float64)
    out.update(rec)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    if "op" not in rec:  # not a trace (e.g. the metrics fixture)
        print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")
        return
    ends = int((rec["done"] | rec["truncated"]).sum())
    assert rec["op"].ndim == 2
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, calls={rec['op'].shape}, "
          f"resets={int((rec['op'] == 0).sum())}, ends={ends}, done={int(rec['done'].sum())}")


def ref_kwargs(cfg):
    kw = {k: cfg[k] for k in ("positions", "windows", "trading_fees", "borrow_interest_rate",
                              "portfolio_initial_value", "initial_position",
                              "max_episode_duration")}
    kw["verbose"] = 0
    return kw


def base_cfg(**over):
    cfg = dict(positions=[0, 1], windows=None, trading_fees=0, borrow_interest_rate=0,
               portfolio_initial_value=1000, initial_position="random",
               max_episode_duration="max", reward_function="basic_reward_function")
    cfg.update(over)
    return cfg


def btc_dataset(n_rows):
    """examples/example_environnement.py:11-23 preprocessing of the shipped CSV."""
    df = pd.read_csv(REF_CSV, parse_dates=["date"], index_col="date")
    df.sort_index(inplace=True)
    df.dropna(inplace=True)
    df.drop_duplicates(inplace=True)
    df["feature_close"] = df["close"].pct_change()
    df["feature_open"] = df["open"] / df["close"]
    df["feature_high"] = df["high"] / df["close"]
    df["feature_low"] = df["low"] / df["close"]
    df["feature_volume"] = df["Volume USD"] / df["Volume USD"].rolling(7 * 24).max()
    df.dropna(inplace=True)
    df = df.iloc[:n_rows]
    cols = [c for c in df.columns if "feature" in c]
    return df, np.asarray(df[cols], np.float32), np.asarray(df["close"], np.float64)


def main():
    rng = np.random.default_rng(20240607)

    # -- C1: shipped BTC-USD hourly data, reference defaults (positions [0,1]) ------
    df, feat, close = btc_dataset(1500)
    cfg = base_cfg()
    rec = run_trace(lambda e: TradingEnv(df=df, **ref_kwargs(cfg)), cfg["positions"],
                    n_envs=3, n_calls=1500, action_rng=rng, autoreset=False)
    save("c1_btc_default", cfg, [(feat, close)], rec,
         "BASELINE config 1: examples/data/BTC_USD-Hourly.csv first 1500 processed rows, "
         "constructor defaults, one full episode to truncation at the last row")

    # -- C1b: the example script's own configuration --------------------------------
    cfg = base_cfg(positions=[-1, -0.5, 0, 0.5, 1, 1.5, 2], windows=5,
                   trading_fees=0.01 / 100, borrow_interest_rate=0.0003 / 100,
                   max_episode_duration=500)
    rec = run_trace(lambda e: TradingEnv(df=df, **ref_kwargs(cfg)), cfg["positions"],
                    n_envs=4, n_calls=700, action_rng=rng, fresh_env_each_episode=True)
    save("c1_btc_example", cfg, [(feat, close)], rec,
         "examples/example_environnement.py:30-43 configuration (7 positions incl. "
         "leverage, window 5, fees, borrow interest, random starts, duration 500), "
         "fresh reference env per episode")

    # -- C2 shape: no window, 14 static features, shorting, margin ------------------
    feat, close = random_walk(1234, 400, 14, sigma=5e-3)
    df2 = make_df(feat, close)
    cfg = base_cfg(positions=[-1, 0, 1], trading_fees=1e-4, borrow_interest_rate=3e-6)
    rec = run_trace(lambda e: TradingEnv(df=df2, **ref_kwargs(cfg)), cfg["positions"],
                    n_envs=8, n_calls=900, action_rng=rng)
    save("c2_nowindow", cfg, [(feat, close)], rec,
         "BASELINE config 2 shape (F_obs=16, positions [-1,0,1], fees 1e-4, borrow 3e-6), "
         "T=400, two full episodes per env on the same reference env object")

    # -- C3 shape: window 20, 30 static features, random starts, duration 100 -------
    feat, close = random_walk(1235, 360, 30, sigma=5e-3)
    df3 = make_df(feat, close)
    cfg = base_cfg(positions=[-1, 0, 1], windows=20, trading_fees=1e-4,
                   borrow_interest_rate=3e-6, max_episode_duration=100)
    rec = run_trace(lambda e: TradingEnv(df=df3, **ref_kwargs(cfg)), cfg["positions"],
                    n_envs=4, n_calls=230, action_rng=rng, fresh_env_each_episode=True)
    save("c3_window20", cfg, [(feat, close)], rec,
         "BASELINE config 3 shape (obs (20,32), margin path live), T=360, duration 100, "
         "fresh reference env per episode (zero dynamic columns before the start row)")

    # -- drawdown termination: leveraged positions on trending prices ---------------
    feat, close = random_walk(77, 300, 3, sigma=2e-2, drift=-4e-3)
    dfd = make_df(feat, close)
    cfg = base_cfg(positions=[-2, -1, 0, 1, 2, 3], windows=4, trading_fees=1e-3,
                   borrow_interest_rate=1e-3)
    rec = run_trace(lambda e: TradingEnv(df=dfd, **ref_kwargs(cfg)), cfg["positions"],
                    n_envs=8, n_calls=600, action_rng=rng, p_none=0.3)
    save("drawdown_done", cfg, [(feat, close)], rec,
         "falling, volatile prices with leverage: the pv/initial <= 0.7 `done` rule "
         "(environments.py:246) fires repeatedly; reward stays 0 on those steps")

    # -- no auto-reset: the reference keeps stepping after done / duration truncation
    cfg = base_cfg(positions=[-2, -1, 0, 1, 2, 3], trading_fees=1e-3,
                   borrow_interest_rate=1e-3, max_episode_duration=60, initial_position=2)
    rec = run_trace(lambda e: TradingEnv(df=dfd, **ref_kwargs(cfg)), cfg["positions"],
                    n_envs=6, n_calls=90, action_rng=rng, autoreset=False, seed_base=4242)
    save("no_autoreset", cfg, [(feat, close)], rec,
         "one reset then 89 steps: stepping continues after done and after the "
         "duration truncation exactly like the reference (flags stay raised)")

    # -- the in-place dynamic-feature write persists across episodes ----------------
    feat, close = random_walk(99, 120, 2, sigma=5e-3)
    dfp = make_df(feat, close)
    cfg = base_cfg(positions=[-1, 0, 1], windows=6, trading_fees=1e-4,
                   max_episode_duration=25, dyn_persist=True)
    rec = run_trace(lambda e: TradingEnv(df=dfp, **ref_kwargs(cfg)), cfg["positions"],
                    n_envs=4, n_calls=400, action_rng=rng)
    save("persist_dynamic", cfg, [(feat, close)], rec,
         "same reference env object over many random-start episodes: windows contain "
         "dynamic-feature values written by EARLIER episodes (environments.py:153-154)")

    # -- dynamic feature list variants + fork rewards --------------------------------
    def r_scaled(history):  # luckymodel/scripts/test_env.py:20-22
        return 100 * np.log(history["portfolio_valuation", -1] / history["portfolio_valuation", -2])

    def r_clipped(history):  # luckymodel/envs/env.py:16-18
        lr = np.log(history["portfolio_valuation", -1] / history["portfolio_valuation", -2])
        return np.clip(lr, -0.002, 0.005)

    from gym_trading_env.environments import dynamic_feature_real_position
    feat, close = random_walk(5, 200, 4, sigma=1e-2)
    dfr = make_df(feat, close)
    cfg = base_cfg(positions=[0, 0.5, 1], windows=3, trading_fees=0.01 / 100,
                   borrow_interest_rate=0.0003 / 100, portfolio_initial_value=1000000,
                   reward_function=["scaled_log_return", 100.0],
                   dynamic_feature_functions=["real_position"])
    rec = run_trace(lambda e: TradingEnv(df=dfr, reward_function=r_scaled,
                                         dynamic_feature_functions=[dynamic_feature_real_position],
                                         **ref_kwargs(cfg)),
                    cfg["positions"], n_envs=3, n_calls=420, action_rng=rng)
    save("reward_scaled_onedyn", cfg, [(feat, close)], rec,
         "fork usage (luckymodel/scripts/test_env.py): 100x log-return reward, positions "
         "[0,0.5,1], initial value 1e6, and a single dynamic feature (real_position)")
    cfg = base_cfg(positions=[0, 0.5, 1], trading_fees=0.01 / 100,
                   reward_function=["clipped_log_return", 1.0, -0.002, 0.005],
                   dynamic_feature_functions=[])
    rec = run_trace(lambda e: TradingEnv(df=dfr, reward_function=r_clipped,
                                         dynamic_feature_functions=[], **ref_kwargs(cfg)),
                    cfg["positions"], n_envs=3, n_calls=420, action_rng=rng)
    save("reward_clipped_nodyn", cfg, [(feat, close)], rec,
         "fork usage (luckymodel/envs/env.py:16-18): np.clip(log_return, -0.002, 0.005), "
         "no dynamic features, no window")

    # -- persistent limit orders (environments.py:217-231) ---------------------------
    feat, close = random_walk(41, 260, 3, sigma=1e-2)
    r2 = np.random.default_rng(42)
    high = close * (1 + np.abs(r2.normal(0, 8e-3, 260)))
    low = close * (1 - np.abs(r2.normal(0, 8e-3, 260)))
    dfl = make_df(feat, close, high, low)
    cfg = base_cfg(positions=[-1, 0, 1], windows=3, trading_fees=1e-3, borrow_interest_rate=1e-4)
    rec = run_trace(lambda e: TradingEnv(df=dfl, **ref_kwargs(cfg)), cfg["positions"],
                    n_envs=6, n_calls=600, action_rng=rng, p_none=0.5, p_order=0.15)
    save("limit_orders", cfg, [(feat, close, high, low)], rec,
         "persistent limit orders added between steps; they fill at the limit price when "
         "low <= limit <= high at the new row, several may fill in one step, reset clears them")

    # -- MultiDatasetTradingEnv over pickles written by THIS script ------------------
    with tempfile.TemporaryDirectory() as tmp:
        sets, names = [], []
        for d in range(5):
            f, c = random_walk(300 + d, 90 + 10 * d, 3, sigma=5e-3)
            sets.append((f, c))
            names.append(f"sym{d}.pkl")
            make_df(f, c).to_pickle(os.path.join(tmp, names[-1]))
        for switch, tag in ((1, "k1"), (3, "k3")):
            cfg = base_cfg(positions=[-1, 0, 1], windows=4, trading_fees=1e-4,
                           borrow_interest_rate=3e-6, max_episode_duration=30,
                           episodes_between_dataset_switch=switch,
                           dyn_persist=(switch > 1))
            # pick the constructor's dataset from a seeded global RNG too
            def mk(e, switch=switch, cfg=cfg):
                np.random.seed(555 + e)
                return MultiDatasetTradingEnv(os.path.join(tmp, "*.pkl"),
                                              episodes_between_dataset_switch=switch,
                                              **ref_kwargs(cfg))
            rec = run_trace(mk, cfg["positions"], n_envs=4, n_calls=330, action_rng=rng,
                            ds_names=[len(c) for _, c in sets])
            # glob order is filesystem order (environments.py:375) and decides which file
            # `np.random.randint(n)` (:385) selects: record it for single-env replays
            import glob as _glob
            rec["glob_order"] = np.array([names.index(os.path.basename(q))
                                          for q in _glob.glob(os.path.join(tmp, "*.pkl"))], np.int32)
            save(f"multidataset_{tag}", cfg, sets, rec,
                 f"MultiDatasetTradingEnv, 5 datasets of different lengths, switch every "
                 f"{switch} episode(s); with switch > 1 a dataset's _obs_array "
                 f"lives across episodes, so that trace needs dyn_persist")


def portfolio_vectors():
    """Random known answers straight from the reference's Portfolio class
    (utils/portfolio.py:1-66, which imports nothing): every branch of trade_to_position
    (interest repayment on both sides, buys, sells, leverage, shorts)."""
    from gym_trading_env.utils.portfolio import Portfolio, TargetPortfolio
    rng = np.random.default_rng(777)
    n = 3000
    inp = np.zeros((n, 9)); out = np.zeros((n, 6))
    for k in range(n):
        price = float(np.exp(rng.normal(4, 1)))
        start_pos = float(rng.choice([-2, -1, -0.5, 0, 0.3, 1, 1.5, 2, 3]))
        pf = TargetPortfolio(position=start_pos, value=float(rng.uniform(100, 5000)), price=price)
        rate = float(rng.choice([0, 1e-5, 1e-3]))
        pf.update_interest(rate)                      # some accrued interest to repay
        if rng.random() < 0.3:
            pf.interest_asset *= float(rng.uniform(1, 50)); pf.interest_fiat *= float(rng.uniform(1, 50))
        position = float(rng.choice([-2, -1, -0.5, 0, 0.5, 1, 1.5, 2, 3]))
        fees = float(rng.choice([0, 1e-4, 1e-3, 1e-2]))
        trade_px = price * float(np.exp(rng.normal(0, 0.05)))
        next_px = trade_px * float(np.exp(rng.normal(0, 0.05)))
        inp[k] = (pf.asset, pf.fiat, pf.interest_asset, pf.interest_fiat, position, trade_px, fees, rate, next_px)
        pf.trade_to_position(position, trade_px, fees)
        pf.update_interest(rate)
        out[k] = (pf.asset, pf.fiat, pf.interest_asset, pf.interest_fiat, pf.valorisation(next_px),
                  pf.real_position(next_px))
    np.savez_compressed(os.path.join(HERE, "portfolio_random.npz"), inputs=inp, outputs=out,
                        note=np.array("columns: asset, fiat, interest_asset, interest_fiat, position, "
                                      "trade_price, fees, rate, next_price -> asset, fiat, interest_asset, "
                                      "interest_fiat, valorisation(next), real_position(next)"))
    print("portfolio_random:", n, "vectors")


def custom_callables_trace():
    """hostcb_custom_callables.npz: custom Python `dynamic_feature_functions` and
    `reward_function` (docs/source/customization.rst) run INSIDE the reference.  The batch
    cannot run Python per env, so this fixture is replayed by the N=1 drop-in only
    (tests/test_gpu_dropin.py); the callables live in tests/custom_callables.py."""
    sys.path.insert(0, os.path.dirname(HERE))
    import custom_callables as cc
    from gym_trading_env.environments import dynamic_feature_last_position_taken
    rng = np.random.default_rng(424242)
    feat, close = random_walk(31, 260, 3, sigma=1.5e-2)
    df = make_df(feat, close)
    cfg = base_cfg(positions=[-1, 0, 0.5, 1, 2], windows=4, trading_fees=1e-3,
                   borrow_interest_rate=1e-4, max_episode_duration=40,
                   reward_function=["custom", "reward_simple_return_minus_turnover"],
                   dynamic_feature_functions=["last_position_taken", "custom:dyn_valuation_ratio",
                                              "custom:dyn_exposure_change"],
                   dyn_persist=True)
    rec = run_trace(lambda e: TradingEnv(
        df=df, reward_function=cc.reward_simple_return_minus_turnover,
        dynamic_feature_functions=[dynamic_feature_last_position_taken, cc.dyn_valuation_ratio,
                                   cc.dyn_exposure_change], **ref_kwargs(cfg)),
        cfg["positions"], n_envs=3, n_calls=300, action_rng=rng, seed_base=5150)
    save("hostcb_custom_callables", cfg, [(feat, close)], rec,
         "custom dynamic features and a custom reward evaluated by the reference over its "
         "History; same env object across episodes (in-place dynamic columns persist)")


def vector_example_fixture():
    """hostcb_vector_example.npz: the reference's VECTORISED example
    (examples/example_vectorized_environment.py:39-62: `gym.make_vec("TradingEnv", num_envs=3,
    windows=5, positions=[-1 .. 2], initial_position=0, fees, borrow rate, reward_function=...)`)
    as three reference env objects driven with the next-step convention, on a synthetic frame with
    the example's five feature recipes — plus the `info` dict (`history[-1]`, environments.py:272)
    of every call: its key list and every value (dates as int64 ns).  Episodes are bounded
    (max_episode_duration=40; the example runs 'max') so that several resets are covered."""
    sys.path.insert(0, os.path.dirname(HERE))
    import custom_callables as cc
    rng = np.random.default_rng(31337)
    T = 400
    close = 100.0 * np.exp(np.cumsum(rng.normal(0, 8e-3, T)))
    df = pd.DataFrame({"open": close * (1 + rng.normal(0, 2e-3, T)), "high": close * 1.006,
                       "low": close * 0.994, "close": close, "volume": rng.uniform(10, 500, T)},
                      index=pd.date_range("2019-01-01", periods=T, freq="30min"))
    df["feature_close"] = df["close"].pct_change()        # the example's recipes (:30-34)
    df["feature_open"] = df["open"] / df["close"]
    df["feature_high"] = df["high"] / df["close"]
    df["feature_low"] = df["low"] / df["close"]
    df["feature_volume"] = df["volume"] / df["volume"].rolling(48).max()
    df.dropna(inplace=True)
    feat = np.asarray(df[[c for c in df.columns if "feature" in c]], np.float32)
    cfg = base_cfg(positions=[-1, -0.5, 0, 0.5, 1, 1.5, 2], windows=5, initial_position=0,
                   trading_fees=0.01 / 100, borrow_interest_rate=0.0003 / 100,
                   portfolio_initial_value=1000, max_episode_duration=40,
                   reward_function=["custom", "reward_log_return_example"])
    infos = []

    def make(e):
        env = TradingEnv(df=df, name="BTCUSD", reward_function=cc.reward_log_return_example,
                         **ref_kwargs(cfg))
        real_step, real_reset = env.step, env.reset

        def step(a=None):
            out = real_step(a)
            infos.append(dict(out[4]))
            return out

        def reset(*a, **k):
            out = real_reset(*a, **k)
            infos.append(dict(out[1]))
            return out
        env.step, env.reset = step, reset
        return env
    n_envs, n_calls = 3, 150
    rec = run_trace(make, cfg["positions"], n_envs=n_envs, n_calls=n_calls, action_rng=rng, seed_base=2468)
    assert len(infos) == n_envs * n_calls
    keys = list(infos[0])
    assert all(list(i) == keys for i in infos)
    # the reference builds its info column list through a set() (environments.py:131): the ORDER
    # of the data_* keys changes with the interpreter's hash seed, so the fixture stores them sorted
    rec["info_keys"] = np.array(sorted(keys))
    for key in keys:  # run_trace goes env by env: infos[e * n_calls + k]
        col = [infos[e * n_calls + k][key] for k in range(n_calls) for e in range(n_envs)]
        if key == "date":
            arr = np.array(col, dtype="datetime64[ns]").astype(np.int64)
        elif key == "position_index":  # None for step(None) and for the reset row's int
            arr = np.array([-1 if v is None else int(v) for v in col], np.int64)
        else:
            arr = np.array(col, dtype=np.float64)
        rec[f"info_{key}"] = arr.reshape(n_calls, n_envs)
    for c in ("open", "high", "low", "volume"):
        rec[f"df_{c}"] = df[c].to_numpy(np.float64)
    rec["df_index_ns"] = df.index.values.astype("datetime64[ns]").astype(np.int64)
    save("hostcb_vector_example", cfg, [(feat, df["close"].to_numpy(np.float64))], rec,
         "the reference's vectorised example: its constructor arguments and Python reward function, "
         "three env objects, every call's info dict")


def metrics_fixture():
    """hostcb_metrics.npz: what `get_metrics()` returns at the end of each episode of one
    reference env (the two built-in percentage strings, environments.py:279-285, plus two
    user metrics added with add_metric), for a fixed seed and action sequence."""
    feat, close = random_walk(77, 400, 3, sigma=2e-2)
    df = make_df(feat, close)
    cfg = base_cfg(positions=[-1, 0, 1, 2], windows=3, trading_fees=1e-3,
                   borrow_interest_rate=1e-4, max_episode_duration=60)
    env = TradingEnv(df=df, **ref_kwargs(cfg))
    env.add_metric("Position Changes", lambda h: int(np.sum(np.diff(h["position"]) != 0)))
    env.add_metric("Episode Length", lambda h: len(h["position"]))
    rng = np.random.default_rng(99)
    actions = rng.integers(0, 4, (5, 80)).astype(np.int32)
    results = []
    for ep in range(5):
        np.random.seed(900 + ep)
        env.reset()
        done = trunc = False
        k = 0
        while not (done or trunc):
            _, _, done, trunc, _ = env.step(int(actions[ep, k]))
            k += 1
        results.append({str(a): (b if isinstance(b, str) else int(b))
                        for a, b in env.get_metrics().items()})
    save("hostcb_metrics", cfg, [(feat, close)],
         {"actions": actions, "metrics_json": np.array(json.dumps(results))},
         "get_metrics() of the reference after each of 5 episodes (np.random.seed(900 + ep) "
         "before each reset, actions[ep, k] at step k)")


HISTORY_ROWS = [
    dict(idx=0, step=0, position=1, data={"close": 10.0, "open": 9.0},
         portfolio_distribution={"asset": 1.0, "fiat": 0.0}, levels=[1, 2], reward=0),
    dict(idx=1, step=1, position=0, data={"close": 11.0, "open": 10.0},
         portfolio_distribution={"asset": 0.0, "fiat": 11.0}, levels=[3, 4], reward=0.5),
    dict(idx=2, step=2, position=-1, data={"close": 12.5, "open": 11.0},
         portfolio_distribution={"asset": -0.5, "fiat": 17.0}, levels=[5, 6], reward=-0.25),
]
HISTORY_EXPRESSIONS = [
    "h.columns", "len(h)", "h.size", "h['idx'].tolist()", "h['data_close'].tolist()", "h[1]", "h[-1]",
    "h['data_close', -1]", "h['position', 0]", "h['reward', 1]", "h['levels_1', -2]",
    "h[['position', 'levels_1']].tolist()", "h[['idx']].shape", "h['data_open', 0:2].tolist()",
    "h['step', -2:].tolist()", "str(h['idx'].dtype)", "h['nope']", "h['nope', 0]", "h[7]",
    "h['idx', 7]", "h.add(idx=3, step=3)", "h.add(**dict(ROWS[0], extra=1))",
]


def _jsonable(v):
    if isinstance(v, dict):
        return {str(k): _jsonable(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_jsonable(x) for x in v]
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (np.floating,)):
        return float(v)
    return v


def history_fixture():
    """history_ops.json: what the reference's History (utils/history.py) answers to a list of
    expressions (documented access patterns and a few misuse cases) after set() + two add()s,
    and after a write through history[col, t] = v.  Exceptions are recorded by type name."""
    from gym_trading_env.utils.history import History

    def build():
        h = History(max_size=10)
        h.set(**HISTORY_ROWS[0])
        for row in HISTORY_ROWS[1:]:
            h.add(**row)
        return h

    out = {}
    for expr in HISTORY_EXPRESSIONS:
        h = build()
        try:
            out[expr] = {"value": _jsonable(eval(expr, {"h": h, "ROWS": HISTORY_ROWS}))}
        except Exception as exc:  # noqa: BLE001
            out[expr] = {"raises": type(exc).__name__}
    h = build()
    h["reward", -1] = 0.75
    out["after h['reward', -1] = 0.75: h['reward'].tolist()"] = {"value": _jsonable(h["reward"].tolist())}
    full = History(max_size=2)
    full.set(a=1)
    full.add(a=2)
    try:
        full.add(a=3)
        out["add() beyond max_size"] = {"value": None}
    except Exception as exc:  # noqa: BLE001
        out["add() beyond max_size"] = {"raises": type(exc).__name__}
    with open(os.path.join(HERE, "history_ops.json"), "w") as f:
        json.dump({"rows": HISTORY_ROWS, "expressions": HISTORY_EXPRESSIONS, "answers": out}, f, indent=1)
    print("history_ops.json:", len(out), "answers")


def staging_fixture():
    """set_df.npz: what TradingEnv._set_df (environments.py:128-143) derives from a DataFrame
    with an awkward column set: which columns count as features (name CONTAINS 'feature'), which
    go to the info array, and the staged arrays themselves."""
    rng = np.random.default_rng(8)
    T = 40
    close = 100 * np.exp(np.cumsum(rng.normal(0, 1e-2, T)))
    cols = ["volume", "feature_b", "open", "my_feature_a", "close", "Feature_caps", "high",
            "refeatured", "low", "date_close"]
    df = pd.DataFrame({c: rng.normal(0, 1, T) for c in cols},
                      index=pd.date_range("2021-03-01", periods=T, freq="h"))
    df["close"] = close
    env = TradingEnv(df=df, positions=[0, 1], windows=None, verbose=0)
    np.savez_compressed(
        os.path.join(HERE, "set_df.npz"),
        columns=np.array(cols), values=df[cols].to_numpy(np.float64),
        features_columns=np.array(env._features_columns),
        # (the reference builds this list through a set(): its order follows the hash seed; the
        # fixture stores the columns sorted by name so that it regenerates bit-identically)
        info_columns=np.array(sorted(env._info_columns)),
        nb_features=np.array(env._nb_features), nb_static_features=np.array(env._nb_static_features),
        obs_array=np.asarray(env._obs_array, np.float32),
        price_array=np.asarray(env._price_array, np.float64),
        info_array=np.asarray(env._info_array, np.float64)[:, np.argsort(env._info_columns)])
    print("set_df.npz: features", env._features_columns, "info", env._info_columns)


if __name__ == "__main__":
    if "--only-vector" in sys.argv:
        vector_example_fixture()
        sys.exit(0)
    if "--only-custom" in sys.argv:
        custom_callables_trace()
        metrics_fixture()
        history_fixture()
        staging_fixture()
        sys.exit(0)
    main()
    portfolio_vectors()
    custom_callables_trace()
    vector_example_fixture()
    metrics_fixture()
    history_fixture()
    staging_fixture()
