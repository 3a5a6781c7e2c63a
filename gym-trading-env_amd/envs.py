"""Drop-in `TradingEnv` / `MultiDatasetTradingEnv`: the reference's single-environment
API (src/gym_trading_env/environments.py:26-400) on top of the HIP hot path.

One environment is a batch of one: every step is one libgte launch (the arithmetic
never runs on the host) and the results are copied back to build the same return
values as the reference: `(obs ndarray, reward, done, truncated, info dict)`, a
`History` log, episode metrics, `save_for_render`.  Episode draws use NumPy's GLOBAL
legacy RNG with the reference's calls in the reference's order
(`np.random.choice(positions)` :167, `np.random.randint(low, high)` :174,
`np.random.randint(n)` :385) and are injected into the kernel, so under
`np.random.seed(k)` the drop-in replays the reference's episodes exactly.

This N=1 form is interpreter- and PCIe-bound like the reference; throughput comes from
`BatchedTradingEnv`.  Differences, all documented in DESIGN.md: dynamic features and
device rewards are limited to the built-ins; any other `reward_function` /
`dynamic_feature_functions` callable — whatever its name — runs on the host over the History.
"""
from __future__ import annotations

import datetime
import glob
import os
from pathlib import Path

import numpy as np

from . import _abi, spaces, staging
from .batched import BatchedTradingEnv
from .config import HOST_CALLABLE, resolve_dynamic_features, resolve_reward
from .defaults import (basic_reward_function, dynamic_feature_last_position_taken,  # noqa: F401
                       dynamic_feature_real_position)
from .history import History

try:  # soft dependency
    import gymnasium as _gym
    _EnvBase = _gym.Env
except Exception:  # gymnasium absent: same reset(seed=) contract, nothing else needed
    class _EnvBase:
        metadata = {}

        def reset(self, seed=None, options=None):
            if seed is not None:
                self.np_random = np.random.default_rng(seed)


class TradingEnv(_EnvBase):
    """Single-asset trading environment; same constructor as the reference (:79-93)."""

    metadata = {"render_modes": ["logs"]}

    def __init__(self, df, positions=[0, 1],
                 dynamic_feature_functions=[dynamic_feature_last_position_taken,
                                            dynamic_feature_real_position],
                 reward_function=basic_reward_function, windows=None, trading_fees=0,
                 borrow_interest_rate=0, portfolio_initial_value=1000,
                 initial_position="random", max_episode_duration="max", verbose=1,
                 name="Stock", render_mode="logs", device=0):
        self.max_episode_duration = max_episode_duration
        self.name = name
        self.verbose = verbose
        self.positions = positions
        self.dynamic_feature_functions = dynamic_feature_functions
        self.reward_function = reward_function
        self.windows = windows
        self.trading_fees = trading_fees
        self.borrow_interest_rate = borrow_interest_rate
        self.portfolio_initial_value = float(portfolio_initial_value)
        self.initial_position = initial_position
        assert self.initial_position in self.positions or self.initial_position == "random", (
            "The 'initial_position' parameter must be 'random' or a position mentionned in the "
            "'position' (default is [0, 1]) parameter.")
        assert render_mode is None or render_mode in self.metadata["render_modes"]
        self.render_mode = render_mode
        self._device = device
        # The device computes this package's own default objects (recognised by IDENTITY, never
        # by name) and the string / tuple specs.  Any other callable is the user's code: one such
        # dynamic feature makes ALL of them host-side columns, evaluated over the History like the
        # reference does (:153-154) ...
        kinds = resolve_dynamic_features(dynamic_feature_functions)
        self._host_dyn = HOST_CALLABLE in kinds
        if self._host_dyn and not all(callable(f) for f in dynamic_feature_functions):
            raise TypeError("with a custom dynamic feature every entry of dynamic_feature_functions "
                            "must be a callable")
        # ... and a custom reward callable is evaluated on the host over the History (:265-267)
        self._host_reward = resolve_reward(reward_function)[0] == HOST_CALLABLE
        self._batch = None
        self._set_df(df)
        self.action_space = spaces.Discrete(len(positions))
        shape = [self._nb_features] if windows is None else [windows, self._nb_features]
        self.observation_space = spaces.Box(-np.inf, np.inf, shape=shape)
        self.log_metrics = []

    # -- data staging (:128-143) ---------------------------------------------------------
    def _set_df(self, df):
        self.df = df.copy()
        n_dyn = len(self.dynamic_feature_functions)
        # host-side dynamic features: the device holds the static columns only, the dynamic
        # ones live in `_dyn_array` (the last columns of the reference's `_obs_array`)
        self._staged = staging.stage_dataframe(self.df, n_dyn=0 if self._host_dyn else n_dyn,
                                               name=self.name)
        self._features_columns = self._staged.feature_columns
        self._info_columns = self._staged.info_columns
        self._nb_static_features = self._staged.n_static
        self._nb_features = self._staged.n_static + n_dyn
        self._info_array = self._staged.info_array
        self._price_array = self._staged.close
        if self._host_dyn:
            if self._nb_static_features == 0:
                raise NotImplementedError("custom dynamic features need at least one static "
                                          "'feature' column in the DataFrame")
            self._dyn_array = np.zeros((len(self.df), n_dyn), np.float32)
        if self._batch is not None:
            self._batch.close()
        # a fresh batch == a fresh `_obs_array` (dynamic columns zero); dyn_persist keeps
        # this env's in-place dynamic-feature writes across episodes like :153-154
        self._batch = BatchedTradingEnv(
            self._staged, num_envs=1, positions=self.positions,
            dynamic_feature_functions=[] if self._host_dyn else self.dynamic_feature_functions,
            reward_function="basic_reward_function" if self._host_reward else self.reward_function,
            windows=self.windows, trading_fees=self.trading_fees,
            borrow_interest_rate=self.borrow_interest_rate,
            portfolio_initial_value=self.portfolio_initial_value,
            initial_position=self.initial_position,
            max_episode_duration=self.max_episode_duration, verbose=0, name=self.name,
            autoreset=None, dyn_persist=True, device=self._device, output="numpy")

    # -- helpers -------------------------------------------------------------------------
    def _sync_state(self):
        """One transfer per step/reset: state, returns and observation of the env."""
        snap, self._last_obs = self._batch.read_env(0)
        self._idx, self._step = snap.idx, snap.step
        self._position = self.positions[snap.position_index]
        self._portfolio_state = {"asset": snap.asset, "fiat": snap.fiat,
                                 "interest_asset": snap.interest_asset,
                                 "interest_fiat": snap.interest_fiat}
        self._portfolio_value = snap.portfolio_valuation
        self._real_position = snap.real_position
        self._snap = snap

    def _distribution(self):  # Portfolio.get_portfolio_distribution, portfolio.py:49-57
        s = self._portfolio_state
        return {"asset": max(0, s["asset"]), "fiat": max(0, s["fiat"]),
                "borrowed_asset": max(0, -s["asset"]), "borrowed_fiat": max(0, -s["fiat"]),
                "interest_asset": s["interest_asset"], "interest_fiat": s["interest_fiat"]}

    def _row(self, position_index, real_position, valuation):
        """One History row; the column set and order the reference logs (:186-197, :253-264)."""
        t = self._idx
        return {"idx": t, "step": self._step, "date": self.df.index.values[t],
                "position_index": position_index, "position": self._position,
                "real_position": real_position,
                "data": dict(zip(self._info_columns, self._info_array[t])),
                "portfolio_valuation": valuation,
                "portfolio_distribution": self._distribution(), "reward": 0}

    def _get_price(self, delta=0):
        return self._price_array[self._idx + delta]

    def _get_ticker(self, delta=0):
        return self.df.iloc[self._idx + delta]

    def _obs(self):
        """`_get_obs` (:152-160).  Device-side dynamic features arrive inside the snapshot's
        observation; host-side ones are evaluated over the History now, written in place at
        row `_idx` (f32, they persist like the reference's) and appended to the static part."""
        if not self._host_dyn:
            return self._last_obs
        t = self._idx
        for i, fn in enumerate(self.dynamic_feature_functions):
            self._dyn_array[t, i] = fn(self.historical_info)
        if self.windows is None:
            return np.concatenate([self._last_obs, self._dyn_array[t]])
        return np.concatenate([self._last_obs, self._dyn_array[t + 1 - self.windows:t + 1]], axis=1)

    # -- reset (:163-199) ----------------------------------------------------------------
    def reset(self, seed=None, options=None, **kwargs):
        super().reset(seed=seed, options=options, **kwargs)
        position = (np.random.choice(self.positions) if self.initial_position == "random"
                    else self.initial_position)
        idx = 0 if self.windows is None else self.windows - 1
        if self.max_episode_duration != "max":
            idx = np.random.randint(low=idx, high=len(self.df) - self.max_episode_duration - idx)
        self._limit_orders = {}
        self._batch.reset(inject_idx=[int(idx)],
                          inject_position_index=[self.positions.index(position)])
        self._sync_state()
        self.historical_info = History(max_size=len(self.df))
        self.historical_info.set(**self._row(self.positions.index(self._position),
                                             real_position=self._position,
                                             valuation=self.portfolio_initial_value))
        return self._obs(), self.historical_info[0]

    def render(self):
        pass

    def add_limit_order(self, position, limit, persistent=False):
        """Pending order (environments.py:227-231): when `low <= limit <= high` at a new
        row and `position` differs from the current one, trade to it at `limit`.  The
        order table lives on the device; `self._limit_orders` mirrors what was asked.
        NB a filled non-persistent order is removed and the step goes on — the
        reference deletes it while iterating its dict and raises RuntimeError (:223)."""
        self._limit_orders[position] = {"limit": limit, "persistent": persistent}
        self._batch.add_limit_order([self.positions.index(position)], [float(limit)],
                                    bool(persistent))

    # -- step (:233-272) -----------------------------------------------------------------
    def step(self, position_index=None):
        if position_index is not None:
            self.positions[position_index]  # IndexError / TypeError like :234
        if self._idx + 1 >= len(self._price_array):
            raise IndexError(f"index {self._idx + 1} is out of bounds for axis 0 with size "
                             f"{len(self._price_array)}")  # :239 past the last row
        self._batch._launch_step([-1 if position_index is None else int(position_index)])
        self._sync_state()
        done, truncated = bool(self._snap.terminated), bool(self._snap.truncated)
        self.historical_info.add(**self._row(position_index, real_position=self._real_position,
                                             valuation=self._portfolio_value))
        if not done:
            reward = (self.reward_function(self.historical_info) if self._host_reward
                      else np.float64(self._snap.reward))
            self.historical_info["reward", -1] = reward
        if done or truncated:
            self.calculate_metrics()
            self.log()
        return (self._obs(), self.historical_info["reward", -1], done, truncated,
                self.historical_info[-1])

    # -- metrics / logs (:274-294) ---------------------------------------------------------
    def add_metric(self, name, function):
        self.log_metrics.append({"name": name, "function": function})

    def calculate_metrics(self):
        h = self.historical_info

        def pct(column):  # first-to-last change of a History column, the reference's format
            return f"{100 * (h[column, -1] / h[column, 0] - 1):5.2f}%"
        results = {"Market Return": pct("data_close"),
                   "Portfolio Return": pct("portfolio_valuation")}
        results.update((m["name"], m["function"](h)) for m in self.log_metrics)
        self.results_metrics = results

    def get_metrics(self):
        return self.results_metrics

    def log(self):
        if self.verbose <= 0:
            return
        print("".join(f"{name} : {value}   |   " for name, value in self.results_metrics.items()))

    def save_for_render(self, dir="render_logs"):
        """Pickle `df` joined with the episode History (:296-307), for the renderer."""
        import pandas as pd
        missing = [c for c in ("open", "high", "low", "close") if c not in self.df]
        assert not missing, (
            "Your DataFrame needs to contain columns : open, high, low, close to render !")
        h = self.historical_info
        logged = pd.DataFrame({c: h[c] for c in h.columns}).set_index("date").sort_index()
        os.makedirs(dir, exist_ok=True)
        stamp = datetime.datetime.now().strftime("%Y-%m-%d_%H-%M-%S")
        self.df.join(logged, how="inner").to_pickle(os.path.join(dir, f"{self.name}_{stamp}.pkl"))

    def close(self):
        if self._batch is not None:
            self._batch.close()
            self._batch = None


class MultiDatasetTradingEnv(TradingEnv):
    """A TradingEnv that moves to another dataset every
    `episodes_between_dataset_switch` episodes (:365-400): uniform choice among the
    least-used datasets, `preprocess` applied to each loaded frame."""

    def __init__(self, dataset_dir, *args, preprocess=lambda df: df,
                 episodes_between_dataset_switch=1, **kwargs):
        self.dataset_dir = dataset_dir
        self.preprocess = preprocess
        self.episodes_between_dataset_switch = episodes_between_dataset_switch
        self.dataset_pathes = glob.glob(self.dataset_dir)
        if len(self.dataset_pathes) == 0:
            raise FileNotFoundError(f"No dataset found with the path : {self.dataset_dir}")
        self.dataset_nb_uses = np.zeros(shape=(len(self.dataset_pathes),))
        super().__init__(self.next_dataset(), *args, **kwargs)

    def next_dataset(self):
        """Uniform draw among the least-used dataset files (:380-391); same single
        `np.random.randint(n)` call as the reference, so the global RNG stays in step."""
        import pandas as pd
        self._episodes_on_this_dataset = 0
        uses = self.dataset_nb_uses
        least_used = np.flatnonzero(uses == uses.min())
        chosen = int(least_used[np.random.randint(least_used.size)])
        uses[chosen] += 1
        self.name = Path(self.dataset_pathes[chosen]).name
        frame = pd.read_pickle(self.dataset_pathes[chosen])  # the user's own dataset files
        return self.preprocess(frame)

    def reset(self, seed=None, options=None, **kwargs):
        self._episodes_on_this_dataset += 1
        if self._episodes_on_this_dataset % self.episodes_between_dataset_switch == 0:
            self._set_df(self.next_dataset())
        if self.verbose > 1:
            print(f"Selected dataset {self.name} ...")
        return super().reset(seed=seed, options=options, **kwargs)
