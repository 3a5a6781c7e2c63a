"""BatchedHistory — what a batch hands to `reward_function(history)`,
`dynamic_feature_functions` and metrics instead of the reference's per-env `History`
(src/gym_trading_env/utils/history.py:3-76, docs/source/history.rst:18-46).

Same access patterns, one value PER ENV:

    h["portfolio_valuation", -1]   [N]     the newest row of every env     (history.py:55-59)
    h["portfolio_valuation", -2]   [N]     the row before
    h["position", 0]               [N]     the first row of every env's CURRENT episode
    h["position"]                  [R, N]  the last R = len(h) rows of every env, oldest first
    h[["idx", "reward"]]           [R, N, 2]
    h[-1]                          dict of [N] arrays (the `info` of the step)
    h["reward", -1] = x            overwrite the newest reward (environments.py:267)

Backed by the device trajectory log (`log_steps` rows per env, gte_config.log_steps, what
`History.add` records each step, environments.py:253-264).  With `output="torch"` values
are `DeviceArray`s — torch tensors in HBM that NumPy formulas such as
`np.log(h[..., -1] / h[..., -2])` compute with unchanged (device_array.py); with
`output="numpy"` they are ndarrays copied from the device on access.

Differences from one env's History, inherent to a batch: `h[col]` is a sliding window of the
last `log_steps` steps of every env (episodes of different envs start at different rows;
`h.episode_mask()` marks the rows of each env's current episode), and non-numeric columns
(`date`, object-valued `data_*`) are host arrays.
"""
from __future__ import annotations

import numpy as np

from . import _abi

_DIST = ("asset", "fiat", "borrowed_asset", "borrowed_fiat", "interest_asset", "interest_fiat")
_LOG_COLUMNS = {"idx": "idx", "step": "step", "position_index": "position_index",
                "dataset_index": "dataset_index", "portfolio_valuation": "portfolio_valuation",
                "real_position": "real_position", "reward": "reward"}


def history_columns(env) -> list:
    """Flattened column names in the reference's order (environments.py:186-197, 253-264;
    history.py:20-33 flattening of the `data` and `portfolio_distribution` dicts)."""
    info = env.datasets[0].info_columns or ["close"]
    return (["idx", "step", "date", "position_index", "position", "real_position"]
            + [f"data_{c}" for c in info] + ["portfolio_valuation"]
            + [f"portfolio_distribution_{k}" for k in _DIST] + ["reward"])


class BatchedHistory:
    def __init__(self, env, terminal: bool = False):
        """terminal=True (same-step auto-reset): for the envs whose episode ended in the last
        step the NEWEST row is the terminal row `TradingEnv.step` logged (environments.py:253-264)
        — taken from the terminal records, because the log row of that step already holds the
        state after the in-launch reset; every other row and env reads the log as usual."""
        if not env.cfg.log_steps:
            raise ValueError("BatchedHistory needs log_steps > 0 (it reads the device trajectory log)")
        self._env = env
        self._terminal = bool(terminal)
        self.columns = history_columns(env) + ["dataset_index"]
        view = env._log_view()
        self._rows, self._L = int(view.rows), int(view.L)
        self._have = min(self._rows, self._L)

    def __len__(self):
        """Rows available per env: min(steps logged so far, log_steps)."""
        return self._have

    # -- row selection -------------------------------------------------------------------------
    def _phys(self, t):
        """Physical log row (scalar) of relative row t < 0, or per-env rows [N] of episode row
        t >= 0 (row t of every env's current episode)."""
        if t < 0:
            if -t > self._have:
                raise IndexError(f"index {t} is out of bounds: {self._have} rows are logged")
            return (self._rows + t) % self._L
        step = self._env._log_row("step", (self._rows - 1) % self._L, raw=True)
        if self._terminal:
            step = self._env._overlay(step, "step")
        back = step - t  # rows between the wanted row and the newest one
        if bool((back < 0).any()) or bool((back >= self._have).any()):
            raise IndexError(f"index {t} is outside the current episode / the {self._L} logged rows "
                             "of some env")
        return (self._rows - 1 - back) % self._L

    def _column(self, name, phys):
        """Values of column `name` at physical row(s) `phys`: a scalar row, a per-env row vector
        [N] or None for every logged row, oldest first ([R, N])."""
        v = self._log_column(name, phys)
        if not self._terminal:
            return v
        e, newest = self._env, (self._rows - 1) % self._L
        if phys is None:      # the last row of the window is the newest one
            v = v.clone() if hasattr(v, "clone") else v.copy()
            v[-1] = e._overlay(v[-1], name)
            return v
        if np.ndim(phys) == 0:
            return e._overlay(v, name) if int(phys) == newest else v
        return e._overlay(v, name, only=(phys == newest))

    def _log_column(self, name, phys):
        e = self._env
        if name in _LOG_COLUMNS:
            return e._log_rows(name, phys, self._order())
        if name == "position":
            return e._take(e._positions_table(), e._log_rows("position_index", phys, self._order()))
        if name.startswith("portfolio_distribution_"):
            k = name[len("portfolio_distribution_"):]
            if k in ("interest_asset", "interest_fiat"):
                return e._log_rows(k, phys, self._order())
            if k in _DIST:  # Portfolio.get_portfolio_distribution, portfolio.py:49-57
                src = e._log_rows("asset" if k.endswith("asset") else "fiat", phys, self._order())
                return e._relu(-src if k.startswith("borrowed") else src)
        if name == "date" or name.startswith("data_"):
            return e._dataset_column(name, e._log_rows("dataset_index", phys, self._order()),
                                     e._log_rows("idx", phys, self._order()))
        raise ValueError(f"Feature {name} does not exist ... Check the available features : {self.columns}")

    def _order(self):
        """Physical rows of the logged window, oldest first."""
        return (np.arange(self._have) + self._rows - self._have) % self._L

    # -- the History protocol --------------------------------------------------------------------
    def __getitem__(self, arg):
        e = self._env
        if isinstance(arg, tuple):
            column, t = arg
            if isinstance(t, slice):
                return self[column][t]
            return e._wrap(self._column(column, self._phys(int(t))))
        if isinstance(arg, (int, np.integer)):
            phys = self._phys(int(arg))
            return {c: e._wrap(self._column(c, phys)) for c in self.columns}
        if isinstance(arg, str):
            return e._wrap(self._column(arg, None))
        if isinstance(arg, list):
            return e._wrap(e._stack([self._column(c, None) for c in arg]))
        raise TypeError(f"unsupported History index {arg!r}")

    def __setitem__(self, arg, value):
        column, t = arg
        if column != "reward" or int(t) != -1:
            raise ValueError("only h['reward', -1] can be assigned (environments.py:267)")
        self._env._set_log_reward(value)

    def episode_mask(self):
        """bool [R, N]: True where the logged row belongs to the env's CURRENT episode."""
        e = self._env
        step_now = e._log_rows("step", (self._rows - 1) % self._L, self._order())
        age = e._arange_rows(self._have)  # 0 = oldest logged row
        return e._wrap((self._have - 1 - age)[:, None] <= step_now[None, :])
