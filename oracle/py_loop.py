"""TEST INFRASTRUCTURE / CPU BASELINE ONLY — never imported by the product.

A second, independent restatement of the reference hot path: ONE environment per Python
object, Python floats, a dict appended to a log every step — i.e. interpreter-bound in the
same way the reference is.  Two uses:

* tests/test_oracle_golden.py replays the golden traces through it (pure-Python loops,
  small cases), so the C oracle and this file pin each other against the reference's vectors;
* bench.py times it for ~2 s on one core of the GPU box.  The same loop was timed in the
  build container next to the reference itself (profiles/reference_cpu_timing.json); the
  ratio box/container rescales the reference's own timing, whose files cannot travel.

Scope: single dataset, the two default dynamic features (any subset via `dyn`), the three
built-in rewards, windows or none, margin path, persistent limit orders are NOT restated here
(the C oracle covers them).  Reference lines cited per method.
"""
from __future__ import annotations

import math

import numpy as np


class Book:
    """Portfolio / TargetPortfolio (utils/portfolio.py:1-66) on Python floats."""
    __slots__ = ("asset", "fiat", "ia", "ifi")

    def __init__(self, position, value, price):  # TargetPortfolio.__init__ :59-66
        self.asset = position * value / price
        self.fiat = (1 - position) * value
        self.ia = 0
        self.ifi = 0

    def worth(self, price):  # valorisation :7-13 — sum() of a 4-list, left to right from 0
        return sum([self.asset * price, self.fiat, -self.ia * price, -self.ifi])

    def exposure(self, price):  # position :16-17
        return self.asset * price / self.worth(price)

    def net_exposure(self, price):  # real_position :14-15
        return (self.asset - self.ia) * price / self.worth(price)

    def retarget(self, position, price, fees):  # trade_to_position :18-43
        now = self.exposure(price)
        keep = 1
        if position <= 0 and now < 0:
            keep = min(1, position / now)
        elif position >= 1 and now > 1:
            keep = min(1, (position - 1) / (now - 1))
        if keep != 1:  # repay part of the accrued interest (:26-30)
            self.asset = self.asset - (1 - keep) * self.ia
            self.fiat = self.fiat - (1 - keep) * self.ifi
            self.ia = keep * self.ia
            self.ifi = keep * self.ifi
        qty = position * self.worth(price) / price - self.asset  # :33
        if qty > 0:  # buy: fees taken in asset (:34-38)
            qty = qty / (1 - fees + fees * position)
            cash = -qty * price
            self.asset = self.asset + qty * (1 - fees)
            self.fiat = self.fiat + cash
        else:  # sell: fees taken in fiat (:39-43)
            qty = qty / (1 - fees * position)
            cash = -qty * price
            self.asset = self.asset + qty
            self.fiat = self.fiat + cash * (1 - fees)

    def accrue(self, rate):  # update_interest :44-46 — assignment, not accumulation
        self.ia = max(0, -self.asset) * rate
        self.ifi = max(0, -self.fiat) * rate


class PyEnv:
    """One TradingEnv (environments.py:79-272), restated.  `table` is f32 [T, F_s + n_dyn]
    with the dynamic columns last; it is written in place like the reference's `_obs_array`."""

    def __init__(self, table, close, positions, windows=None, trading_fees=0.0,
                 borrow_interest_rate=0.0, portfolio_initial_value=1000.0,
                 initial_position="random", max_episode_duration="max", dyn=("position", "real"),
                 reward=("log",), persist=True):
        self.table, self.close = table, close
        self.T, self.n_dyn = len(close), len(dyn)
        self.fs = table.shape[1] - self.n_dyn
        self.positions, self.W = list(positions), windows
        self.fees, self.rate, self.v0 = trading_fees, borrow_interest_rate, portfolio_initial_value
        self.initial_position, self.max_dur = initial_position, max_episode_duration
        self.dyn, self.reward_spec, self.persist = tuple(dyn), tuple(reward), persist
        self.ended = False

    # -- reset :163-199 ------------------------------------------------------------------
    def reset(self, idx=None, pos_index=None):
        """`idx` / `pos_index` None: drawn from NumPy's global RNG in the reference's order
        (:167 position first, :173-177 start row second)."""
        if not self.persist:
            self.table[:, self.fs:] = 0  # a fresh env object: no values from older episodes
        self.step_no = 0
        if pos_index is None:
            position = (np.random.choice(self.positions) if self.initial_position == "random"
                        else self.initial_position)
        else:
            position = self.positions[pos_index]
        self.position = position
        first = 0 if self.W is None else self.W - 1
        if idx is None:
            idx = first
            if self.max_dur != "max":
                idx = np.random.randint(low=first, high=self.T - self.max_dur - first)
        self.idx = int(idx)
        price = float(self.close[self.idx])
        self.book = Book(position, self.v0, price)
        self.log = [dict(idx=self.idx, step=0, position=position, real_position=position,
                         portfolio_valuation=self.v0, reward=0, data_close=price)]
        self.ended = self.done = self.truncated = False
        self.reward = 0
        return self._observe()

    # -- step :233-272 -------------------------------------------------------------------
    def step(self, position_index=None):
        if position_index is not None and position_index >= 0:  # _take_action :213-215
            target = self.positions[position_index]
            if target != self.position:  # value compare; _trade :204-211 at close[idx]
                self.book.retarget(target, float(self.close[self.idx]), self.fees)
                self.position = target
        self.idx += 1
        self.step_no += 1
        price = float(self.close[self.idx])  # IndexError past the last row, like the reference
        self.book.accrue(self.rate)
        value = self.book.worth(price)
        done = truncated = False
        if value / self.v0 <= 0.7:  # :246-247 (historical 30 % drawdown rule)
            done = True
        if self.idx >= self.T - 1:  # :249
            truncated = True
        if isinstance(self.max_dur, int) and self.step_no >= self.max_dur - 1:  # :250-251
            truncated = True
        entry = dict(idx=self.idx, step=self.step_no, position=self.position,
                     real_position=self.book.net_exposure(price), portfolio_valuation=value,
                     reward=0, data_close=price)
        self.log.append(entry)
        if not done:  # :265-267
            entry["reward"] = self._reward(value, self.log[-2]["portfolio_valuation"])
        self.reward, self.done, self.truncated = entry["reward"], done, truncated
        self.ended = done or truncated
        return self._observe(), entry["reward"], done, truncated, entry

    def _reward(self, value, before):
        lr = float(np.log(value / before))  # basic_reward_function :17-18
        kind = self.reward_spec[0]
        if kind == "scaled":
            return self.reward_spec[1] * lr
        if kind == "clipped":
            return float(np.clip(self.reward_spec[1] * lr, self.reward_spec[2], self.reward_spec[3]))
        return lr

    def _observe(self):  # _get_obs :152-160
        last = self.log[-1]
        for i, name in enumerate(self.dyn):
            self.table[self.idx, self.fs + i] = last["position" if name == "position"
                                                     else "real_position"]
        if self.W is None:
            return self.table[self.idx]
        return self.table[self.idx + 1 - self.W:self.idx + 1]


def time_loop(n_static=30, windows=20, T=100_000, max_dur=500, seconds=2.0, seed=3):
    """env-steps/s of PyEnv on ONE core for a config-3-shaped env with uniform random
    actions and a reset whenever an episode ends (what a user loop over the reference does)."""
    import time
    rng = np.random.default_rng(seed)
    close = 100.0 * np.exp(np.cumsum(rng.normal(0, 1e-3, T)))
    table = np.zeros((T, n_static + 2), np.float32)
    table[:, :n_static] = rng.normal(0, 1, (T, n_static)).astype(np.float32)
    env = PyEnv(table, close, [-1, 0, 1], windows=windows, trading_fees=1e-4,
                borrow_interest_rate=3e-6, max_episode_duration=max_dur)
    np.random.seed(seed)
    env.reset()
    acts = rng.integers(0, 3, 4096).tolist()
    n, t0 = 0, time.perf_counter()
    while True:
        for a in acts:
            env.step(a)
            if env.ended:
                env.reset()
        n += len(acts)
        el = time.perf_counter() - t0
        if el >= seconds:
            return n / el, n, el
