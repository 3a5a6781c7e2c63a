"""Stable-Baselines3 `VecEnv` face of the batched env (host arrays).

SB3's algorithms drive a `VecEnv`: `reset() -> obs`, `step_async(actions)` /
`step_wait() -> (obs, rewards, dones, infos)` with envs resetting themselves when an episode
ends, the last observation of the finished episode under `infos[i]["terminal_observation"]`
and `infos[i]["TimeLimit.truncated"]` telling truncation from termination.  That is
`BatchedTradingEnv(autoreset="same_step", final_obs=True, output="numpy")`; this class adds the
method names (the fork's training scripts, luckymodel/scripts, drive the reference through
`DummyVecEnv([make_env] * n)` — this replaces that list of Python envs by one launch per step).

Duck-typed: it does not import stable_baselines3 (not installed in the build image); when SB3 is
importable the class also derives from its `VecEnv`, so `isinstance` checks pass.
"""
from __future__ import annotations

import numpy as np

from . import spaces
from .batched import BatchedTradingEnv

try:  # pragma: no cover - exercised only where SB3 is installed
    from stable_baselines3.common.vec_env import VecEnv as _Base
except Exception:  # noqa: BLE001
    _Base = object


class SB3TradingVecEnv(_Base):
    """`SB3TradingVecEnv(df, num_envs, **TradingEnv kwargs)`.

    infos: one dict per env and step, as SB3 expects; besides the two SB3 keys each carries the
    scalars of the reference's History row that policies and callbacks usually read
    (`portfolio_valuation`, `position`, `idx`); `info_keys=` selects others (LazyInfo names)."""

    def __init__(self, df, num_envs, info_keys=("idx", "position", "portfolio_valuation"), **kw):
        for k in ("autoreset", "final_obs", "output"):
            if k in kw:
                raise TypeError(f"{k} is fixed by the VecEnv contract")
        self.env = BatchedTradingEnv(df, num_envs=num_envs, autoreset="same_step", final_obs=True,
                                     output="numpy", **kw)
        self.num_envs = int(num_envs)
        self.observation_space = spaces.Box(-np.inf, np.inf, shape=self.env.obs_shape)
        self.action_space = spaces.Discrete(len(self.env.positions))
        self.info_keys = tuple(info_keys)
        self.render_mode = None
        self._actions = None
        if _Base is not object:  # pragma: no cover
            _Base.__init__(self, self.num_envs, self.observation_space, self.action_space)

    # -- VecEnv API --------------------------------------------------------------------------
    def reset(self):
        obs, _ = self.env.reset()
        return obs

    def step_async(self, actions):
        self._actions = np.asarray(actions, dtype=np.int32).reshape(self.num_envs)

    def step_wait(self):
        obs, reward, terminated, truncated, info = self.env.step(self._actions)
        dones = terminated | truncated
        cols = {k: info[k] for k in self.info_keys}
        infos = [{k: cols[k][e] for k in self.info_keys} for e in range(self.num_envs)]
        for e in range(self.num_envs):
            infos[e]["TimeLimit.truncated"] = bool(truncated[e] and not terminated[e])
        if dones.any():
            ids, last = self.env.final_observations()
            for e, o in zip(ids, last):
                infos[int(e)]["terminal_observation"] = o
        return obs, reward.astype(np.float32), dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.env.close()

    def seed(self, seed=None):
        """Episode draws come from the device Philox stream keyed by the constructor's `seed`
        (the reference draws from NumPy's global RNG and ignores `seed=` too)."""
        return [None] * self.num_envs

    def get_attr(self, attr_name, indices=None):
        return [getattr(self.env, attr_name)] * len(self._indices(indices))

    def set_attr(self, attr_name, value, indices=None):
        setattr(self.env, attr_name, value)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        return [getattr(self.env, method_name)(*args, **kwargs)] * len(self._indices(indices))

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * len(self._indices(indices))

    def get_images(self):
        return [None] * self.num_envs

    def render(self, mode=None):
        return None

    def _indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        return [indices] if isinstance(indices, int) else list(indices)
