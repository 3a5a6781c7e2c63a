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


class _Columns:
    """The per-step columns every env's info dict reads from (swapped once per step)."""
    __slots__ = ("cols",)

    def __init__(self):
        self.cols = {}


class EnvInfo(dict):
    """`infos[i]` of the VecEnv: a dict whose History scalars (`idx`, `position`,
    `portfolio_valuation`, ... — `info_keys`) are read from the step's shared column arrays on
    access instead of being copied into N dicts per step; the keys SB3 itself looks for
    (`terminal_observation`, `TimeLimit.truncated`, `episode` ...) are ordinary stored items, set
    only for the envs whose episode ended — for those the History scalars are stored items too,
    holding the TERMINAL step's values (what DummyVecEnv over the reference returns,
    environments.py:272), since the shared columns already describe the next episode.  A view of the CURRENT step: values change with the
    next step (use `dict(info)` / `info.copy()` to keep one)."""
    __slots__ = ("_c", "_i")

    def __init__(self, columns, i):
        super().__init__()
        self._c, self._i = columns, i

    def __missing__(self, k):
        return self._c.cols[k][self._i]  # KeyError for unknown keys, like a dict

    def get(self, k, default=None):
        if dict.__contains__(self, k):
            return dict.__getitem__(self, k)
        col = self._c.cols.get(k)
        return default if col is None else col[self._i]

    def __contains__(self, k):
        return dict.__contains__(self, k) or k in self._c.cols

    def keys(self):
        return list(self._c.cols) + [k for k in dict.keys(self) if k not in self._c.cols]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def copy(self):
        return dict(self.items())

    def __repr__(self):
        return repr(self.copy())


class SB3TradingVecEnv(_Base):
    """`SB3TradingVecEnv(df, num_envs, **TradingEnv kwargs)`.

    infos: one dict per env and step, as SB3 expects; besides the two SB3 keys (set for the envs
    whose episode ended) each exposes the scalars of the reference's History row that policies
    and callbacks usually read (`portfolio_valuation`, `position`, `idx`); `info_keys=` selects
    others (any `info` key of the batch: `date`, `data_close`, ...).  The N dicts are created
    ONCE: per step the host does O(envs that ended) work plus one array swap (`EnvInfo`), not
    O(N) dict building — 4 096 envs: `tools/sb3_host_rate.py`."""

    def __init__(self, df, num_envs, info_keys=("idx", "position", "portfolio_valuation"), **kw):
        for k in ("autoreset", "final_obs", "output"):
            if k in kw:
                raise TypeError(f"{k} is fixed by the VecEnv contract")
        from .config import HOST_CALLABLE, resolve_dynamic_features, resolve_reward
        custom = (resolve_reward(kw.get("reward_function", "basic_reward_function"))[0] == HOST_CALLABLE
                  or HOST_CALLABLE in resolve_dynamic_features(
                      kw.get("dynamic_feature_functions", ("last_position_taken", "real_position"))))
        # a user's Python reward / dynamic-feature callable (the fork's training scripts pass one,
        # luckymodel/envs/env.py:16-18) is evaluated on the device over a BatchedHistory: the inner
        # env then keeps torch outputs and this adapter copies them to the host arrays SB3 wants
        self._torch_inner = bool(custom)
        self.env = BatchedTradingEnv(df, num_envs=num_envs, autoreset="same_step", final_obs=True,
                                     output="torch" if custom else "numpy", **kw)
        self.num_envs = int(num_envs)
        self.observation_space = spaces.Box(-np.inf, np.inf, shape=self.env.obs_shape)
        self.action_space = spaces.Discrete(len(self.env.positions))
        self.info_keys = tuple(info_keys)
        self.render_mode = None
        self._actions = None
        self._columns = _Columns()
        self._infos = [EnvInfo(self._columns, e) for e in range(self.num_envs)]
        self._dirty = []  # envs whose dict holds stored items from the previous step
        if _Base is not object:  # pragma: no cover
            _Base.__init__(self, self.num_envs, self.observation_space, self.action_space)

    # -- VecEnv API --------------------------------------------------------------------------
    def _host(self, x):
        return x.cpu().numpy() if self._torch_inner else x

    def reset(self):
        obs, _ = self.env.reset()
        return self._host(obs)

    def step_async(self, actions):
        self._actions = np.asarray(actions, dtype=np.int32).reshape(self.num_envs)

    def step_wait(self):
        obs, reward, terminated, truncated, info = self.env.step(self._actions)
        obs, reward, terminated, truncated = (self._host(x) for x in (obs, reward, terminated, truncated))
        dones = terminated | truncated
        self._columns.cols = {k: info[k] for k in self.info_keys}
        infos = self._infos
        for e in self._dirty:
            dict.clear(infos[e])
        self._dirty = []
        if dones.any():
            env = self.env
            ids, last = env.final_observations()
            # DummyVecEnv over the reference returns the TERMINAL step's info for an env that ended
            # (environments.py:272), with terminal_observation added; the shared columns already
            # describe the episode the in-launch reset started, so the terminal values (from the
            # terminal records, gte_get_final_state) are stored in the dict, shadowing the columns
            r64 = env.read_output("reward64") if "reward" in self.info_keys else None
            final = {k: np.asarray(env._info_value(k, env.final_state, r64))[ids] for k in self.info_keys}
            ids, last = ids.tolist(), self._host(last)
            for j, (e, o) in enumerate(zip(ids, last)):
                d = infos[e]
                for k, col in final.items():
                    dict.__setitem__(d, k, col[j])
                dict.__setitem__(d, "terminal_observation", o)
                dict.__setitem__(d, "TimeLimit.truncated", bool(truncated[e] and not terminated[e]))
            self._dirty = ids
        return obs, reward.astype(np.float32, copy=False), dones, infos

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
