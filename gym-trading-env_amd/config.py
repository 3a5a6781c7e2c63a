"""TradingEnv constructor arguments -> struct gte_config.

Mirrors the argument handling of the reference constructor
(src/gym_trading_env/environments.py:79-109): same names, defaults, coercions
and assertion messages, plus the batch-only knobs.
"""
from __future__ import annotations

import ctypes as C

from . import _abi

from . import defaults

#: rewards the device computes, by EXPLICIT spec: a string, or a tuple carrying parameters
#: ("scaled_log_return", k) / ("clipped_log_return", k, lo, hi) — the fork's forms
#: (luckymodel/envs/env.py:16-18).  Callables are never matched by name: only this package's
#: own default object (identity) maps to the device; see resolve_reward.
_REWARD_BY_NAME = {
    "basic_reward_function": _abi.REWARD_LOG_RETURN,
    "log_return": _abi.REWARD_LOG_RETURN,
    "scaled_log_return": _abi.REWARD_SCALED_LOG_RETURN,
    "clipped_log_return": _abi.REWARD_CLIPPED_LOG_RETURN,
}
_DYN_BY_NAME = {
    "dynamic_feature_last_position_taken": _abi.DYN_LAST_POSITION,
    "last_position_taken": _abi.DYN_LAST_POSITION,
    "dynamic_feature_real_position": _abi.DYN_REAL_POSITION,
    "real_position": _abi.DYN_REAL_POSITION,
}
_DYN_BY_OBJECT = {
    defaults.dynamic_feature_last_position_taken: _abi.DYN_LAST_POSITION,
    defaults.dynamic_feature_real_position: _abi.DYN_REAL_POSITION,
}
_AUTORESET_BY_NAME = {
    None: _abi.AUTORESET_DISABLED, "disabled": _abi.AUTORESET_DISABLED,
    "next_step": _abi.AUTORESET_NEXT_STEP, "same_step": _abi.AUTORESET_SAME_STEP,
}

#: what resolve_* answer for a user's own Python callable (evaluated outside the kernel)
HOST_CALLABLE = -1


def device_dynamic_kind(f):
    """Device enum of one `dynamic_feature_functions` entry, or HOST_CALLABLE.

    On device: this package's two default objects (by identity), the enum ints, and the
    string specs.  ANY other callable is the user's code and is evaluated as such."""
    if isinstance(f, bool):
        raise TypeError(f"dynamic feature {f!r} is neither a callable nor a spec")
    if isinstance(f, int):
        if f not in (_abi.DYN_LAST_POSITION, _abi.DYN_REAL_POSITION):
            raise ValueError(f"unknown dynamic feature enum {f}")
        return int(f)
    if isinstance(f, str):
        if f not in _DYN_BY_NAME:
            raise ValueError(f"unknown dynamic feature {f!r}; known: {sorted(_DYN_BY_NAME)}")
        return _DYN_BY_NAME[f]
    if callable(f):
        try:
            return _DYN_BY_OBJECT.get(f, HOST_CALLABLE)
        except TypeError:  # unhashable callable object
            return HOST_CALLABLE
    raise TypeError(f"dynamic feature {f!r} is neither a callable nor a spec")


def resolve_dynamic_features(dynamic_feature_functions) -> list[int]:
    """Map the reference's `dynamic_feature_functions` list onto device enums; HOST_CALLABLE
    marks the entries that are the user's own callables."""
    kinds = [device_dynamic_kind(f) for f in dynamic_feature_functions]
    if len(kinds) > _abi.GTE_MAX_DYN:
        raise ValueError(f"at most {_abi.GTE_MAX_DYN} dynamic features")
    return kinds


def resolve_reward(reward_function) -> tuple[int, float, float, float]:
    """Map `reward_function` onto (kind, param0, param1, param2); kind == HOST_CALLABLE for a
    user's own callable.

    On device: this package's `basic_reward_function` object (identity), a string spec, or a
    tuple ("scaled_log_return", k) / ("clipped_log_return", k, lo, hi), i.e.
    np.clip(k * log_return, lo, hi) (luckymodel/envs/env.py:16-18)."""
    if isinstance(reward_function, tuple):
        name, *params = reward_function
        if name not in _REWARD_BY_NAME:
            raise ValueError(f"unknown reward spec {name!r}; known: {sorted(_REWARD_BY_NAME)}")
        kind = _REWARD_BY_NAME[name]
        p = [float(x) for x in params] + [1.0, 0.0, 0.0][len(params):]
        if kind == _abi.REWARD_CLIPPED_LOG_RETURN and not p[1] <= p[2]:
            raise ValueError("clipped_log_return needs lo <= hi")
        return kind, p[0], p[1], p[2]
    if isinstance(reward_function, str):
        if reward_function not in _REWARD_BY_NAME:
            raise ValueError(f"unknown reward spec {reward_function!r}; known: {sorted(_REWARD_BY_NAME)}")
        return _REWARD_BY_NAME[reward_function], 1.0, 0.0, 0.0
    if reward_function is defaults.basic_reward_function:
        return _abi.REWARD_LOG_RETURN, 1.0, 0.0, 0.0
    if callable(reward_function):
        return HOST_CALLABLE, 1.0, 0.0, 0.0
    raise TypeError(f"reward_function {reward_function!r} is neither a callable nor a spec")


def make_config(*, n_envs: int, n_static: int, n_datasets: int = 1,
                positions=(0, 1),
                dynamic_feature_functions=("last_position_taken", "real_position"),
                reward_function="basic_reward_function",
                windows=None, trading_fees=0, borrow_interest_rate=0,
                portfolio_initial_value=1000, initial_position="random",
                max_episode_duration="max", autoreset=None,
                episodes_between_dataset_switch: int = 1,
                dyn_persist: bool = False, seed: int = 0, env_id_base: int = 0,
                device: int = 0, envs_per_wave: int = 0,
                nontemporal_obs: int = 3, kernel_variant: int = 0,
                debug_flags: int = 0, affinity_period: int = 0,
                final_obs: bool = False, log_steps: int = 0) -> _abi.GteConfig:
    positions = list(positions)
    if not 0 < len(positions) <= _abi.GTE_MAX_POSITIONS:
        raise ValueError(f"1..{_abi.GTE_MAX_POSITIONS} positions supported")
    # environments.py:106
    assert initial_position in positions or initial_position == "random", (
        "The 'initial_position' parameter must be 'random' or a position "
        "mentionned in the 'position' (default is [0, 1]) parameter.")
    if n_envs <= 0 or n_datasets <= 0 or n_static < 0:
        raise ValueError("n_envs, n_datasets must be > 0 and n_static >= 0")
    if windows is not None and (not isinstance(windows, int) or windows < 1):
        raise ValueError("windows must be None or a positive int")
    if max_episode_duration != "max" and (
            not isinstance(max_episode_duration, int) or max_episode_duration < 2):
        raise ValueError("max_episode_duration must be 'max' or an int >= 2")
    if episodes_between_dataset_switch < 1:
        raise ValueError("episodes_between_dataset_switch must be >= 1")
    kinds = resolve_dynamic_features(dynamic_feature_functions)
    # a user's callable keeps a device placeholder: the kernel's value for that column is
    # overwritten after every launch (gte_set_dynamic_features) / the kernel's log-return is
    # replaced by the callable's value
    kinds = [_abi.DYN_LAST_POSITION if k == HOST_CALLABLE else k for k in kinds]
    rk, rp0, rp1, rp2 = resolve_reward(reward_function)
    if rk == HOST_CALLABLE:
        rk = _abi.REWARD_LOG_RETURN
    if autoreset not in _AUTORESET_BY_NAME:
        raise ValueError(f"autoreset must be one of {list(_AUTORESET_BY_NAME)}")

    cfg = _abi.GteConfig()
    cfg.abi_version = _abi.GTE_ABI_VERSION
    cfg.struct_bytes = C.sizeof(_abi.GteConfig)
    cfg.device = device
    cfg.n_envs = n_envs
    cfg.n_datasets = n_datasets
    cfg.n_static = n_static
    cfg.n_dyn = len(kinds)
    for i, k in enumerate(kinds):
        cfg.dyn_kind[i] = k
    cfg.window = 0 if windows is None else int(windows)
    cfg.n_positions = len(positions)
    for i, p in enumerate(positions):
        cfg.positions[i] = float(p)
    cfg.trading_fees = float(trading_fees)
    cfg.borrow_interest_rate = float(borrow_interest_rate)
    cfg.portfolio_initial_value = float(portfolio_initial_value)  # :104
    cfg.initial_position_index = (
        -1 if initial_position == "random" else positions.index(initial_position))
    cfg.max_episode_duration = 0 if max_episode_duration == "max" else int(max_episode_duration)
    cfg.reward_kind = rk
    cfg.reward_param0 = rp0
    cfg.reward_param1 = rp1
    cfg.reward_param2 = rp2
    cfg.autoreset = _AUTORESET_BY_NAME[autoreset]
    cfg.episodes_between_dataset_switch = int(episodes_between_dataset_switch)
    cfg.dyn_persist = int(bool(dyn_persist))
    cfg.seed = int(seed) & (2**64 - 1)
    cfg.env_id_base = int(env_id_base)
    cfg.envs_per_wave = int(envs_per_wave)
    cfg.nontemporal_obs = int(nontemporal_obs)  # 0 plain, 1 nt, 2 sc1, 3 automatic
    cfg.kernel_variant = int(kernel_variant)
    cfg.debug_flags = int(debug_flags)
    cfg.affinity_period = int(affinity_period)
    if final_obs and _AUTORESET_BY_NAME[autoreset] != _abi.AUTORESET_SAME_STEP:
        raise ValueError("final_obs needs autoreset='same_step'")
    cfg.final_obs = int(bool(final_obs))
    if log_steps < 0:
        raise ValueError("log_steps must be >= 0")
    cfg.log_steps = int(log_steps)
    return cfg
