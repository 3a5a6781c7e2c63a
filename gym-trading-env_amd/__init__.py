"""MI355X-native batched drop-in for Gym-Trading-Env's step()/reset() hot path.

Public surface (same names as src/gym_trading_env/environments.py):
``TradingEnv``, ``MultiDatasetTradingEnv``, ``basic_reward_function``,
``dynamic_feature_last_position_taken``, ``dynamic_feature_real_position``;
plus ``BatchedTradingEnv``, the N-environment form the HIP kernels serve.
Heavy imports (torch, the HIP library) happen on first use, not here.
"""
from . import _abi, config, staging  # noqa: F401
from ._abi import GteError  # noqa: F401

__all__ = ["TradingEnv", "MultiDatasetTradingEnv", "BatchedTradingEnv", "SB3TradingVecEnv", "GteError",
           "basic_reward_function", "dynamic_feature_last_position_taken",
           "dynamic_feature_real_position"]

_LAZY = {
    "BatchedTradingEnv": "batched", "TradingEnv": "envs",
    "MultiDatasetTradingEnv": "envs", "basic_reward_function": "defaults",
    "dynamic_feature_last_position_taken": "defaults",
    "dynamic_feature_real_position": "defaults", "History": "history",
    "SB3TradingVecEnv": "sb3", "BatchedHistory": "batched_history", "DeviceArray": "device_array",
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod = importlib.import_module(f"{__name__}.{_LAZY[name]}")
        return getattr(mod, name)
    raise AttributeError(name)


def register_gymnasium_ids():
    """Register 'TradingEnv' and 'MultiDatasetTradingEnv' like the reference's
    __init__.py:3-14 (same ids and flags), plus — where Gymnasium supports it — a vector entry
    point so that the reference's `gym.make_vec("TradingEnv", num_envs=N, ...)`
    (examples/example_vectorized_environment.py:44) yields ONE BatchedTradingEnv.  No-op when
    gymnasium is not installed (it is not, in the build and test images: this wiring is written
    against Gymnasium's documented API and is unpinned)."""
    try:
        from gymnasium.envs.registration import register, registry
    except Exception:
        return False
    from . import envs

    def make_batch(num_envs=1, **kwargs):  # gym.make_vec(id, num_envs=N, ...) -> one batched env
        from .batched import BatchedTradingEnv
        if "dataset_dir" in kwargs:
            return BatchedTradingEnv.from_dataset_dir(kwargs.pop("dataset_dir"), num_envs, **kwargs)
        return BatchedTradingEnv(kwargs.pop("df"), num_envs, **kwargs)

    for env_id, cls in (("TradingEnv", envs.TradingEnv),
                        ("MultiDatasetTradingEnv", envs.MultiDatasetTradingEnv)):
        if env_id in registry:
            continue
        try:  # Gymnasium >= 1.0: `gym.make_vec` builds the HBM-resident batch, not N Python envs
            register(id=env_id, entry_point=cls, vector_entry_point=make_batch,
                     disable_env_checker=True, order_enforce=False)
        except TypeError:  # older Gymnasium: no vector entry point
            register(id=env_id, entry_point=cls, disable_env_checker=True, order_enforce=False)
    return True


register_gymnasium_ids()
