"""ctypes mirror of include/gte.h and the loader of libgte.so.

The product path has no CPU fallback: :func:`load_library` raises when the HIP
library has not been built, and every device entry point of the library
returns GTE_ERR_NO_DEVICE on a host without a gfx950 GPU, which
:func:`check` turns into :class:`GteError`.
"""
from __future__ import annotations

import ctypes as C
import os

GTE_ABI_VERSION = 2
GTE_MAX_POSITIONS = 32
GTE_MAX_DYN = 4
GTE_COMM_ID_BYTES = 128

GTE_OK = 0
GTE_ERR_INVALID = -1
GTE_ERR_NO_DEVICE = -2
GTE_ERR_HIP = -3
GTE_ERR_STATE = -4
GTE_ERR_OOM = -5

DYN_LAST_POSITION = 0
DYN_REAL_POSITION = 1

REWARD_LOG_RETURN = 0
REWARD_SCALED_LOG_RETURN = 1
REWARD_CLIPPED_LOG_RETURN = 2

AUTORESET_DISABLED = 0
AUTORESET_NEXT_STEP = 1
AUTORESET_SAME_STEP = 2


class GteError(RuntimeError):
    """A libgte call failed (status code + the library's message)."""

    def __init__(self, status: int, message: str):
        super().__init__(f"libgte error {status}: {message}")
        self.status = status


class GteConfig(C.Structure):
    """struct gte_config (include/gte.h)."""

    _fields_ = [
        ("abi_version", C.c_int32),
        ("struct_bytes", C.c_int32),
        ("device", C.c_int32),
        ("n_envs", C.c_int32),
        ("n_datasets", C.c_int32),
        ("n_static", C.c_int32),
        ("n_dyn", C.c_int32),
        ("dyn_kind", C.c_int32 * GTE_MAX_DYN),
        ("window", C.c_int32),
        ("n_positions", C.c_int32),
        ("positions", C.c_double * GTE_MAX_POSITIONS),
        ("trading_fees", C.c_double),
        ("borrow_interest_rate", C.c_double),
        ("portfolio_initial_value", C.c_double),
        ("initial_position_index", C.c_int32),
        ("max_episode_duration", C.c_int32),
        ("reward_kind", C.c_int32),
        ("autoreset", C.c_int32),
        ("reward_param0", C.c_double),
        ("reward_param1", C.c_double),
        ("reward_param2", C.c_double),
        ("episodes_between_dataset_switch", C.c_int32),
        ("dyn_persist", C.c_int32),
        ("seed", C.c_uint64),
        ("env_id_base", C.c_int64),
        ("envs_per_wave", C.c_int32),
        ("nontemporal_obs", C.c_int32),
        ("kernel_variant", C.c_int32),
        ("debug_flags", C.c_int32),
        ("affinity_period", C.c_int32),
        ("log_steps", C.c_int32),
        ("reserved2", C.c_int32),
        ("final_obs", C.c_int32),
    ]


class GteEnvSnapshot(C.Structure):
    """struct gte_env_snapshot (include/gte.h)."""
    _fields_ = [(n, C.c_int32) for n in ("idx", "step", "position_index", "dataset_index",
                                         "start_idx", "episode", "needs_reset", "terminated",
                                         "truncated", "reserved")] + \
               [(n, C.c_double) for n in ("asset", "fiat", "interest_asset", "interest_fiat",
                                          "portfolio_valuation", "real_position", "reward")]


# numpy view of an array of gte_env_snapshot (gte_read_envs)
SNAPSHOT_DTYPE = [(n, "<i4") for n in ("idx", "step", "position_index", "dataset_index",
                                       "start_idx", "episode", "needs_reset", "terminated",
                                       "truncated", "reserved")] + \
                 [(n, "<f8") for n in ("asset", "fiat", "interest_asset", "interest_fiat",
                                       "portfolio_valuation", "real_position", "reward")]


class GteRolloutBufs(C.Structure):
    """struct gte_rollout_bufs (include/gte.h): optional per-step result arrays."""
    _fields_ = [("obs", C.c_void_p), ("reward", C.c_void_p), ("reward64", C.c_void_p),
                ("terminated", C.c_void_p), ("truncated", C.c_void_p), ("valuation", C.c_void_p)]


class GteOutputs(C.Structure):
    """struct gte_outputs: device pointers, kept as integers."""

    _fields_ = [
        ("obs", C.c_void_p),
        ("reward", C.c_void_p),
        ("reward64", C.c_void_p),
        ("terminated", C.c_void_p),
        ("truncated", C.c_void_p),
        ("term_count", C.c_void_p),
        ("term_ids", C.c_void_p),
        ("obs_elems_per_env", C.c_int64),
        ("term_slot", C.c_int32),
        ("reserved0", C.c_int32),
        ("final_obs", C.c_void_p),
    ]


class GteStateView(C.Structure):
    """struct gte_state_view: device pointers, kept as integers."""

    _fields_ = [
        ("idx", C.c_void_p),
        ("step", C.c_void_p),
        ("position_index", C.c_void_p),
        ("dataset_index", C.c_void_p),
        ("start_idx", C.c_void_p),
        ("episode", C.c_void_p),
        ("needs_reset", C.c_void_p),
        ("asset", C.c_void_p),
        ("fiat", C.c_void_p),
        ("interest_asset", C.c_void_p),
        ("interest_fiat", C.c_void_p),
        ("portfolio_valuation", C.c_void_p),
        ("real_position", C.c_void_p),
    ]


class GteLogView(C.Structure):
    """struct gte_log_view: device pointers of the trajectory log, kept as integers."""

    _fields_ = [("idx", C.c_void_p), ("step", C.c_void_p), ("position_index", C.c_void_p),
                ("dataset_index", C.c_void_p), ("portfolio_valuation", C.c_void_p),
                ("real_position", C.c_void_p), ("reward", C.c_void_p), ("flags", C.c_void_p),
                ("rows", C.c_int64), ("L", C.c_int32), ("N", C.c_int32),
                ("asset", C.c_void_p), ("fiat", C.c_void_p), ("interest_asset", C.c_void_p),
                ("interest_fiat", C.c_void_p), ("env_stride", C.c_int64), ("row_stride", C.c_int64)]


class GteLogBatch(C.Structure):
    """struct gte_log_batch: host pointers into the library's pinned buffer (gte_read_log_envs)."""

    _fields_ = [("n_ids", C.c_int32), ("max_rows", C.c_int32), ("n_rows", C.c_void_p),
                ("idx", C.c_void_p), ("step", C.c_void_p), ("position_index", C.c_void_p),
                ("dataset_index", C.c_void_p), ("portfolio_valuation", C.c_void_p),
                ("real_position", C.c_void_p), ("reward", C.c_void_p), ("asset", C.c_void_p),
                ("fiat", C.c_void_p), ("interest_asset", C.c_void_p), ("interest_fiat", C.c_void_p),
                ("flags", C.c_void_p)]


#: dtype of every log array
LOG_DTYPES = {"idx": "int32", "step": "int32", "position_index": "int32", "dataset_index": "int32",
              "portfolio_valuation": "float64", "real_position": "float64", "reward": "float64",
              "flags": "uint8", "asset": "float64", "fiat": "float64",
              "interest_asset": "float64", "interest_fiat": "float64"}

#: dtype of every gte_state_view member, in declaration order
STATE_DTYPES = {
    "idx": "int32", "step": "int32", "position_index": "int32",
    "dataset_index": "int32", "start_idx": "int32", "episode": "int32",
    "needs_reset": "int32", "asset": "float64", "fiat": "float64",
    "interest_asset": "float64", "interest_fiat": "float64",
    "portfolio_valuation": "float64", "real_position": "float64",
}

_P = C.POINTER

#: every symbol include/gte.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "gte_create": (C.c_int, [_P(GteConfig), _P(C.c_void_p)]),
    "gte_upload_dataset": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int64]),
    "gte_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gte_set_autoreset_injection": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p,
                                               C.c_void_p, C.c_void_p]),
    "gte_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "gte_add_limit_orders": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gte_get_log": (C.c_int, [C.c_void_p, _P(GteLogView)]),
    "gte_read_log_portfolio": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 4
                               + [_P(C.c_int32)]),
    "gte_read_log_envs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                    _P(GteLogBatch)]),
    "gte_set_log_reward": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gte_apply_reward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "gte_get_final_state": (C.c_int, [C.c_void_p, _P(GteStateView)]),
    "gte_set_dynamic_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "gte_set_dynamic_columns": (C.c_int, [C.c_void_p, _P(C.c_void_p), _P(C.c_int32)]),
    "gte_read_log": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 8 + [_P(C.c_int32)]),
    "gte_get_outputs": (C.c_int, [C.c_void_p, _P(GteOutputs)]),
    "gte_get_state": (C.c_int, [C.c_void_p, _P(GteStateView)]),
    "gte_bind_outputs": (C.c_int, [C.c_void_p, _P(GteOutputs)]),
    "gte_read_env": (C.c_int, [C.c_void_p, C.c_int32, _P(GteEnvSnapshot), C.c_void_p]),
    "gte_read_envs": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "gte_read_envs_view": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                     _P(C.c_void_p), _P(C.c_void_p)]),
    "gte_rollout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, _P(GteRolloutBufs)]),
    "gte_bind_returns": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gte_comm_unique_id": (C.c_int, [C.c_void_p]),
    "gte_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "gte_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32]),
    "gte_allgather_returns": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, _P(C.c_void_p)]),
    "gte_allgather_obs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "gte_comm_wait": (C.c_int, [C.c_void_p, C.c_int32]),
    "gte_comm_synchronize": (C.c_int, [C.c_void_p]),
    "gte_comm_destroy": (C.c_int, [C.c_void_p]),
    "gte_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gte_use_own_stream": (C.c_int, [C.c_void_p]),
    "gte_synchronize": (C.c_int, [C.c_void_p]),
    "gte_timer_start": (C.c_int, [C.c_void_p]),
    "gte_timer_stop": (C.c_int, [C.c_void_p, _P(C.c_float)]),
    "gte_read_obs": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "gte_copy_to_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "gte_get_launch_info": (C.c_int, [C.c_void_p, _P(C.c_int32), _P(C.c_int32),
                                      _P(C.c_int32), _P(C.c_int32)]),
    "gte_destroy": (None, [C.c_void_p]),
    "gte_last_error": (C.c_char_p, []),
    "gte_abi_version": (C.c_int, []),
    "gte_device_count": (C.c_int, []),
}

_HERE = os.path.dirname(os.path.abspath(__file__))
# GTE_LIBRARY: another BUILD of libgte to load instead (A/B builds of the kernels, tools/)
LIB_PATH = os.environ.get("GTE_LIBRARY") or os.path.join(_HERE, "csrc", "libgte.so")
_lib = None


def load_library(path: str | None = None) -> C.CDLL:
    """Load libgte.so (the HIP build).  Fails loudly: there is no fallback."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError(
            f"{p} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
            "gym-trading-env_amd/csrc`). There is no CPU fallback for the env.")
    # PyTorch-ROCm bundles its own libamdhip64.so.7; libgte needs the same SONAME.  Two
    # HIP runtimes in one process do not share devices, so when torch is installed load
    # it FIRST: the dynamic linker then binds libgte to the runtime torch uses.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.gte_abi_version() != GTE_ABI_VERSION:
        raise ImportError(f"{p}: ABI {lib.gte_abi_version()} != {GTE_ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


def check(lib: C.CDLL, status: int) -> None:
    if status != GTE_OK:
        msg = lib.gte_last_error()
        raise GteError(status, msg.decode() if msg else "unknown")
