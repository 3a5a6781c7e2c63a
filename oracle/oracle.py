"""ctypes wrapper of oracle/libgte_oracle.so — ORACLE, test infrastructure only.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by the product package (gym-trading-env_amd/).  See gte_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

import gym_trading_env_amd  # noqa: F401  (struct gte_config mirror + config builder)
from gym_trading_env_amd._abi import GteConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgte_oracle.so")
_lib = None

_I32 = ("idx", "step", "position_index", "dataset_index", "start_idx", "episode",
        "needs_reset")
_F64 = ("asset", "fiat", "interest_asset", "interest_fiat", "portfolio_valuation",
        "real_position")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gte_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "gte.h")
    stale = (not os.path.exists(_SO)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_SO) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libgte_oracle.so"])
    return _SO


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        l = C.CDLL(_SO)
        l.gto_create.restype = C.c_void_p
        l.gto_create.argtypes = [C.POINTER(GteConfig)]
        l.gto_upload_dataset.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]
        l.gto_reset.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        l.gto_upload_high_low.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        l.gto_add_limit_orders.argtypes = [C.c_void_p] * 4
        l.gto_set_autoreset_injection.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 3
        l.gto_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        l.gto_destroy.argtypes = [C.c_void_p]
        l.gto_term_count.argtypes = [C.c_void_p]
        l.gto_term_count.restype = C.c_int32
        for n in _I32 + _F64 + ("final_obs", "obs", "reward", "reward64", "terminated", "truncated", "term_ids"):
            f = getattr(l, "gto_get_" + n)
            f.restype = C.c_void_p
            f.argtypes = [C.c_void_p]
        l.gto_portfolio_trade.argtypes = [C.c_void_p] + [C.c_double] * 5 + [C.c_void_p] * 2
        l.gto_philox.argtypes = [C.c_void_p] * 3
        _lib = l
    return _lib


def _opt(a, dtype):
    if a is None:
        return None, None
    a = np.ascontiguousarray(np.asarray(a, dtype=dtype))
    return a, a.ctypes.data


class OracleEnv:
    """Batch of N reference-semantics environments on the CPU (fp64, scalar)."""

    def __init__(self, cfg: GteConfig, datasets):
        """datasets: list of (feat f32 [T, F_obs], close f64 [T][, high f64 [T], low f64 [T]])."""
        self._l = lib()
        self.cfg = cfg
        self.N = cfg.n_envs
        self.W = cfg.window if cfg.window > 0 else 1
        self.F = cfg.n_static + cfg.n_dyn
        self._h = self._l.gto_create(C.byref(cfg))
        if not self._h:
            raise RuntimeError("gto_create failed (ABI mismatch?)")
        assert len(datasets) == cfg.n_datasets
        for d, ds in enumerate(datasets):
            feat, close = ds[0], ds[1]
            feat = np.ascontiguousarray(feat, dtype=np.float32)
            close = np.ascontiguousarray(close, dtype=np.float64)
            assert feat.shape == (close.shape[0], self.F), (feat.shape, close.shape, self.F)
            rc = self._l.gto_upload_dataset(self._h, d, feat.ctypes.data, close.ctypes.data,
                                            close.shape[0])
            assert rc == 0
            if len(ds) >= 4 and ds[2] is not None:
                hi = np.ascontiguousarray(ds[2], dtype=np.float64)
                lo = np.ascontiguousarray(ds[3], dtype=np.float64)
                assert self._l.gto_upload_high_low(self._h, d, hi.ctypes.data, lo.ctypes.data) == 0

    def add_limit_orders(self, pos_index, limit, persistent=None):
        a = np.ascontiguousarray(pos_index, dtype=np.int32)
        b = np.ascontiguousarray(limit, dtype=np.float64)
        c = None if persistent is None else np.ascontiguousarray(persistent, dtype=np.uint8)
        assert a.shape == b.shape == (self.N,)
        rc = self._l.gto_add_limit_orders(self._h, a.ctypes.data, b.ctypes.data,
                                          None if c is None else c.ctypes.data)
        assert rc == 0

    def _view(self, name, dtype, shape):
        p = getattr(self._l, "gto_get_" + name)(self._h)
        n = int(np.prod(shape))
        buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)

    def reset(self, mask=None, inj_idx=None, inj_pos=None, inj_ds=None):
        m, mp = _opt(mask, np.uint8)
        a, ap = _opt(inj_idx, np.int32)
        b, bp = _opt(inj_pos, np.int32)
        c, cp = _opt(inj_ds, np.int32)
        rc = self._l.gto_reset(self._h, mp, ap, bp, cp)
        assert rc == 0, "gto_reset failed (dataset missing?)"
        return self.obs

    def set_autoreset_injection(self, inj_idx=None, inj_pos=None, inj_ds=None):
        arrs = [x for x in (inj_idx, inj_pos, inj_ds) if x is not None]
        n = 0 if not arrs else np.asarray(arrs[0]).reshape(self.N, -1).shape[1]
        a, ap = _opt(inj_idx, np.int32)
        b, bp = _opt(inj_pos, np.int32)
        c, cp = _opt(inj_ds, np.int32)
        assert self._l.gto_set_autoreset_injection(self._h, n, ap, bp, cp) == 0

    def step(self, actions, threads: int = 1):
        a = np.ascontiguousarray(np.asarray(actions, dtype=np.int32))
        assert a.shape == (self.N,)
        assert self._l.gto_step(self._h, a.ctypes.data, threads) == 0
        return self.obs, self.reward64, self.terminated, self.truncated

    @property
    def obs(self):
        shape = (self.N, self.W, self.F) if self.cfg.window > 0 else (self.N, self.F)
        return self._view("obs", np.float32, shape)

    @property
    def final_obs(self):
        shape = (self.N, self.W, self.F) if self.cfg.window > 0 else (self.N, self.F)
        return self._view("final_obs", np.float32, shape)

    reward = property(lambda s: s._view("reward", np.float32, (s.N,)))
    reward64 = property(lambda s: s._view("reward64", np.float64, (s.N,)))
    terminated = property(lambda s: s._view("terminated", np.uint8, (s.N,)))
    truncated = property(lambda s: s._view("truncated", np.uint8, (s.N,)))

    @property
    def term_ids(self):
        n = self._l.gto_term_count(self._h)
        return self._view("term_ids", np.int32, (self.N,))[:n]

    def state(self):
        out = {n: self._view(n, np.int32, (self.N,)) for n in _I32}
        out.update({n: self._view(n, np.float64, (self.N,)) for n in _F64})
        return out

    def close(self):
        if self._h:
            self._l.gto_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def portfolio_trade(state4, position, price, fees, rate, next_price):
    """One Portfolio.trade_to_position + update_interest + valuation/real_position."""
    s = np.array(state4, dtype=np.float64)
    val = C.c_double()
    rp = C.c_double()
    lib().gto_portfolio_trade(s.ctypes.data, position, price, fees, rate, next_price,
                              C.addressof(val), C.addressof(rp))
    return s, val.value, rp.value


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    o = np.zeros(4, dtype=np.uint32)
    lib().gto_philox(c.ctypes.data, k.ctypes.data, o.ctypes.data)
    return o
