"""DataFrame -> dense arrays, the host half of TradingEnv._set_df.

Follows src/gym_trading_env/environments.py:128-143: feature columns are the
columns whose name contains "feature" (:130), in DataFrame order; one zero
column per dynamic feature is appended (:135-138); the observation table is
float32 row-major [T, F_obs] (:141) and prices are float64 from "close" (:143).
The arrays are what gte_upload_dataset puts in HBM.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


@dataclass
class StagedDataset:
    feat: np.ndarray            # f32 [T, F_obs], trailing n_dyn columns zero
    close: np.ndarray           # f64 [T]
    high: np.ndarray | None     # f64 [T] or None
    low: np.ndarray | None
    n_static: int
    n_dyn: int
    feature_columns: list = field(default_factory=list)
    info_columns: list = field(default_factory=list)
    info_array: np.ndarray | None = None   # [T, I] like _info_array (:142)
    index: np.ndarray | None = None        # df.index.values (dates, :189)
    name: str = "Stock"

    @property
    def T(self) -> int:
        return int(self.close.shape[0])

    @property
    def n_obs(self) -> int:
        return self.n_static + self.n_dyn


def stage_dataframe(df, n_dyn: int = 2, name: str = "Stock") -> StagedDataset:
    """`_set_df` (environments.py:128-143) for one pandas DataFrame."""
    if "close" not in df.columns:
        raise KeyError("close")  # the reference fails on df["close"] (:143)
    feature_columns = [col for col in df.columns if "feature" in col]  # :130
    # :131 builds the info column list through a set; order is unspecified there,
    # here it is the DataFrame order (close included once).
    info_columns = [col for col in df.columns if col not in feature_columns]
    n_static = len(feature_columns)
    T = len(df)
    feat = np.zeros((T, n_static + n_dyn), dtype=np.float32)
    if n_static:
        feat[:, :n_static] = np.asarray(df[feature_columns], dtype=np.float32)
    close = np.ascontiguousarray(np.asarray(df["close"], dtype=np.float64))
    high = (np.ascontiguousarray(np.asarray(df["high"], dtype=np.float64))
            if "high" in df.columns else None)
    low = (np.ascontiguousarray(np.asarray(df["low"], dtype=np.float64))
           if "low" in df.columns else None)
    return StagedDataset(
        feat=feat, close=close, high=high, low=low, n_static=n_static, n_dyn=n_dyn,
        feature_columns=feature_columns + [f"dynamic_feature__{i}" for i in range(n_dyn)],
        info_columns=info_columns, info_array=np.array(df[info_columns]),
        index=df.index.values, name=name)


def stage_arrays(features, close, n_dyn: int = 2, high=None, low=None,
                 name: str = "Stock") -> StagedDataset:
    """Stage plain arrays: features [T, F_s] (any float dtype) and close [T]."""
    features = np.asarray(features)
    close = np.ascontiguousarray(np.asarray(close, dtype=np.float64))
    if features.ndim != 2 or features.shape[0] != close.shape[0]:
        raise ValueError("features must be [T, F_s] with T == len(close)")
    T, n_static = features.shape
    feat = np.zeros((T, n_static + n_dyn), dtype=np.float32)
    feat[:, :n_static] = features.astype(np.float32, copy=False)
    as64 = lambda a: None if a is None else np.ascontiguousarray(np.asarray(a, np.float64))
    return StagedDataset(feat=feat, close=close, high=as64(high), low=as64(low),
                         n_static=n_static, n_dyn=n_dyn, name=name)


def check_episode_geometry(T: int, windows, max_episode_duration) -> None:
    """The conditions under which reset/step of the reference are well defined.

    reset draws `np.random.randint(low=idx0, high=T - max_dur - idx0)` (:174-177)
    which raises ValueError unless high > low; a window needs W rows."""
    idx0 = 0 if windows is None else windows - 1
    if T < idx0 + 2:
        raise ValueError(f"dataset of {T} rows is too short for windows={windows}")
    if max_episode_duration != "max":
        high = T - int(max_episode_duration) - idx0
        if high <= idx0:
            raise ValueError(
                f"low >= high: a dataset of {T} rows cannot host episodes of "
                f"{max_episode_duration} steps with windows={windows}")
