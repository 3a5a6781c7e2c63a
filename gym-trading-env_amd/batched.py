"""BatchedTradingEnv — N reference-semantics TradingEnvs stepped by one HIP launch.

Host-side mirror of the reference's `TradingEnv` surface
(src/gym_trading_env/environments.py:79-125 constructor, :163 reset, :233 step)
for a batch: same argument names and meaning, same asserts, `step(actions)`
returning `(obs, reward, terminated, truncated, info)`.  All arithmetic happens
in libgte's kernels (csrc/); this file only stages data, marshals pointers and
wraps device buffers.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi, spaces, staging
from .config import make_config

_NP = {"int32": np.int32, "float64": np.float64}


def _as_staged(ds, n_dyn, name="Stock"):
    if isinstance(ds, staging.StagedDataset):
        return ds
    if isinstance(ds, tuple) and len(ds) in (2, 4):  # (features, close[, high, low])
        return staging.stage_arrays(ds[0], ds[1], n_dyn=n_dyn, name=name,
                                    high=ds[2] if len(ds) == 4 else None,
                                    low=ds[3] if len(ds) == 4 else None)
    return staging.stage_dataframe(ds, n_dyn=n_dyn, name=name)  # a pandas DataFrame


class LazyInfo(dict):
    """`info` of a batched step: struct-of-arrays view of what the reference logs in
    `History` each step and returns as `history[-1]` (environments.py:253-264, :272), fetched
    from HBM only on access, one host array of length N per key.

    Keys are History's flattened column names (docs/source/history.rst:18-46,
    docs/source/vectorize_env.rst:25-33): idx, step, date, position_index, position,
    real_position, data_<every non-feature column of the DataFrame>, portfolio_valuation,
    portfolio_distribution_{asset,fiat,borrowed_asset,borrowed_fiat,interest_asset,
    interest_fiat}, reward — plus dataset_index.  `_key` is Gymnasium's presence mask (all True).
    With ``autoreset="same_step", final_obs=True`` the envs whose episode ended in this step
    also get Gymnasium's `final_observation` / `final_info` (object arrays, `None` elsewhere)
    with their `_final_observation` / `_final_info` masks."""

    def __init__(self, env):
        super().__init__()
        self._env = env
        self._KEYS = tuple(env.info_keys)
        self._final = bool(env.cfg.final_obs)

    def keys(self):
        return list(self._KEYS) + (["final_observation", "final_info"] if self._final else [])

    def __contains__(self, k):
        return k in self._KEYS or (self._final and k in ("final_observation", "final_info",
                                                         "_final_observation", "_final_info")) \
            or (isinstance(k, str) and k.startswith("_") and k[1:] in self._KEYS)

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def __missing__(self, key):
        e = self._env
        if self._final and key in ("final_observation", "final_info", "_final_observation",
                                   "_final_info"):
            self.update(e._final_entries())
            return dict.__getitem__(self, key)
        if key.startswith("_") and key[1:] in self._KEYS:
            # Gymnasium's vector-env convention (docs/source/vectorize_env.rst:25-33): `_key`
            # marks the envs for which `key` is present — here always all of them
            v = np.ones(e.num_envs, dtype=bool)
        elif key in self._KEYS:
            v = e._info_value(key, e.state, e.read_output("reward64") if key == "reward" else None)
        else:
            raise KeyError(key)
        self[key] = v
        return v


try:  # soft dependency: lets `isinstance(env, gymnasium.vector.VectorEnv)` hold
    from gymnasium.vector import VectorEnv as _VectorEnvBase  # type: ignore
except Exception:
    _VectorEnvBase = object


class BatchedTradingEnv(_VectorEnvBase):
    """N independent trading environments resident in HBM.

    Follows Gymnasium's vector-env conventions (docs/source/vectorize_env.rst:17-33):
    `num_envs`, `single_observation_space` / `single_action_space`, batched
    `observation_space` / `action_space`, `reset() -> (obs, info)`, `step(actions) ->
    (obs, rewards, terminations, truncations, infos)` with dict-of-arrays infos and `_key`
    masks, auto-reset in next-step (Gymnasium >= 1.0) or same-step mode.

    Parameters mirror `TradingEnv.__init__` (environments.py:79-93); the extra ones:

    :param df: one dataset or a list of datasets: pandas DataFrames (feature columns
        contain "feature", a "close" column), ``StagedDataset`` objects, or
        ``(features[T, F_s], close[T][, high[T], low[T]])`` tuples.  Several datasets give the
        `MultiDatasetTradingEnv` behaviour (:365-400) with all of them resident.
    :param num_envs: N.
    :param autoreset: None/"disabled", "next_step" (Gymnasium >= 1.0) or "same_step".
    :param output: "torch" — observations/rewards/flags stay on the device as torch
        tensors (zero-copy views of the buffers the kernel writes); "numpy" — copied to
        host each step (compatibility mode).
:param final_obs: with ``autoreset="same_step"``: keep the terminal observation of
        every env that ends (Gymnasium's ``final_observation`` / SB3's
        ``terminal_observation``); read it with :meth:`final_observations`.
:param log_steps: L > 0 keeps the last L steps of every env in a device trajectory
        log; :meth:`history` turns one env's episode into a `History` (the object the
        reference hands to custom reward / metric functions) and `add_metric` functions
        are evaluated by :meth:`episode_metrics`.
    :param dyn_persist: keep the per-env dynamic-feature column across episodes like
        the reference's in-place write into `_obs_array` (:153-154); costs
        N*T*n_dyn*4 bytes of HBM.
    """

    metadata = {"render_modes": ["logs"]}

    def __init__(self, df, num_envs: int, positions=(0, 1),
                 dynamic_feature_functions=("last_position_taken", "real_position"),
                 reward_function="basic_reward_function", windows=None, trading_fees=0,
                 borrow_interest_rate=0, portfolio_initial_value=1000,
                 initial_position="random", max_episode_duration="max", verbose=1,
                 name="Stock", render_mode="logs", *, autoreset="next_step",
                 episodes_between_dataset_switch=1, dyn_persist=False, seed=0,
                 env_id_base=0, device=0, output="torch", envs_per_wave=0,
                 nontemporal_obs=3, kernel_variant=0, library_path=None, debug_flags=0,
                 affinity_period=0, final_obs=False, log_steps=0, return_slots=1, copy=True):
        assert render_mode is None or render_mode in self.metadata["render_modes"]
        if output not in ("torch", "numpy"):
            raise ValueError("output must be 'torch' or 'numpy'")
        self.positions = list(positions)
        self.windows = windows
        self.verbose, self.name, self.render_mode = verbose, name, render_mode
        self.num_envs = int(num_envs)
        self.output = output
        # return_slots=K > 1 (torch output): step t writes reward/terminated/truncated into
        # row t % K of one [K, 6N] buffer, so a collective still reading older returns is not
        # overwritten by the next K-1 steps, and runs of consecutive steps are contiguous
        # blocks (distributed.ReturnPipeline gathers them a block at a time)
        # copy=False (output="numpy", Gymnasium's vector-env convention): step()/reset() return
        # views of the library's pinned staging buffer, overwritten by the next call — no
        # 168 MB allocation + copy per step at the headline shape
        self.copy = bool(copy)
        # verbose > 0 with a trajectory log the USER asked for: the reference's episode-end line
        # (environments.py:269-271, :289-294) for the envs that finished — at most
        # `verbose_max_lines` per report and one report per `verbose_interval` seconds (a report
        # costs a device synchronisation; 0 = after every step)
        self.verbose_max_lines = 16
        self.verbose_interval = 1.0
        self._last_report = 0.0
        self._user_log = int(log_steps) > 0
        self.return_slots = int(return_slots)
        if self.return_slots < 1 or (self.return_slots > 1 and output != "torch"):
            raise ValueError("return_slots must be >= 1 (and > 1 only with output='torch')")
        mode = autoreset or "disabled"
        try:  # Gymnasium >= 1.0 wrappers read an AutoresetMode enum here (unpinned: not installed here)
            from gymnasium.vector import AutoresetMode  # type: ignore
            mode = {"next_step": AutoresetMode.NEXT_STEP, "same_step": AutoresetMode.SAME_STEP,
                    "disabled": AutoresetMode.DISABLED}[mode]
        except Exception:  # noqa: BLE001
            pass
        self.metadata = dict(self.metadata, autoreset_mode=mode)
        self.closed = False
        self._lib = _abi.load_library(library_path)  # raises if the HIP build is missing
        self._h = C.c_void_p()

        from .config import HOST_CALLABLE, resolve_dynamic_features, resolve_reward
        dyn_kinds = resolve_dynamic_features(dynamic_feature_functions)
        n_dyn = len(dyn_kinds)
        # a user's own callables (anything but this package's default objects / the string and
        # tuple specs) are evaluated after every launch, vectorised over a BatchedHistory
        self._dyn_callables = [(i, f) for i, (k, f) in enumerate(zip(dyn_kinds, dynamic_feature_functions))
                               if k == HOST_CALLABLE]
        self._reward_callable = (reward_function if resolve_reward(reward_function)[0] == HOST_CALLABLE
                                 else None)
        if self._dyn_callables or self._reward_callable is not None:
            if output != "torch":
                raise NotImplementedError("custom reward / dynamic-feature callables in the batch are "
                                          "evaluated on the device: use output='torch'")
            if autoreset == "same_step":
                final_obs = True  # the terminal History row of an env that ends comes from its terminal record
            log_steps = max(int(log_steps), 2)  # the callables read h[..., -1] and h[..., -2]
        raw = list(df) if isinstance(df, (list, tuple)) and not (
            isinstance(df, tuple) and len(df) in (2, 4) and hasattr(df[0], "shape")) else [df]
        self.datasets = [_as_staged(d, n_dyn, name) for d in raw]
        first = self.datasets[0]
        for d in self.datasets:
            if d.n_static != first.n_static or d.n_dyn != n_dyn:
                raise ValueError("all datasets must have the same feature columns")
            staging.check_episode_geometry(d.T, windows, max_episode_duration)

        self.cfg = make_config(
            n_envs=self.num_envs, n_static=first.n_static, n_datasets=len(self.datasets),
            positions=self.positions, dynamic_feature_functions=dynamic_feature_functions,
            reward_function=reward_function, windows=windows, trading_fees=trading_fees,
            borrow_interest_rate=borrow_interest_rate,
            portfolio_initial_value=portfolio_initial_value,
            initial_position=initial_position, max_episode_duration=max_episode_duration,
            autoreset=autoreset, episodes_between_dataset_switch=episodes_between_dataset_switch,
            dyn_persist=dyn_persist, seed=seed, env_id_base=env_id_base, device=device,
            envs_per_wave=envs_per_wave, nontemporal_obs=nontemporal_obs,
            kernel_variant=kernel_variant, debug_flags=debug_flags,
            affinity_period=affinity_period, final_obs=final_obs, log_steps=log_steps)
        self.log_metrics = []
        from .batched_history import history_columns
        self.info_keys = history_columns(self) + ["dataset_index"]
        self._ds_offsets = np.concatenate([[0], np.cumsum([d.T for d in self.datasets])]).astype(np.int64)
        self._host_columns, self._dev_columns = {}, {}
        _abi.check(self._lib, self._lib.gte_create(C.byref(self.cfg), C.byref(self._h)))

        self.n_obs = first.n_static + n_dyn
        self.obs_shape = (self.n_obs,) if windows is None else (int(windows), self.n_obs)
        # environments.py:112-123
        self.single_action_space = spaces.Discrete(len(self.positions))
        self.single_observation_space = spaces.Box(-np.inf, np.inf, shape=self.obs_shape)
        self.action_space = spaces.MultiDiscrete([len(self.positions)] * self.num_envs)
        self.observation_space = spaces.Box(-np.inf, np.inf,
                                            shape=(self.num_envs,) + self.obs_shape)
        for d, s in enumerate(self.datasets):
            self.upload_dataset(d, s)

        self._final_view, self._final_epoch, self._final_tensors = _abi.GteStateView(), -1, {}
        self._logv, self._log_tensors, self._pos_table = _abi.GteLogView(), {}, None
        self._state = _abi.GteStateView()
        self._epoch, self._state_epoch = 0, -1  # state snapshots are taken lazily
        self._snap_epoch, self._snap, self._snap_obs = -1, None, None  # numpy mode, per step
        self._state_tensors = {}  # torch views of the state snapshot (state_tensor)
        self._torch = None
        self._t = {}
        if output == "torch":
            self._bind_torch_outputs()
        self._out = _abi.GteOutputs()
        _abi.check(self._lib, self._lib.gte_get_outputs(self._h, C.byref(self._out)))
        self._was_reset = False
        if _VectorEnvBase is not object:
            # gymnasium installed: run the base initialiser too (unpinned — gymnasium is not in the
            # build / test images; >= 1.0 takes no arguments, 0.29 takes the three below), then put
            # this class's own spaces back
            keep = (self.observation_space, self.action_space)
            try:
                try:
                    _VectorEnvBase.__init__(self)
                except TypeError:
                    _VectorEnvBase.__init__(self, self.num_envs, self.single_observation_space,
                                            self.single_action_space)
            except Exception:  # noqa: BLE001 - never let an optional base class break the env
                pass
            self.observation_space, self.action_space = keep

    @classmethod
    def from_dataset_dir(cls, dataset_dir: str, num_envs: int, *args,
                         preprocess=lambda df: df, episodes_between_dataset_switch: int = 1,
                         **kwargs):
        """The batched `MultiDatasetTradingEnv` (environments.py:365-400): every file matched
        by the glob `dataset_dir` is loaded (`pd.read_pickle`), passed through `preprocess`
        and kept RESIDENT in HBM; each env moves to another dataset every
        `episodes_between_dataset_switch` episodes, visiting all of them once per round in
        random order (== uniform among the least-used ones, :383-388)."""
        import glob
        from pathlib import Path

        import pandas as pd
        paths = glob.glob(dataset_dir)
        if len(paths) == 0:  # :376
            raise FileNotFoundError(f"No dataset found with the path : {dataset_dir}")
        frames = [preprocess(pd.read_pickle(p)) for p in paths]  # the user's own dataset files
        env = cls(frames, num_envs, *args,
                  episodes_between_dataset_switch=episodes_between_dataset_switch, **kwargs)
        env.dataset_pathes = paths
        env.dataset_names = [Path(p).name for p in paths]
        return env

    # -- setup ---------------------------------------------------------------------
    def upload_dataset(self, d: int, s: staging.StagedDataset):
        """`_set_df` (environments.py:128-143): stage one dataset into HBM."""
        feat = np.ascontiguousarray(s.feat, dtype=np.float32)
        close = np.ascontiguousarray(s.close, dtype=np.float64)
        ptr = lambda a: None if a is None else np.ascontiguousarray(a, np.float64).ctypes.data
        _abi.check(self._lib, self._lib.gte_upload_dataset(
            self._h, d, feat.ctypes.data, close.ctypes.data, ptr(s.high), ptr(s.low), s.T))
        self.datasets[d] = s

    def _bind_torch_outputs(self):
        import torch
        if not torch.cuda.is_available():
            raise _abi.GteError(_abi.GTE_ERR_NO_DEVICE, "torch sees no GPU; there is no CPU fallback")
        self._torch = torch
        dev = torch.device("cuda", self.cfg.device)
        N = self.num_envs
        with torch.cuda.device(dev):
            # reward | terminated | truncated live in ONE buffer (distributed.packed_layout)
            # so that a sharded run all-gathers it without a packing kernel
            self._packed_all = torch.zeros((self.return_slots, 6 * N), dtype=torch.uint8,
                                           device=dev)
            self._packed = list(self._packed_all.unbind(0))
            # per slot, made once: the buffer, its (reward, terminated, truncated) views and
            # the three device pointers gte_bind_returns takes
            self._slot_views = []
            for buf in self._packed:
                base = buf.data_ptr()
                self._slot_views.append((
                    buf,
                    (buf[:4 * N].view(torch.float32), buf[4 * N:5 * N].view(torch.bool),
                     buf[5 * N:].view(torch.bool)),
                    (C.c_void_p(base), C.c_void_p(base + 4 * N), C.c_void_p(base + 5 * N))))
            # outputs start bound to row 0 (reset writes there); the first step rotates to it
            self._ret_slot = self.return_slots - 1
            self.packed_returns = self._packed[0]
            self._t = {
                "obs": torch.zeros((N,) + self.obs_shape, dtype=torch.float32, device=dev),
                "reward": self.packed_returns[:4 * N].view(torch.float32),
                "reward64": torch.zeros(N, dtype=torch.float64, device=dev),
                "terminated": self.packed_returns[4 * N:5 * N].view(torch.bool),
                "truncated": self.packed_returns[5 * N:].view(torch.bool),
                "term_count": torch.zeros(2, dtype=torch.int32, device=dev),  # two slots
                "term_ids": torch.zeros(N, dtype=torch.int32, device=dev),
            }
            if self.cfg.final_obs:
                self._t["final_obs"] = torch.zeros((N,) + self.obs_shape, dtype=torch.float32,
                                                   device=dev)
            torch.cuda.synchronize(dev)
            b = _abi.GteOutputs()
            for k, t in self._t.items():
                setattr(b, k, t.data_ptr())
            _abi.check(self._lib, self._lib.gte_bind_outputs(self._h, C.byref(b)))
            # run on torch's current stream so torch ops and env launches are ordered
            _abi.check(self._lib, self._lib.gte_set_stream(
                self._h, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))

    # -- data movement helpers -------------------------------------------------------
    def _to_host(self, dev_ptr: int, dtype, count: int) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        _abi.check(self._lib, self._lib.gte_copy_to_host(self._h, C.c_void_p(dev_ptr),
                                                         out.ctypes.data, out.nbytes))
        return out

    def state(self, name: str) -> np.ndarray:
        """Host copy of one per-env state array (struct gte_state_view member)."""
        dt = _NP[_abi.STATE_DTYPES[name]]
        if self._snap_epoch == self._epoch:  # numpy mode: already fetched with the results
            return np.ascontiguousarray(self._snap[name])
        if self._state_epoch != self._epoch:  # snapshot once per step/reset, not per field
            _abi.check(self._lib, self._lib.gte_get_state(self._h, C.byref(self._state)))
            self._state_epoch = self._epoch
        return self._to_host(getattr(self._state, name), dt, self.num_envs)

    def state_tensor(self, name: str):
        """One per-env state array as a torch tensor ON THE DEVICE, without a copy: a view of the
        library's struct-of-arrays snapshot (`gte_get_state`, refreshed once per step/reset on
        first use).  For device-side reward shaping / extra observation features:
        `envs.state_tensor("portfolio_valuation")`, `"real_position"`, `"idx"`, ...  The view is
        overwritten by the snapshot taken after a later step."""
        torch = self._torch
        if torch is None:
            raise ValueError("state_tensor needs output='torch'")
        if self._state_epoch != self._epoch:
            _abi.check(self._lib, self._lib.gte_get_state(self._h, C.byref(self._state)))
            self._state_epoch = self._epoch
        if name not in self._state_tensors:
            dt = _abi.STATE_DTYPES[name]

            class _Raw:  # the CUDA array interface torch.as_tensor understands (also on ROCm)
                __cuda_array_interface__ = {
                    "shape": (self.num_envs,), "typestr": "<i4" if dt == "int32" else "<f8",
                    "data": (int(getattr(self._state, name)), False), "version": 2, "strides": None}
            self._state_tensors[name] = torch.as_tensor(_Raw(), device=self._t["obs"].device)
        return self._state_tensors[name]

    def read_output(self, name: str) -> np.ndarray:
        """Host copy of one output array of the last step/reset."""
        N = self.num_envs
        spec = {"obs": (np.float32, N * int(np.prod(self.obs_shape))),
                "final_obs": (np.float32, N * int(np.prod(self.obs_shape))),
                "reward": (np.float32, N), "reward64": (np.float64, N),
                "terminated": (np.uint8, N), "truncated": (np.uint8, N),
                "term_count": (np.int32, 2), "term_ids": (np.int32, N)}[name]
        a = self._to_host(getattr(self._out, name), *spec)
        return a.reshape((N,) + self.obs_shape) if name in ("obs", "final_obs") else a

    def terminal_ids(self) -> np.ndarray:
        """Ids of the envs whose episode ended in the last step (sorted)."""
        _abi.check(self._lib, self._lib.gte_get_outputs(self._h, C.byref(self._out)))
        n = int(self.read_output("term_count")[self._out.term_slot])
        return np.sort(self.read_output("term_ids")[:n])

    def final_observations(self):
        """(env ids, terminal observations) of the envs that ended in the last step — the
        observation `TradingEnv.step` returned for them before the same-step reset
        (needs ``autoreset="same_step", final_obs=True``).  Torch mode: rows gathered on
        the device."""
        if not self.cfg.final_obs:
            raise ValueError("constructed without final_obs=True")
        ids = self.terminal_ids()
        if self.output == "torch":
            idx = self._torch.from_numpy(ids.astype(np.int64)).to(self._t["final_obs"].device)
            return ids, self._t["final_obs"][idx]
        return ids, self.read_output("final_obs")[ids]


    # -- vectorised access to what History logs (LazyInfo, BatchedHistory, metrics) ------------
    def _dataset_host_column(self, name):
        """One column of every dataset's info table (`data_<col>`, environments.py:142,260) or
        the index (`date`, :256), concatenated over the datasets: a (dataset, row) pair is ONE
        fancy index away, whatever N."""
        col = self._host_columns.get(name)
        if col is None:
            parts = []
            for ds in self.datasets:
                if name == "date":
                    parts.append(np.asarray(ds.index) if ds.index is not None else np.arange(ds.T))
                    continue
                c = name[len("data_"):]
                if ds.info_array is not None and c in ds.info_columns:
                    parts.append(ds.info_array[:, ds.info_columns.index(c)])
                elif c == "close":
                    parts.append(ds.close)
                else:
                    raise ValueError(f"Feature {name} does not exist ... Check the available "
                                     f"features : {self.info_keys}")
            col = np.concatenate(parts) if len(parts) > 1 else np.asarray(parts[0])
            if col.dtype == object:  # a numeric column staged through an object matrix
                try:
                    col = col.astype(np.float64)
                except (TypeError, ValueError):
                    pass
            self._host_columns[name] = col
        return col

    def _dataset_column(self, name, ds, idx):
        """Values of info column `name` at (dataset, row) pairs; on the device for numeric
        columns in torch mode, else host arrays."""
        col = self._dataset_host_column(name)
        torch = self._torch
        if torch is not None and isinstance(ds, torch.Tensor):
            if col.dtype.kind in "fiub":
                dev = self._dev_columns.get(name)
                if dev is None:
                    dev = torch.as_tensor(col, device=ds.device)
                    self._dev_columns[name] = dev
                    self._dev_offsets = torch.as_tensor(self._ds_offsets, device=ds.device)
                return dev[self._dev_offsets[ds.long()] + idx.long()]
            ds, idx = ds.cpu().numpy(), idx.cpu().numpy()
        return col[self._ds_offsets[np.asarray(ds)] + np.asarray(idx)]

    def _info_value(self, key, state, reward=None):
        """One `info` / History column for every env from a struct-of-arrays state reader
        (`state(name) -> host array`): the current state, or the terminal one."""
        if key in ("idx", "step", "position_index", "real_position", "portfolio_valuation",
                   "dataset_index"):
            return state(key)
        if key == "position":
            return np.asarray(self.positions, dtype=np.float64)[state("position_index")]
        if key == "reward":
            return reward
        if key == "date" or key.startswith("data_"):
            return self._dataset_column(key, state("dataset_index"), state("idx"))
        if key.startswith("portfolio_distribution_"):
            # Portfolio.get_portfolio_distribution, portfolio.py:49-57
            f = key[len("portfolio_distribution_"):]
            if f in ("interest_asset", "interest_fiat"):
                return state(f)
            src = state("asset" if f.endswith("asset") else "fiat")
            return np.maximum(0.0, -src if f.startswith("borrowed") else src)
        raise KeyError(key)

    def final_state(self, name: str) -> np.ndarray:
        """Host copy of one per-env array of the TERMINAL states (`gte_get_final_state`): row e
        is env e as it was when its last episode ended (same-step mode with final_obs)."""
        if self._final_epoch != self._epoch:
            _abi.check(self._lib, self._lib.gte_get_final_state(self._h, C.byref(self._final_view)))
            self._final_epoch = self._epoch
        return self._to_host(getattr(self._final_view, name), _NP[_abi.STATE_DTYPES[name]],
                             self.num_envs)

    def _final_tensor(self, name):
        """torch view [N] of one array of the terminal records (`gte_get_final_state`, refreshed
        once per step)."""
        torch = self._torch
        if self._final_epoch != self._epoch:
            _abi.check(self._lib, self._lib.gte_get_final_state(self._h, C.byref(self._final_view)))
            self._final_epoch = self._epoch
        t = self._final_tensors.get(name)
        if t is None:
            dt = _abi.STATE_DTYPES[name]

            class _Raw:
                __cuda_array_interface__ = {
                    "shape": (self.num_envs,), "typestr": "<i4" if dt == "int32" else "<f8",
                    "data": (int(getattr(self._final_view, name)), False), "version": 2, "strides": None}
            t = torch.as_tensor(_Raw(), device=self._t["obs"].device)
            self._final_tensors[name] = t
        return t

    def _overlay(self, v, name, only=None):
        """`v` (a History column at the newest row, [N]) with the entries of the envs that ended
        in the last step replaced by their TERMINAL row's value (same-step mode, BatchedHistory
        terminal view); `only`: restrict to these envs."""
        torch = self._torch
        ended = self._t["terminated"] | self._t["truncated"]
        if only is not None:
            ended = ended & only
        f = self._final_tensor
        if name in ("idx", "step", "position_index", "dataset_index", "portfolio_valuation",
                    "real_position"):
            t = f(name)
        elif name == "position":
            t = self._positions_table()[f("position_index").long()]
        elif name == "reward":
            t = self._t["reward64"]  # the terminal step's reward (the log row under it is a reset row: 0)
        elif name == "date" or name.startswith("data_"):
            t = self._dataset_column(name, f("dataset_index"), f("idx"))
            if not isinstance(t, torch.Tensor):  # host column (dates, objects)
                m = ended.cpu().numpy()
                out = np.array(v, copy=True)
                out[m] = np.asarray(t)[m]
                return out
        elif name.startswith("portfolio_distribution_"):
            k = name[len("portfolio_distribution_"):]
            if k in ("interest_asset", "interest_fiat"):
                t = f(k)
            else:
                src = f("asset" if k.endswith("asset") else "fiat")
                t = (-src if k.startswith("borrowed") else src).clamp_min(0.0)
        else:
            return v
        return torch.where(ended, t.to(v.dtype), v)

    def _final_entries(self) -> dict:
        """Gymnasium's `final_observation` / `final_info` (+ masks) for the envs that ended in
        the last step: the observation and the History row `TradingEnv.step` returned for them
        (environments.py:272) before the same-step reset."""
        N = self.num_envs
        ids, last = self.final_observations()
        mask = np.zeros(N, dtype=bool)
        mask[ids] = True
        f_obs = np.full(N, None, dtype=object)
        f_info = np.full(N, None, dtype=object)
        if len(ids):
            reward = self.read_output("reward64")  # the terminal step's reward (same-step mode)
            cols = {k: np.asarray(self._info_value(k, self.final_state, reward))[ids]
                    for k in self.info_keys}
            for j, e in enumerate(ids):
                f_obs[e] = last[j]
                f_info[e] = {k: v[j] for k, v in cols.items()}
        return {"final_observation": f_obs, "_final_observation": mask,
                "final_info": f_info, "_final_info": mask.copy()}

    # -- the device trajectory log ---------------------------------------------------------------
    def _log_view(self):
        _abi.check(self._lib, self._lib.gte_get_log(self._h, C.byref(self._logv)))
        return self._logv

    def _log_tensor(self, name):
        """torch view [L, N] of one log array (no copy)."""
        t = self._log_tensors.get(name)
        if t is None:
            v, dt = self._log_view(), np.dtype(_abi.LOG_DTYPES[name])

            # the log is one array of records [L, N]: a column is a strided view of it
            class _Raw:
                __cuda_array_interface__ = {
                    "shape": (int(v.L), int(v.N)), "typestr": dt.str, "version": 2,
                    "strides": (int(v.row_stride), int(v.env_stride)),
                    "data": (int(getattr(v, name)), False)}
            t = self._torch.as_tensor(_Raw(), device=self._t["obs"].device)
            self._log_tensors[name] = t
        return t

    def _log_rows(self, name, phys, order):
        """Log column `name`: physical row `phys` ([N]), per-env rows `phys[N]` ([N]), or — phys
        None — the rows `order`, oldest first ([R, N])."""
        N = self.num_envs
        if self._torch is not None:
            t = self._log_tensor(name)
            if phys is None:
                return t[self._torch.as_tensor(order, device=t.device)]
            if np.ndim(phys) == 0:
                return t[int(phys)]
            return t[phys.long(), self._torch.arange(N, device=t.device)]
        v, dt = self._log_view(), np.dtype(_abi.LOG_DTYPES[name])
        base, es, rs = int(getattr(v, name)), int(v.env_stride), int(v.row_stride)

        def rows_to_host(first, count):  # the records of `count` rows in one copy; the column = a strided view
            nbytes = (count - 1) * rs + (N - 1) * es + dt.itemsize
            raw = self._to_host(base + first * rs, np.dtype(np.uint8), nbytes)
            return np.ndarray(shape=(count, N), dtype=dt, buffer=raw, strides=(rs, es)).copy()
        if phys is not None and np.ndim(phys) == 0:
            return rows_to_host(int(phys), 1)[0]
        whole = rows_to_host(0, int(v.L))
        return whole[order] if phys is None else whole[np.asarray(phys), np.arange(N)]

    def _log_row(self, name, phys, raw=False):
        return self._log_rows(name, phys, None)

    def _wrap(self, x):
        """What user callables receive: DeviceArray around device tensors, host arrays as is."""
        if self._torch is not None and isinstance(x, self._torch.Tensor):
            from .device_array import DeviceArray
            return DeviceArray(x)
        return x

    def _positions_table(self):
        if self._torch is None:
            return np.asarray(self.positions, dtype=np.float64)
        if self._pos_table is None:
            self._pos_table = self._torch.tensor(self.positions, dtype=self._torch.float64,
                                                 device=self._t["obs"].device)
        return self._pos_table

    def _take(self, table, index):
        return table[index.long()] if self._torch is not None else table[index]

    def _relu(self, x):
        return x.clamp_min(0.0) if self._torch is not None else np.maximum(0.0, x)

    def _stack(self, cols):
        return self._torch.stack(cols, dim=-1) if self._torch is not None else np.stack(cols, axis=-1)

    def _arange_rows(self, n):
        if self._torch is not None:
            return self._torch.arange(n, device=self._t["obs"].device)
        return np.arange(n)

    def _set_log_reward(self, value):
        """`historical_info["reward", -1] = reward` (environments.py:267) for the batch."""
        if self._torch is not None:
            from .device_array import to_tensor
            r = to_tensor(value, self._t["obs"].device, self._torch.float64).contiguous()
            self._keep_reward = r
            _abi.check(self._lib, self._lib.gte_set_log_reward(self._h, C.c_void_p(r.data_ptr())))
        else:
            raise NotImplementedError("assigning log rewards needs output='torch'")

    def batched_history(self, terminal: bool = False):
        """The `BatchedHistory` of the batch right now (needs ``log_steps``): what custom reward /
        dynamic-feature / metric callables receive.  terminal=True: the same-step view in which
        envs that just ended show their terminal row as the newest one."""
        from .batched_history import BatchedHistory
        return BatchedHistory(self, terminal=terminal)

    def _apply_callables(self, after_reset: bool):
        """The user's own `reward_function` / `dynamic_feature_functions` (any callable that is
        not this package's default object), evaluated ONCE for the whole batch over a
        BatchedHistory, where the reference calls them per env: reward after `History.add`
        (environments.py:265-267: skipped — reward 0 — when done; rows a reset wrote have reward
        0, :196), then the dynamic features inside `_get_obs` (:153-154), which therefore see the
        reward."""
        if self._reward_callable is None and not self._dyn_callables:
            return
        torch = self._torch
        if torch is None:
            raise NotImplementedError("custom callables in the batch need output='torch'")
        from .device_array import to_tensor
        dev = self._t["obs"].device
        N = self.num_envs
        same_step = self.cfg.autoreset == _abi.AUTORESET_SAME_STEP and not after_reset
        # same-step mode: an env that ended shows the callable its TERMINAL row (the log row of
        # this step already holds the state after the in-launch reset, which is what the NEXT
        # step's h[..., -2] must be)
        h = self.batched_history(terminal=same_step)
        ended = (self._t["terminated"] | self._t["truncated"]) if same_step else None
        newest = (h._rows - 1) % h._L
        if self._reward_callable is not None and not after_reset:
            r = to_tensor(self._reward_callable(h), dev, torch.float64)
            if r.shape != (N,):
                raise ValueError(f"reward_function must return one value per env, got shape {tuple(r.shape)}")
            # one kernel applies the reference's rules (0 where terminated, :265, and on reset rows,
            # :196 — except, in same-step mode, under an ended env's terminal row) and writes the
            # f64 / f32 returns and the newest log row (:267)
            r = r.contiguous()
            self._keep_reward = r
            _abi.check(self._lib, self._lib.gte_apply_reward(self._h, C.c_void_p(r.data_ptr()),
                                                             1 if same_step else 0))
        if self._dyn_callables:
            cols = (C.c_void_p * self.cfg.n_dyn)()
            is_f64 = (C.c_int32 * self.cfg.n_dyn)()
            keep = []
            fresh = self.batched_history() if (same_step and bool(ended.any())) else None
            for i, fn in self._dyn_callables:
                v = to_tensor(fn(h), dev, None)
                if v.dtype not in (torch.float32, torch.float64):
                    v = v.to(torch.float64)
                if v.shape != (N,):
                    v = v.expand(N) if v.dim() == 0 else v.reshape(N)
                if fresh is not None:
                    # the terminal observation shows the feature of the terminal row, the returned
                    # observation the one of the reset row (reset() evaluates it on a 1-row History)
                    col = self.datasets[0].n_static + i
                    fo = self._t["final_obs"]
                    if fo.dim() == 3:
                        fo[ended, -1, col] = v[ended].to(fo.dtype)
                    else:  # windows=None: the observation is one row
                        fo[ended, col] = v[ended].to(fo.dtype)
                    v = torch.where(ended, to_tensor(fn(fresh), dev, v.dtype), v)
                v = v.contiguous()
                keep.append(v)
                cols[i] = v.data_ptr()
                is_f64[i] = 1 if v.dtype == torch.float64 else 0
            self._keep_dyn = keep  # alive until the (stream-ordered) kernel has read them
            _abi.check(self._lib, self._lib.gte_set_dynamic_columns(self._h, cols, is_f64))

    # -- trajectory log ------------------------------------------------------------------
    def add_metric(self, name, function):
        """`TradingEnv.add_metric` (environments.py:274-278): `function(history)` is evaluated
        for every finished env by :meth:`episode_metrics` (needs ``log_steps``)."""
        if not self.cfg.log_steps:
            raise ValueError("add_metric needs log_steps > 0 (the History comes from the device log)")
        self.log_metrics.append({"name": name, "function": function})

    def read_log_envs(self, env_ids, finished: bool = False, max_rows=None) -> dict:
        """The logged episode of each listed env in ONE device->host transfer
        (`gte_read_log_envs`): dict of arrays [n_ids, max_rows] — the log columns of
        `_abi.LOG_DTYPES` — plus "n_rows" [n_ids]; rows 0 .. n_rows[j]-1 of env j are valid,
        oldest first.  Views of the library's pinned staging buffer: valid until the next call.
        finished=True: the episodes that just ENDED (same-step auto-reset with ``final_obs``)."""
        L = int(self.cfg.log_steps)
        if not L:
            raise ValueError("constructed with log_steps=0")
        ids = np.ascontiguousarray(np.asarray(env_ids, dtype=np.int32).reshape(-1))
        b = _abi.GteLogBatch()
        _abi.check(self._lib, self._lib.gte_read_log_envs(
            self._h, ids.ctypes.data, len(ids), int(max_rows or 0), 1 if finished else 0, C.byref(b)))
        n, R = int(b.n_ids), int(b.max_rows)
        out = {"n_rows": np.zeros(0, np.int32)}
        if n:
            out["n_rows"] = np.frombuffer((C.c_int32 * n).from_address(b.n_rows), dtype=np.int32)
        for name, dt in _abi.LOG_DTYPES.items():
            dt = np.dtype(dt)
            if not n:
                out[name] = np.zeros((0, R), dt)
                continue
            raw = (C.c_char * (n * R * dt.itemsize)).from_address(getattr(b, name))
            out[name] = np.frombuffer(raw, dtype=dt).reshape(n, R)
        return out

    def histories(self, env_ids, finished: bool = False) -> list:
        """One `History` per listed env: its current (or, finished=True, just finished) episode
        with the reference's columns (environments.py:186-197, 253-264), rebuilt from the device
        trajectory log.  ONE transfer for all of them; the columns are built once over the
        concatenated rows and cut per env."""
        from .history import ColumnBlock, History
        ids = np.asarray(env_ids, dtype=np.int64).reshape(-1)
        if finished and not self.cfg.final_obs:
            raise ValueError("history(finished=True) needs autoreset='same_step', final_obs=True")
        b = self.read_log_envs(ids, finished=finished)
        n = b["n_rows"].astype(np.int64)
        if finished and len(ids):
            last = b["flags"][np.arange(len(ids)), np.maximum(n - 1, 0)]
            bad = np.nonzero((n == 0) | ((last & 3) == 0))[0]
            if len(bad):
                raise ValueError(f"env {int(ids[bad[0]])} did not end in the last step")
        R = b["idx"].shape[1]
        valid = np.arange(R)[None, :] < n[:, None]
        raw = {}

        def flat(name):  # the valid rows of every env, env after env (copied out of the staging buffer)
            v = raw.get(name)
            if v is None:
                v = raw[name] = b[name][valid]
            return v
        for name in ("dataset_index", "idx"):  # the lookups below outlive the staging buffer
            flat(name)
        pos_table = np.empty(len(self.positions), dtype=object)
        pos_table[:] = self.positions
        dist = lambda src, sign: (lambda: np.maximum(0, sign * flat(src)))
        # Every column is built on first use, over all the episodes at once, and cut per env: a
        # metric that reads `h["position"]` never pays for dates or portfolio distributions.
        builders = {"idx": lambda: flat("idx"), "step": lambda: flat("step"),
                    "date": lambda: self._dataset_column("date", flat("dataset_index"), flat("idx")),
                    "position_index": lambda: flat("position_index"),
                    "position": lambda: pos_table[flat("position_index")],
                    "real_position": lambda: flat("real_position")}
        for c in self.datasets[0].info_columns or ["close"]:
            builders[f"data_{c}"] = (lambda c=c: self._dataset_column(
                f"data_{c}", flat("dataset_index"), flat("idx")))
        builders["portfolio_valuation"] = lambda: flat("portfolio_valuation")
        # Portfolio.get_portfolio_distribution, portfolio.py:49-57
        builders["portfolio_distribution_asset"] = dist("asset", 1.0)
        builders["portfolio_distribution_fiat"] = dist("fiat", 1.0)
        builders["portfolio_distribution_borrowed_asset"] = dist("asset", -1.0)
        builders["portfolio_distribution_borrowed_fiat"] = dist("fiat", -1.0)
        builders["portfolio_distribution_interest_asset"] = lambda: flat("interest_asset")
        builders["portfolio_distribution_interest_fiat"] = lambda: flat("interest_fiat")
        builders["reward"] = lambda: flat("reward")
        # the staging buffer is rewritten by the next read: take the raw columns out of it now (one
        # boolean-mask copy each, 12 small arrays), the derived columns stay lazy
        for name in _abi.LOG_DTYPES:
            flat(name)
        block = ColumnBlock(builders)
        ends = np.cumsum(n).tolist()
        out, lo = [], 0
        for hi in ends:
            out.append(History.from_block(block, lo, hi))
            lo = hi
        return out

    def history(self, env_id: int, finished: bool = False):
        """The current (or just finished) episode of one env as a `History` with the
        reference's columns (environments.py:186-197, 253-264: idx, step, date, position_index,
        position, real_position, data_*, portfolio_valuation, portfolio_distribution_*, reward),
        rebuilt from the device trajectory log.  Episodes longer than ``log_steps`` are
        truncated at the front.

        finished=True (same-step auto-reset with ``final_obs``, right after the step in which the
        env ended): the episode that just FINISHED — its rows from the log plus the terminal row
        from the env's terminal record (that step's log row already describes the new episode).
        For many envs use :meth:`histories` (one transfer for all of them)."""
        return self.histories([int(env_id)], finished=finished)[0]

    def save_for_render(self, env_id: int, dir="render_logs"):
        """`TradingEnv.save_for_render` (environments.py:296-307) for one env of the batch: the
        dataset's frame joined with the episode History rebuilt from the device trajectory log
        (needs ``log_steps``), pickled as `<name>_<timestamp>.pkl` for the reference's renderer.
        -> path of the file."""
        import datetime
        import os
        import pandas as pd
        h = self.history(env_id)
        ds = self.datasets[int(self.state("dataset_index")[env_id])]
        missing = [c for c in ("open", "high", "low", "close") if c not in (ds.info_columns or [])]
        assert not missing, (
            "Your DataFrame needs to contain columns : open, high, low, close to render !")
        frame = pd.DataFrame(ds.info_array, columns=ds.info_columns, index=ds.index)
        logged = pd.DataFrame({c: h[c] for c in h.columns}).set_index("date").sort_index()
        os.makedirs(dir, exist_ok=True)
        stamp = datetime.datetime.now().strftime("%Y-%m-%d_%H-%M-%S")
        path = os.path.join(dir, f"{ds.name}_env{int(env_id)}_{stamp}.pkl")
        frame.join(logged, how="inner").to_pickle(path)
        return path

    def episode_metrics(self, env_ids=None) -> dict:
        """Episode-end metrics of `calculate_metrics` (environments.py:279-286) for the envs
        whose episode just ended (default: `terminal_ids()`), from the device state:
        "Market Return" = close[idx] / close[start_idx] - 1 and "Portfolio Return" =
        portfolio_valuation / initial value - 1, as float arrays plus the reference's
        formatted strings.  Call it right after the terminal step (with next-step auto-reset
        the state still belongs to the finished episode until the following step; with
        same-step auto-reset and ``final_obs`` the terminal records are used)."""
        ids = self.terminal_ids() if env_ids is None else np.asarray(env_ids, dtype=np.int64)
        state = self.final_state if (self.cfg.final_obs and env_ids is None) else self.state
        use_final = bool(self.cfg.final_obs and env_ids is None)
        ds, idx, start = (state(k)[ids] for k in ("dataset_index", "idx", "start_idx"))
        pv = state("portfolio_valuation")[ids]
        close_now = np.asarray(self._dataset_column("data_close", ds, idx), dtype=np.float64)
        close_0 = np.asarray(self._dataset_column("data_close", ds, start), dtype=np.float64)
        market = close_now / close_0 - 1 if len(ids) else np.zeros(0)
        portfolio = pv / self.cfg.portfolio_initial_value - 1
        out = {"env_ids": ids, "market_return": market, "portfolio_return": portfolio,
               "episode_length": state("step")[ids] + 1,
               "Market Return": [f"{100 * m:5.2f}%" for m in market.tolist()],
               "Portfolio Return": [f"{100 * r:5.2f}%" for r in portfolio.tolist()]}
        if self.log_metrics:  # custom metrics over each finished env's History (:285-286)
            hists = self.histories(ids, finished=use_final)  # ONE transfer for all of them
            for metric in self.log_metrics:
                out[metric["name"]] = [metric["function"](h) for h in hists]
        return out

    def _rotate_returns(self):
        """Point the next step at the next packed return buffer (gte_bind_returns)."""
        self._ret_slot = (self._ret_slot + 1) % self.return_slots
        buf, views, ptrs = self._slot_views[self._ret_slot]
        _abi.check(self._lib, self._lib.gte_bind_returns(self._h, *ptrs))
        self.packed_returns = buf
        self._out.reward, self._out.terminated, self._out.truncated = (p.value for p in ptrs)
        t = self._t
        t["reward"], t["terminated"], t["truncated"] = views

    @property
    def return_slot(self) -> int:
        """Row of the [return_slots, 6N] buffer the NEXT step writes its returns into."""
        return (self._ret_slot + 1) % self.return_slots

    def return_block(self, first: int, count: int):
        """uint8 [count, 6N]: the packed returns in rows first..first+count-1 (contiguous)."""
        if not (0 <= first and first + count <= self.return_slots):
            raise IndexError("return block out of range")
        return self._packed_all[first:first + count]

    def _results(self):
        if self.output == "torch":
            t = self._t
            return t["obs"], t["reward"], t["terminated"], t["truncated"]
        # host arrays: state, returns and observations of the whole batch in ONE transfer
        self._snap, self._snap_obs = self.read_envs(view=not self.copy)
        self._snap_epoch = self._epoch
        snap = self._snap
        return (self._snap_obs, np.ascontiguousarray(snap["reward"]),
                snap["terminated"].astype(bool), snap["truncated"].astype(bool))

    # -- the Env API -------------------------------------------------------------------
    def reset(self, seed=None, options=None, *, mask=None, inject_idx=None,
              inject_position_index=None, inject_dataset=None):
        """`TradingEnv.reset` (environments.py:163-199) for all envs, or those in `mask`.

        `seed` is accepted for API compatibility; like in the reference (whose reset
        draws from the global NumPy RNG) it does not reseed the episode draws — the
        device Philox stream is keyed by the constructor's `seed`."""
        def arr(a, dt):
            if a is None:
                return None, None
            a = np.ascontiguousarray(np.asarray(a, dtype=dt))
            if a.shape != (self.num_envs,):
                raise ValueError(f"expected shape ({self.num_envs},), got {a.shape}")
            return a, a.ctypes.data
        m, mp = arr(mask, np.uint8)
        a, ap = arr(inject_idx, np.int32)
        b, bp = arr(inject_position_index, np.int32)
        c, cp = arr(inject_dataset, np.int32)
        _abi.check(self._lib, self._lib.gte_reset(self._h, mp, ap, bp, cp))
        self._was_reset = True
        self._epoch += 1
        self._apply_callables(after_reset=True)
        return self._results()[0], LazyInfo(self)

    def set_autoreset_injection(self, idx=None, position_index=None, dataset=None):
        """Queue the draws of later auto-resets: i32 [N, n_episodes] arrays (parity runs)."""
        arrs = [np.asarray(x) for x in (idx, position_index, dataset) if x is not None]
        n = 0 if not arrs else arrs[0].reshape(self.num_envs, -1).shape[1]
        def arr(a):
            if a is None:
                return None, None
            a = np.ascontiguousarray(np.asarray(a, dtype=np.int32).reshape(self.num_envs, n))
            return a, a.ctypes.data
        a, ap = arr(idx)
        b, bp = arr(position_index)
        c, cp = arr(dataset)
        _abi.check(self._lib, self._lib.gte_set_autoreset_injection(self._h, n, ap, bp, cp))

    def add_limit_order(self, position_index, limit, persistent=False):
        """`TradingEnv.add_limit_order` (environments.py:227-231) for a batch.

        position_index: i32 [N] index into `positions` of each env's order target, -1 for
        "no order for this env"; limit: f64 [N]; persistent: bool or bool [N]."""
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(position_index, np.int32), (self.num_envs,)))
        b = np.ascontiguousarray(np.broadcast_to(np.asarray(limit, np.float64), (self.num_envs,)))
        c = np.ascontiguousarray(np.broadcast_to(np.asarray(persistent, np.uint8), (self.num_envs,)))
        _abi.check(self._lib, self._lib.gte_add_limit_orders(self._h, a.ctypes.data, b.ctypes.data,
                                                             c.ctypes.data))
        self._epoch += 1

    def step(self, actions):
        """`TradingEnv.step` (environments.py:233-272) for every env in one launch.

        actions: position indices, shape (N,); -1 (or None entries) = hold (:234).
        A torch int32 CUDA tensor is used in place; anything else goes through a
        host->device copy."""
        self._launch_step(actions)
        self._apply_callables(after_reset=False)
        if self.verbose > 0 and self._user_log:
            import time
            now = time.monotonic()
            if now - self._last_report >= self.verbose_interval:
                self._last_report = now
                self.log()
        obs, reward, term, trunc = self._results()
        return obs, reward, term, trunc, LazyInfo(self)

    def log(self):
        """`TradingEnv.log` after `calculate_metrics` (environments.py:269-271, :279-294) for the
        envs whose episode ended in the last step: one line each, the reference's text —
        "Market Return : ..   |   Portfolio Return : ..   |   " plus every `add_metric` entry.  A
        batch can end hundreds of episodes per step, so at most `verbose_max_lines` lines are
        printed, then a count.  Costs a device synchronisation, which is why `step()` only does it
        when the user asked for a trajectory log (``log_steps`` > 0: an env set up for metrics, not
        the bare hot path, and not the 2-row log a Python callable brings with it), `verbose` > 0,
        and at most once per `verbose_interval` seconds (episodes that end between two reports
        are not listed; `episode_metrics()` after any step gives them all).  In same-step mode it needs ``final_obs`` (the terminal
        records); without them the finished episodes' final state is gone and nothing is printed."""
        if self.verbose <= 0:
            return
        if self.cfg.autoreset == _abi.AUTORESET_SAME_STEP and not self.cfg.final_obs:
            return
        m = self.episode_metrics()
        n = len(m["env_ids"])
        names = ["Market Return", "Portfolio Return"] + [x["name"] for x in self.log_metrics]
        for j in range(min(n, self.verbose_max_lines)):
            print("".join(f"{k} : {m[k][j]}   |   " for k in names))
        if n > self.verbose_max_lines:
            print(f"... and {n - self.verbose_max_lines} more episodes ended in this step")

    def _launch_step(self, actions):
        """The launch half of step(): nothing is copied back."""
        torch = self._torch
        on_device = torch is not None and isinstance(actions, torch.Tensor) and actions.is_cuda
        if on_device:
            if actions.dtype != torch.int32 or not actions.is_contiguous():
                actions = actions.to(torch.int32).contiguous()
            if actions.shape != (self.num_envs,):
                raise ValueError(f"expected {self.num_envs} actions")
        else:
            if torch is not None and isinstance(actions, torch.Tensor):
                actions = actions.cpu().numpy()
            a = np.array([-1 if x is None else x for x in actions], dtype=np.int32) \
                if isinstance(actions, (list, tuple)) else np.asarray(actions, dtype=np.int32)
            a = np.ascontiguousarray(a)
            if a.shape != (self.num_envs,):
                raise ValueError(f"expected {self.num_envs} actions")
            if a.size and (a.max() >= len(self.positions) or a.min() < -1):
                raise IndexError("list index out of range")  # positions[position_index] (:234)
        # whichever way the actions arrive, the step writes the NEXT row of the return buffers
        if self.return_slots > 1:
            self._rotate_returns()
        if on_device:
            self._keep = actions  # keep alive until the launch has consumed it
            _abi.check(self._lib, self._lib.gte_step(self._h, C.c_void_p(actions.data_ptr()), 1))
        else:
            _abi.check(self._lib, self._lib.gte_step(self._h, a.ctypes.data, 0))
        self._epoch += 1

    def capture_steps(self, body, n_steps: int):
        """Record `body(i)` for i in range(n_steps) — each call taking ONE `step()` with a CUDA
        int32 action tensor, plus any torch code around it (the policy) — into a HIP graph
        (`torch.cuda.graph`); `.replay()` on the returned `StepGraph` runs them again with one
        host call.  For launch-bound batches (config 2: 4 096 envs).  n_steps must be even."""
        from .step_graph import StepGraph
        return StepGraph(self, body, n_steps)

    def read_envs(self, first: int = 0, count=None, with_obs: bool = True, view: bool = False):
        """(snapshots, obs): a structured array [count] with the fields of struct
        gte_env_snapshot (state, reward f64, terminated, truncated) and the observations
        [count, *obs_shape] (or None) of envs first..first+count-1, fetched with one
        device->host transfer (`gte_read_envs`).  view=True: arrays over the library's pinned
        staging buffer, valid until the next read (`gte_read_envs_view`)."""
        count = self.num_envs - first if count is None else int(count)
        if view:
            ps, po = C.c_void_p(), C.c_void_p()
            _abi.check(self._lib, self._lib.gte_read_envs_view(
                self._h, int(first), count, 1 if with_obs else 0, C.byref(ps), C.byref(po)))
            snap = np.frombuffer((C.c_char * (96 * count)).from_address(ps.value),
                                 dtype=_abi.SNAPSHOT_DTYPE)
            obs = None
            if with_obs:
                n = count * int(np.prod(self.obs_shape))
                obs = np.frombuffer((C.c_float * n).from_address(po.value), dtype=np.float32
                                    ).reshape((count,) + self.obs_shape)
            return snap, obs
        snap = np.empty(count, dtype=_abi.SNAPSHOT_DTYPE)
        obs = np.empty((count,) + self.obs_shape, np.float32) if with_obs else None
        _abi.check(self._lib, self._lib.gte_read_envs(
            self._h, int(first), count, snap.ctypes.data, obs.ctypes.data if with_obs else None))
        return snap, obs

    def read_env(self, env_index: int = 0, with_obs: bool = True):
        """(GteEnvSnapshot, obs ndarray | None) of ONE env after the last step/reset: state,
        returns and observation in a single device->host transfer (`gte_read_env`)."""
        snap = _abi.GteEnvSnapshot()
        obs = np.empty(self.obs_shape, np.float32) if with_obs else None
        _abi.check(self._lib, self._lib.gte_read_env(
            self._h, int(env_index), C.byref(snap), obs.ctypes.data if with_obs else None))
        return snap, obs

    def rollout(self, actions, *, keep_obs=False, valuation=False, reward64=False, out=None):
        """K consecutive step() calls for action sequences known in advance, in ONE launch
        (`gte_rollout`: backtests of precomputed strategies, random-policy collection).

        actions: int32 [K, N] (-1 = hold), a torch CUDA tensor or anything array-like.
        Returns a dict of device tensors: reward f32 [K, N], terminated / truncated bool
        [K, N], obs — [K, N, *obs_shape] with keep_obs=True, else the observation after the
        last step [N, *obs_shape] —, and on request valuation f64 [K, N] (portfolio value
        after each step) and reward64.  State and results equal K single steps exactly.
        out: the dict a previous call with the same K and options returned — its tensors are
        written again instead of allocating new ones (K x 168 MB of observations at the headline
        shape; where a buffer lands in memory also moves the store rate by a few percent)."""
        torch = self._torch
        if torch is None:
            raise ValueError("rollout needs output='torch'")
        if self._reward_callable is not None or self._dyn_callables:
            raise NotImplementedError("a fused rollout runs all steps on the device: custom Python "
                                      "reward / dynamic-feature callables need step()")
        dev = self._t["obs"].device
        if not (isinstance(actions, torch.Tensor) and actions.is_cuda):
            a = np.asarray([[-1 if x is None else x for x in row] for row in actions]
                           if isinstance(actions, (list, tuple)) else actions, dtype=np.int32)
            if a.size and (a.max() >= len(self.positions) or a.min() < -1):
                raise IndexError("list index out of range")
            actions = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        actions = actions.to(torch.int32).contiguous()
        if actions.dim() != 2 or actions.shape[1] != self.num_envs or actions.shape[0] < 1:
            raise ValueError(f"expected actions of shape (K, {self.num_envs})")
        K, N = int(actions.shape[0]), self.num_envs
        want = {"reward": ((K, N), torch.float32), "terminated": ((K, N), torch.bool),
                "truncated": ((K, N), torch.bool)}
        if keep_obs:
            want["obs"] = ((K, N) + self.obs_shape, torch.float32)
        if valuation:
            want["valuation"] = ((K, N), torch.float64)
        if reward64:
            want["reward64"] = ((K, N), torch.float64)
        if out is not None:
            out = {k: out[k] for k in want}  # KeyError: not the result of a matching call
            for k, (shape, dt) in want.items():
                t = out[k]
                if tuple(t.shape) != tuple(shape) or t.dtype != dt or t.device != dev or not t.is_contiguous():
                    raise ValueError(f"out[{k!r}] does not match this rollout")
        else:
            with torch.cuda.device(dev):
                out = {k: torch.empty(shape, dtype=dt, device=dev) for k, (shape, dt) in want.items()}
        b = _abi.GteRolloutBufs()
        for k, t in out.items():
            setattr(b, k, t.data_ptr())
        self._keep = (actions, out)  # alive until the launch has consumed them
        _abi.check(self._lib, self._lib.gte_rollout(self._h, C.c_void_p(actions.data_ptr()), K,
                                                    C.byref(b)))
        self._epoch += 1
        if not keep_obs:
            out["obs"] = self._t["obs"]
        return out

    # -- misc ---------------------------------------------------------------------------
    def synchronize(self):
        _abi.check(self._lib, self._lib.gte_synchronize(self._h))

    def launch_info(self) -> dict:
        v = [C.c_int32() for _ in range(4)]
        _abi.check(self._lib, self._lib.gte_get_launch_info(self._h, *[C.byref(x) for x in v]))
        d = dict(zip(("envs_per_wave", "threads_per_block", "n_blocks", "vector_bytes"),
                     (x.value for x in v)))
        flags, d["vector_bytes"] = divmod(d["vector_bytes"], 1000)
        d["phase_a"] = "cooperative" if flags & 1 else "per-wave"
        d["dyn_columns"] = ("global", "lds-raw-rings", "lds-resolved")[(flags >> 1) & 3]
        d["structure"] = "phase A, LDS barrier, gather"
        d["obs_stores"] = ("plain", "non-temporal", "sc1", "?")[(flags >> 4) & 3]
        d["resident_workgroups_per_cu"] = (flags >> 6) & 15
        return d

    def timer_start(self):
        _abi.check(self._lib, self._lib.gte_timer_start(self._h))

    def timer_mark(self):
        """Record the end of the timed span now, without waiting (read it with `timer_stop`)."""
        _abi.check(self._lib, self._lib.gte_timer_stop(self._h, None))

    def timer_stop(self) -> float:
        ms = C.c_float()
        _abi.check(self._lib, self._lib.gte_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def close(self, **kwargs):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.gte_destroy(self._h)
            self._h = C.c_void_p()
        self.closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
