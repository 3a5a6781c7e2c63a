"""Replay a golden trace (tests/golden/*.npz, made by make_golden.py from the
reference itself) through any batched implementation and compare per call.

`adapter` is an object with reset(mask, inj_idx, inj_pos, inj_ds),
set_autoreset_injection(idx, pos, ds), step(actions) and numpy accessors
obs(), reward64(), terminated(), truncated(), state() -> dict.
"""
from __future__ import annotations

import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

STATE_F64 = {"asset": "asset", "fiat": "fiat", "interest_asset": "interest_asset",
             "interest_fiat": "interest_fiat", "portfolio_valuation": "portfolio_valuation",
             "real_position": "real_position"}
STATE_I32 = {"idx": "idx", "step": "step", "pos_index": "position_index",
             "dataset": "dataset_index"}


def golden_names():
    """Names of the trace fixtures a BATCHED implementation can replay (portfolio_random.npz is
    a known-answer table and set_df.npz a staging fixture, not traces; hostcb_* traces need
    Python callables per env and are replayed by the N=1 drop-in only)."""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                  if not os.path.basename(p).startswith(("portfolio_", "hostcb_", "set_df")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    g["cfg"] = json.loads(str(g.pop("cfg_json")))
    d = 0
    sets = []
    while f"feat_{d}" in g:
        ds = (g.pop(f"feat_{d}"), g.pop(f"close_{d}"))
        if f"high_{d}" in g:
            ds += (g.pop(f"high_{d}"), g.pop(f"low_{d}"))
        sets.append(ds)
        d += 1
    g["datasets"] = sets
    return g


def config_kwargs(g, tile: int = 1, **over):
    """make_config kwargs for a trace, its E envs repeated `tile` times."""
    cfg = dict(g["cfg"])
    K, E = g["op"].shape
    rf = cfg.get("reward_function", "basic_reward_function")
    cfg["reward_function"] = tuple(rf) if isinstance(rf, list) else rf
    kw = dict(n_envs=E * tile, n_static=g["datasets"][0][0].shape[1],
              n_datasets=len(g["datasets"]),
              autoreset="next_step" if (g["op"][1:] == 0).any() else None, **cfg)
    kw.update(over)
    return kw


def staged(g, n_dyn):
    """(feat [T, F_obs] with zero dynamic columns, close) per dataset."""
    out = []
    for ds in g["datasets"]:
        feat = ds[0]
        full = np.zeros((feat.shape[0], feat.shape[1] + n_dyn), np.float32)
        full[:, :feat.shape[1]] = feat
        out.append((full,) + tuple(ds[1:]))
    return out


def injection_queue(g, tile: int = 1):
    """Per env, the reference's draws at every reset after call 0, padded with -1."""
    op = g["op"]
    K, E = op.shape
    counts = (op[1:] == 0).sum(axis=0)
    n = int(counts.max()) if counts.size else 0
    q = {f: np.full((E, max(n, 1)), -1, np.int32) for f in ("idx", "pos_index", "dataset")}
    for e in range(E):
        ks = np.nonzero(op[1:, e] == 0)[0] + 1
        for j, k in enumerate(ks):
            for f in q:
                q[f][e, j] = g[f][k, e]
    return {f: np.tile(a, (tile, 1)) for f, a in q.items()}, n


def replay(adapter, g, tile: int = 1, rtol: float = 1e-12, obs_exact: bool = True,
           check_state: bool = True):
    """Drive `adapter` through trace g; assert parity at every call."""
    K, E = g["op"].shape
    t = lambda a: np.tile(a, tile)
    assert (g["op"][0] == 0).all()
    q, n = injection_queue(g, tile)
    if n:
        adapter.set_autoreset_injection(q["idx"], q["pos_index"], q["dataset"])
    adapter.reset(None, t(g["idx"][0]), t(g["pos_index"][0]), t(g["dataset"][0]))
    worst = 0.0
    for k in range(K):
        if k > 0:
            if "lo_pos" in g and (g["lo_pos"][k] >= 0).any():
                adapter.add_limit_orders(t(g["lo_pos"][k]), t(g["lo_limit"][k]),
                                         np.ones(E * tile, np.uint8))
            adapter.step(t(g["action"][k]))
        st = adapter.state()
        tag = f"call {k}"
        for gk, sk in STATE_I32.items():
            np.testing.assert_array_equal(st[sk], t(g[gk][k]), err_msg=f"{tag} {gk}")
        np.testing.assert_array_equal(adapter.terminated().astype(bool),
                                      t(g["done"][k]).astype(bool), err_msg=f"{tag} done")
        np.testing.assert_array_equal(adapter.truncated().astype(bool),
                                      t(g["truncated"][k]).astype(bool), err_msg=f"{tag} truncated")
        if check_state:
            for gk, sk in STATE_F64.items():
                ref = t(g[gk][k])
                got = st[sk]
                err = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)
                err = np.where(ref == got, 0.0, err)
                worst = max(worst, float(err.max()))
                np.testing.assert_allclose(got, ref, rtol=rtol, atol=1e-14 if rtol > 0 else 0,
                                           err_msg=f"{tag} {gk}")
        np.testing.assert_allclose(adapter.reward64(), t(g["reward"][k]), rtol=max(rtol, 1e-12),
                                   atol=1e-15, err_msg=f"{tag} reward")
        obs = adapter.obs()
        ref_obs = np.tile(g["obs"][k], (tile,) + (1,) * (g["obs"][k].ndim - 1))
        if obs_exact:
            np.testing.assert_array_equal(obs, ref_obs, err_msg=f"{tag} obs")
        else:
            np.testing.assert_allclose(obs, ref_obs, rtol=1e-6, atol=1e-7, err_msg=f"{tag} obs")
    return worst
