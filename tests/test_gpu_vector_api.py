"""The batch as a drop-in for the reference's VECTORISED usage
(examples/example_vectorized_environment.py:39-62, docs/source/vectorize_env.rst:17-33,
docs/source/customization.rst:9-20): a Python `reward_function(history)` /
`dynamic_feature_functions` given to the batch, the complete `info` dict-of-arrays
(`date`, every `data_*` column), `final_observation` / `final_info` in same-step mode, and
callables recognised by identity only.  Needs an MI355X."""
import numpy as np
import pandas as pd
import pytest

import replay

pytestmark = pytest.mark.gpu


def make_df(feat, close, seed=0):
    T = len(close)
    rng = np.random.default_rng(seed)
    df = pd.DataFrame({"open": close * (1 + rng.normal(0, 1e-3, T)), "high": close * 1.004,
                       "low": close * 0.996, "close": close, "volume": rng.uniform(1, 9, T)},
                      index=pd.date_range("2022-03-01", periods=T, freq="30min"))
    for j in range(feat.shape[1]):
        df[f"feature_{j}"] = feat[:, j]
    return df


def _walk(seed, T, Fs, sigma=1e-2, drift=0.0):
    rng = np.random.default_rng(seed)
    close = 100 * np.exp(np.cumsum(rng.normal(drift, sigma, T)))
    return rng.normal(0, 1, (T, Fs)).astype(np.float32), close


# the reference example's reward function, verbatim (example_vectorized_environment.py:39-40)
def reward_function(history):
    return np.log(history["portfolio_valuation", -1] / history["portfolio_valuation", -2])  # log (p_t / p_t-1 )


EXAMPLE_KW = dict(name="BTCUSD", windows=5, positions=[-1, -0.5, 0, 0.5, 1, 1.5, 2], initial_position=0,
                  trading_fees=0.01 / 100, borrow_interest_rate=0.0003 / 100,
                  portfolio_initial_value=1000)  # example_vectorized_environment.py:44-57


@pytest.mark.parametrize("N", [3, 1500])
def test_batch_runs_the_reference_vector_example(oracle_mod, N):
    """The example's constructor arguments, its own Python reward function included, go
    through BatchedTradingEnv unchanged; results equal the device built-in (the same formula)
    and the oracle, the info dict carries every History column."""
    import torch
    from gym_trading_env_amd.batched import BatchedTradingEnv
    feat, close = _walk(3, 900, 5, sigma=2e-2, drift=-5e-4)
    df = make_df(feat, close)
    kw = dict(EXAMPLE_KW, max_episode_duration=60, seed=5)
    env = BatchedTradingEnv(df, num_envs=N, reward_function=reward_function, **kw)
    twin = BatchedTradingEnv(df, num_envs=N, **kw)            # device log-return
    assert env.cfg.log_steps == 2 and twin.cfg.log_steps == 0
    full = np.zeros((900, 7), np.float32); full[:, :5] = feat
    ora = oracle_mod.OracleEnv(twin.cfg, [(full, close)])
    obs, info = env.reset(); twin.reset(); ora.reset()
    rng = np.random.default_rng(1)
    ended = 0
    for k in range(150):
        a = rng.integers(0, 7, N).astype(np.int32)
        obs, reward, term, trunc, info = env.step(torch.from_numpy(a).cuda())
        o2, r2, t2, u2, _ = twin.step(torch.from_numpy(a).cuda())
        ora.step(a)
        np.testing.assert_array_equal(obs.cpu().numpy(), ora.obs)
        np.testing.assert_array_equal(term.cpu().numpy(), ora.terminated.astype(bool))
        np.testing.assert_array_equal(trunc.cpu().numpy(), ora.truncated.astype(bool))
        # torch's f64 log on the device vs the kernel's: the same formula, <= 1 ulp apart
        np.testing.assert_allclose(env.read_output("reward64"), twin.read_output("reward64"),
                                   rtol=4e-16, atol=0)
        np.testing.assert_allclose(reward.cpu().numpy(), ora.reward, rtol=1e-6, atol=1e-12)
        ended += int((ora.terminated | ora.truncated).sum())
    assert ended > N  # episodes ended and restarted: reset rows got reward 0 like the reference
    # the info dict: every column History logs (environments.py:253-264), arrays of length N
    idx = info["idx"]
    assert set(info.keys()) >= {"idx", "step", "date", "position_index", "position", "real_position",
                                "data_open", "data_high", "data_low", "data_close", "data_volume",
                                "portfolio_valuation", "portfolio_distribution_asset",
                                "portfolio_distribution_interest_fiat", "reward"}
    np.testing.assert_array_equal(info["date"], df.index.values[idx])
    for c in ("open", "high", "low", "close", "volume"):
        np.testing.assert_array_equal(info[f"data_{c}"], df[c].to_numpy()[idx])
    assert info["_data_volume"].all() and info["_date"].shape == (N,)
    np.testing.assert_array_equal(info["position"], np.asarray(kw["positions"])[info["position_index"]])
    np.testing.assert_array_equal(info["reward"], env.read_output("reward64"))
    env.close(); twin.close()


# batch forms of tests/custom_callables.py (the reference ran those per env to make the
# fixture): one value per env; `len(history) < 2` of a single env's History becomes "the newest
# row is a reset row"
def dyn_valuation_ratio(h):
    return h["portfolio_valuation", -1] / 1000.0


def dyn_exposure_change(h):
    if len(h) < 2:
        return np.zeros(h["step", -1].shape[0])
    return np.where(h["step", -1] == 0, 0.0, h["real_position", -1] - h["real_position", -2])


def reward_simple_return_minus_turnover(h):
    ret = h["portfolio_valuation", -1] / h["portfolio_valuation", -2] - 1
    turnover = abs(h["position", -1] - h["position", -2])
    return ret - 1e-4 * turnover


def test_batch_custom_callables_replay_the_reference_fixture():
    """tests/golden/hostcb_custom_callables.npz: custom dynamic features + a custom reward run
    by the REFERENCE over its History (make_golden.py custom_callables_trace).  The batch
    evaluates the same formulas once per step for all envs, on the device, over a
    BatchedHistory; observations (dynamic columns included, in-place persistence included) and
    rewards equal the reference's."""
    import torch
    from gym_trading_env_amd.batched import BatchedTradingEnv
    g = replay.load("hostcb_custom_callables")
    feat, close = g["datasets"][0]
    cfg = g["cfg"]
    K, E = g["op"].shape
    tile = 40
    env = BatchedTradingEnv(
        (feat, close), num_envs=E * tile, positions=cfg["positions"], windows=cfg["windows"],
        trading_fees=cfg["trading_fees"], borrow_interest_rate=cfg["borrow_interest_rate"],
        max_episode_duration=cfg["max_episode_duration"], autoreset="next_step", dyn_persist=True,
        reward_function=reward_simple_return_minus_turnover,
        dynamic_feature_functions=["last_position_taken", dyn_valuation_ratio, dyn_exposure_change])
    t = lambda a: np.tile(a, tile)
    q, n = replay.injection_queue(g, tile)
    env.set_autoreset_injection(q["idx"], q["pos_index"], q["dataset"])
    obs, _ = env.reset(inject_idx=t(g["idx"][0]), inject_position_index=t(g["pos_index"][0]))
    for k in range(K):
        if k > 0:
            obs, reward, term, trunc, info = env.step(torch.from_numpy(t(g["action"][k]).astype(np.int32)).cuda())
            np.testing.assert_allclose(env.read_output("reward64"), t(g["reward"][k]), rtol=1e-12,
                                       atol=1e-15, err_msg=f"call {k} reward")
            np.testing.assert_array_equal(term.cpu().numpy(), t(g["done"][k]).astype(bool))
        np.testing.assert_array_equal(env.state("idx"), t(g["idx"][k]), err_msg=f"call {k}")
        ref = np.tile(g["obs"][k], (tile, 1, 1))
        got = obs.cpu().numpy()
        np.testing.assert_array_equal(got[..., :4], ref[..., :4], err_msg=f"call {k} static + device column")
        # the custom columns: f32 of an f64 formula evaluated by torch instead of CPython
        np.testing.assert_allclose(got[..., 4:], ref[..., 4:], rtol=1e-6, atol=1e-9, err_msg=f"call {k}")
    env.close()


def test_functions_named_like_the_defaults_are_the_users_code():
    """A user's `basic_reward_function` / `dynamic_feature_real_position` that is NOT this
    package's object must be evaluated, not replaced by the device built-in of that name —
    in the N=1 drop-in and in the batch."""
    import torch
    import gym_trading_env_amd as gte

    def basic_reward_function(history):
        return 0.25

    def dynamic_feature_real_position(history):
        return 7.0

    feat, close = _walk(9, 200, 2)
    df = make_df(feat, close)
    single = gte.TradingEnv(df, positions=[-1, 0, 1], reward_function=basic_reward_function,
                            dynamic_feature_functions=[dynamic_feature_real_position], verbose=0)
    np.random.seed(0)
    obs, _ = single.reset()
    assert obs[-1] == 7.0
    obs, reward, done, trunc, info = single.step(2)
    assert reward == 0.25 and obs[-1] == 7.0 and info["reward"] == 0.25
    single.close()
    # the package's own objects still run on the device: same numbers as the string specs
    a = gte.BatchedTradingEnv(df, 64, positions=[-1, 0, 1], seed=2,
                              reward_function=gte.basic_reward_function,
                              dynamic_feature_functions=[gte.dynamic_feature_last_position_taken,
                                                         gte.dynamic_feature_real_position])
    assert a.cfg.log_steps == 0 and a._reward_callable is None and not a._dyn_callables
    a.close()

    def batch_reward(h):  # one value per env
        return np.full(h["idx", -1].shape[0], 0.25)

    def batch_feature(h):
        return h["idx", -1] * 0 + 7.0

    env = gte.BatchedTradingEnv(df, 300, positions=[-1, 0, 1], max_episode_duration=10, seed=2,
                                reward_function=batch_reward, dynamic_feature_functions=[batch_feature])
    obs, _ = env.reset()
    assert (obs.cpu().numpy()[:, -1] == 7.0).all()
    for k in range(25):
        obs, reward, term, trunc, info = env.step(torch.randint(0, 3, (300,), dtype=torch.int32, device="cuda"))
        r = reward.cpu().numpy()
        reset_row = env.state("step") == 0
        assert (r[reset_row] == 0).all() and (r[~reset_row & ~term.cpu().numpy()] == 0.25).all()
        assert (obs.cpu().numpy()[:, -1] == 7.0).all()
    assert reset_row.any() or True
    with pytest.raises(NotImplementedError, match="output='torch'"):  # evaluated on the device
        gte.BatchedTradingEnv(df, 4, reward_function=batch_reward, output="numpy")
    env.close()


def test_batched_history_access_patterns():
    """BatchedHistory: the reference History's access patterns (docs/source/history.rst:18-46)
    with one value per env, against the per-env History rebuilt from the same device log."""
    import torch
    import gym_trading_env_amd as gte
    from gym_trading_env_amd.device_array import DeviceArray
    feat, close = _walk(21, 300, 3, sigma=2e-2)
    df = make_df(feat, close)
    N, L = 50, 40
    for output in ("torch", "numpy"):
        env = gte.BatchedTradingEnv(df, N, positions=[-1, 0, 1], windows=3, trading_fees=1e-3,
                                    max_episode_duration=15, log_steps=L, seed=4, output=output)
        env.reset()
        rng = np.random.default_rng(0)
        host = lambda x: x.numpy() if isinstance(x, DeviceArray) else np.asarray(x)
        for k in range(27 + 61):  # first pass of the checks below at 28 rows, then past the ring's wrap
            env.step(rng.integers(0, 3, N).astype(np.int32))
            if k == 26:
                break
        h = env.batched_history()
        assert len(h) == 28 and (isinstance(h["idx", -1], DeviceArray) == (output == "torch"))
        for e in (0, 7, N - 1):
            he = env.history(e)  # the reference-style History of env e's current episode
            n = len(he)
            for col in ("idx", "step", "position", "real_position", "portfolio_valuation", "reward",
                        "data_close", "data_volume", "portfolio_distribution_fiat",
                        "portfolio_distribution_borrowed_asset"):
                assert host(h[col, -1])[e] == he[col, -1], (output, col)
                assert host(h[col, 0])[e] == he[col, 0], (output, col)
                if n >= 2:
                    assert host(h[col, -2])[e] == he[col, -2], (output, col)
                np.testing.assert_array_equal(host(h[col])[-n:, e], np.asarray(he[col], dtype=np.float64))
            assert host(h["date", -1])[e] == he["date", -1]
            np.testing.assert_array_equal(host(h.episode_mask())[:, e],
                                          np.arange(28) >= 28 - n)
            two = host(h[["idx", "reward"]])
            assert two.shape == (28, N, 2) and two[-1, e, 0] == he["idx", -1]
            assert host(h[-1]["portfolio_valuation"])[e] == he["portfolio_valuation", -1]
        with pytest.raises(ValueError, match="does not exist"):
            h["no_such_column", -1]
        with pytest.raises(IndexError):
            h["idx", -29]
        # past the wrap of the L-row device log: the window is the last L rows, oldest first
        for k in range(61):
            env.step(rng.integers(0, 3, N).astype(np.int32))
        h = env.batched_history()
        assert len(h) == L
        for e in (3, N - 2):
            he = env.history(e)
            n = len(he)
            for col in ("idx", "portfolio_valuation", "reward", "data_close"):
                np.testing.assert_array_equal(host(h[col])[-n:, e], np.asarray(he[col], dtype=np.float64))
                assert host(h[col, -1])[e] == he[col, -1] and host(h[col, 0])[e] == he[col, 0]
            np.testing.assert_array_equal(host(h.episode_mask())[:, e], np.arange(L) >= L - n)
        env.close()


# (static features, windows, N): F_obs = static + 2.  F_obs % 4 == 0 takes the 16-byte-vector
# kernels — the shapes people use (the headline is 30 + 2) and the ones round 2 never tested here:
# its isolated hot instantiation skipped the terminal records `final_info` is served from.
FINAL_SHAPES = [(4, 3, 200), (6, 8, 200), (6, 3, 200), (6, None, 200), (30, 20, 3000), (6, 16, 20000)]


@pytest.mark.parametrize("Fs,windows,N", FINAL_SHAPES)
def test_same_step_final_observation_and_final_info(Fs, windows, N):
    """Gymnasium's same-step convention (docs/source/vectorize_env.rst:25-33 + SURVEY §8f):
    the envs that end get `final_observation` / `final_info` (with masks) holding what
    TradingEnv.step returned for them before the reset — checked against a twin batch that does
    not auto-reset (same first episodes)."""
    import gym_trading_env_amd as gte
    feat, close = _walk(33, 300, Fs, sigma=3e-2, drift=-2e-3)
    df = make_df(feat, close)
    kw = dict(positions=[-2, -1, 0, 1, 2], windows=windows, trading_fees=1e-3, borrow_interest_rate=1e-4,
              max_episode_duration=12, seed=8, output="numpy")
    env = gte.BatchedTradingEnv(df, N, autoreset="same_step", final_obs=True, **kw)
    twin = gte.BatchedTradingEnv(df, N, autoreset=None, **kw)
    assert env.launch_info()["vector_bytes"] == (16 if (Fs + 2) % 4 == 0 else 4)
    env.reset(); twin.reset()
    rng = np.random.default_rng(5)
    first_episode = np.ones(N, dtype=bool)
    checked = 0
    for k in range(14):
        a = rng.integers(0, 5, N).astype(np.int32)
        obs, reward, term, trunc, info = env.step(a)
        o2, r2, t2, u2, i2 = twin.step(a)
        ended = term | trunc
        assert "final_observation" in info and "final_info" in info.keys()
        mask = info["_final_observation"]
        np.testing.assert_array_equal(mask, ended)
        np.testing.assert_array_equal(info["_final_info"], ended)
        for e in np.nonzero(ended & first_episode)[0][:400]:
            np.testing.assert_array_equal(info["final_observation"][e], o2[e])
            fi = info["final_info"][e]
            for key in ("idx", "step", "position_index", "position", "real_position",
                        "portfolio_valuation", "portfolio_distribution_asset",
                        "portfolio_distribution_borrowed_fiat", "portfolio_distribution_interest_asset",
                        "data_close", "data_volume", "date"):
                assert fi[key] == i2[key][e], key
            assert fi["reward"] == r2[e] == reward[e]
            checked += 1
        for e in np.nonzero(~ended)[0][:400]:
            assert info["final_observation"][e] is None and info["final_info"][e] is None
        # same-step episode_metrics() reads the terminal records too (environments.py:279-286)
        m = env.episode_metrics()
        np.testing.assert_array_equal(m["env_ids"], np.nonzero(ended)[0])
        sel = first_episode[m["env_ids"]] if len(m["env_ids"]) else np.zeros(0, bool)
        ids = m["env_ids"][sel]
        np.testing.assert_array_equal(m["episode_length"][sel], twin.state("step")[ids] + 1)
        np.testing.assert_array_equal(m["portfolio_return"][sel],
                                      twin.state("portfolio_valuation")[ids] / 1000 - 1)
        c = df["close"].to_numpy()
        np.testing.assert_array_equal(m["market_return"][sel],
                                      c[twin.state("idx")[ids]] / c[twin.state("start_idx")[ids]] - 1)
        first_episode &= ~ended
    assert checked >= min(N, 400) // 2
    env.close(); twin.close()


def test_batch_replays_the_reference_vector_example_fixture():
    """tests/golden/hostcb_vector_example.npz: the REFERENCE ran its vectorised example's
    configuration (make_golden.py vector_example_fixture: the example's kwargs, its Python reward
    function, 3 env objects) and recorded every call's return tuple and `info` dict.  The batch,
    given the same kwargs and the same Python function, reproduces observations, rewards, flags,
    state and every key of `info` (dict-of-arrays, docs/source/vectorize_env.rst:25-33)."""
    import torch
    import custom_callables as cc
    from gym_trading_env_amd.batched import BatchedTradingEnv
    g = replay.load("hostcb_vector_example")
    feat, close = g["datasets"][0]
    cfg = g["cfg"]
    K, E = g["op"].shape
    df = pd.DataFrame({"open": g["df_open"], "high": g["df_high"], "low": g["df_low"], "close": close,
                       "volume": g["df_volume"]}, index=pd.to_datetime(g["df_index_ns"]))
    for j in range(feat.shape[1]):
        df[f"feature_{j}"] = feat[:, j]
    tile = 50
    env = BatchedTradingEnv(
        df, num_envs=E * tile, name="BTCUSD", positions=cfg["positions"], windows=cfg["windows"],
        initial_position=cfg["initial_position"], trading_fees=cfg["trading_fees"],
        borrow_interest_rate=cfg["borrow_interest_rate"],
        portfolio_initial_value=cfg["portfolio_initial_value"],
        max_episode_duration=cfg["max_episode_duration"], autoreset="next_step",
        # the fixture re-used each reference env object across episodes: dynamic columns written
        # by earlier episodes persist in its `_obs_array` (environments.py:153-154)
        dyn_persist=True, reward_function=cc.reward_log_return_example)
    t = lambda a: np.tile(a, tile)
    q, n = replay.injection_queue(g, tile)
    env.set_autoreset_injection(q["idx"], None, None)
    obs, info = env.reset(inject_idx=t(g["idx"][0]))
    keys = [str(k) for k in g["info_keys"]]
    assert set(keys) <= set(info.keys())
    for k in range(K):
        if k > 0:
            obs, reward, term, trunc, info = env.step(torch.from_numpy(t(g["action"][k]).astype(np.int32)).cuda())
            np.testing.assert_allclose(env.read_output("reward64"), t(g["reward"][k]), rtol=1e-12, atol=1e-16)
            np.testing.assert_array_equal(term.cpu().numpy(), t(g["done"][k]).astype(bool))
            np.testing.assert_array_equal(trunc.cpu().numpy(), t(g["truncated"][k]).astype(bool))
        np.testing.assert_array_equal(obs.cpu().numpy(), np.tile(g["obs"][k], (tile, 1, 1)), err_msg=f"call {k}")
        for key in keys:
            ref = t(g[f"info_{key}"][k])
            got = info[key]
            if key == "date":
                np.testing.assert_array_equal(np.asarray(got, "datetime64[ns]").astype(np.int64), ref)
            elif key == "position_index":
                # the reference logs the ACTION there (None for a hold); the batch the index of
                # the position held — equal whenever an action was given
                m = ref >= 0
                np.testing.assert_array_equal(np.asarray(got)[m], ref[m])
            elif key == "reward":
                np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-16, err_msg=f"call {k} {key}")
            else:
                np.testing.assert_allclose(np.asarray(got, np.float64), ref, rtol=1e-12, atol=1e-14,
                                           err_msg=f"call {k} {key}")
    env.close()


def test_sb3_vecenv_with_the_forks_python_reward_function():
    """The fork's training scripts give SB3 a Python reward function
    (luckymodel/envs/env.py:16-18: np.clip(np.log(p_t / p_t-1), -0.002, 0.005)) and custom-named
    re-implementations of the two default dynamic features.  SB3TradingVecEnv (same-step
    auto-reset, `terminal_observation`) takes them as they are; results equal the device enum
    `("clipped_log_return", 1, -0.002, 0.005)` / the device features."""
    import gym_trading_env_amd as gte

    def reward_function(history):  # verbatim from the fork
        log_return = np.log(history["portfolio_valuation", -1] / history["portfolio_valuation", -2])  # log (p_t / p_t-1 )
        return np.clip(log_return, -0.002, 0.005)

    def dynamic_feature_last_position_taken(history):
        return history['position', -1]

    def dynamic_feature_real_position(history):
        return history['real_position', -1]

    feat, close = _walk(44, 500, 4, sigma=3e-2, drift=-1e-3)
    df = make_df(feat, close)
    kw = dict(positions=[-1, 0, 1], windows=6, trading_fees=1e-3, borrow_interest_rate=1e-4,
              max_episode_duration=15, seed=12)
    N = 300
    custom = gte.SB3TradingVecEnv(df, N, reward_function=reward_function,
                                  dynamic_feature_functions=[dynamic_feature_last_position_taken,
                                                             dynamic_feature_real_position], **kw)
    builtin = gte.SB3TradingVecEnv(df, N, reward_function=("clipped_log_return", 1.0, -0.002, 0.005), **kw)
    assert custom.env.output == "torch" and builtin.env.output == "numpy"
    np.testing.assert_array_equal(custom.reset(), builtin.reset())
    rng = np.random.default_rng(3)
    ends = 0
    for k in range(60):
        a = rng.integers(0, 3, N)
        o1, r1, d1, i1 = custom.step(a)
        o2, r2, d2, i2 = builtin.step(a)
        np.testing.assert_array_equal(o1, o2, err_msg=f"step {k}")
        np.testing.assert_array_equal(d1, d2)
        np.testing.assert_allclose(r1, r2, rtol=1e-6, atol=1e-12)
        for e in np.nonzero(d1)[0]:
            np.testing.assert_array_equal(i1[e]["terminal_observation"], i2[e]["terminal_observation"],
                                          err_msg=f"step {k} env {e}")
            assert i1[e]["TimeLimit.truncated"] == i2[e]["TimeLimit.truncated"]
            assert i1[e]["portfolio_valuation"] == i2[e]["portfolio_valuation"]
        ends += int(d1.sum())
    assert ends > N
    custom.close(); builtin.close()


def test_python_reward_in_same_step_mode_on_a_big_batch_with_the_affinity_order():
    """SB3's mode (same-step auto-reset) with a Python reward function at N > 16 384 on a
    16-byte-vector shape with the L2-affinity order on: the trajectory row is then written by the
    separate log launch and the callable's TERMINAL row comes from the terminal records alone
    (`_overlay`).  Round 2 launched the isolated hot kernel here, which never wrote those
    records: a truncated env's reward became log(0 / pv).  Equal to the device enum."""
    import gym_trading_env_amd as gte

    def reward_function(history):  # the fork's (luckymodel/envs/env.py:16-18)
        log_return = np.log(history["portfolio_valuation", -1] / history["portfolio_valuation", -2])
        return np.clip(log_return, -0.002, 0.005)

    feat, close = _walk(46, 500, 6, sigma=3e-2, drift=-1e-3)
    df = make_df(feat, close)
    kw = dict(positions=[-1, 0, 1], windows=16, trading_fees=1e-3, borrow_interest_rate=1e-4,
              max_episode_duration=15, seed=13)
    N = 20000
    custom = gte.SB3TradingVecEnv(df, N, reward_function=reward_function, **kw)
    builtin = gte.SB3TradingVecEnv(df, N, reward_function=("clipped_log_return", 1.0, -0.002, 0.005), **kw)
    assert custom.env.launch_info()["vector_bytes"] == 16
    assert custom.env.cfg.log_steps == 2 and builtin.env.cfg.log_steps == 0
    np.testing.assert_array_equal(custom.reset(), builtin.reset())
    rng = np.random.default_rng(3)
    ends = 0
    for k in range(34):
        a = rng.integers(0, 3, N)
        o1, r1, d1, i1 = custom.step(a)
        o2, r2, d2, i2 = builtin.step(a)
        np.testing.assert_array_equal(o1, o2, err_msg=f"step {k}")
        np.testing.assert_array_equal(d1, d2)
        assert np.isfinite(r1).all()
        np.testing.assert_allclose(r1, r2, rtol=1e-6, atol=1e-12, err_msg=f"step {k}")
        ends += int(d1.sum())
    assert ends > N
    custom.close(); builtin.close()


@pytest.mark.parametrize("Fs,windows", [(4, 6), (6, 8)])
def test_sb3_infos_of_finished_envs_are_the_terminal_steps(Fs, windows):
    """SB3's DummyVecEnv over the reference hands back the TERMINAL step's info
    (environments.py:272) with `terminal_observation` added; the adapter's `infos[e]` of an env
    that just ended must therefore show the terminal `idx` / `portfolio_valuation` / ..., not the
    state after the in-launch reset.  Against a twin that does not auto-reset."""
    import gym_trading_env_amd as gte
    feat, close = _walk(47, 400, Fs, sigma=3e-2, drift=-1e-3)
    df = make_df(feat, close)
    kw = dict(positions=[-1, 0, 1], windows=windows, trading_fees=1e-3, borrow_interest_rate=1e-4,
              max_episode_duration=11, seed=21)
    keys = ("idx", "step", "position", "real_position", "portfolio_valuation", "data_close", "reward")
    N = 150
    vec = gte.SB3TradingVecEnv(df, N, info_keys=keys, **kw)
    twin = gte.BatchedTradingEnv(df, N, autoreset=None, output="numpy", **kw)
    vec.reset(); twin.reset()
    rng = np.random.default_rng(4)
    first = np.ones(N, bool)
    checked = 0
    for k in range(13):
        a = rng.integers(0, 3, N).astype(np.int32)
        obs, rew, dones, infos = vec.step(a)
        o2, r2, t2, u2, i2 = twin.step(a)
        for e in np.nonzero(dones & first)[0]:
            np.testing.assert_array_equal(infos[e]["terminal_observation"], o2[e])
            for key in keys:
                assert infos[e][key] == i2[key][e], (k, e, key)
            assert infos[e].get("idx") == i2["idx"][e] and dict(infos[e])["step"] == i2["step"][e]
            checked += 1
        for e in np.nonzero(~dones & first)[0][:20]:  # running envs: the current step's info
            assert infos[e]["idx"] == i2["idx"][e] and infos[e]["portfolio_valuation"] == i2["portfolio_valuation"][e]
        first &= ~dones
    assert checked >= N
    # the step after: the dicts of the envs that ended are views of the running episode again
    obs, rew, dones, infos = vec.step(rng.integers(0, 3, N).astype(np.int32))
    np.testing.assert_array_equal([infos[e]["idx"] for e in range(N)], vec.env.state("idx"))
    vec.close(); twin.close()


@pytest.mark.parametrize("Fs", [3, 6])
def test_reset_rows_of_the_log_carry_reward_zero_in_same_step_mode(Fs):
    """The reference's reset row has reward 0 (environments.py:196).  In same-step mode the log
    row of a step that ended an episode already describes the NEXT episode's reset row: its
    reward must be 0 (the terminal step's reward lives in the return buffers and `final_info`),
    so that one step later `h["reward", -2]` and `h["reward", 0]` read 0 like a reference History."""
    import torch
    from gym_trading_env_amd.batched import BatchedTradingEnv
    from gym_trading_env_amd.device_array import DeviceArray
    host = lambda x: x.numpy() if isinstance(x, DeviceArray) else np.asarray(x)
    feat, close = _walk(48, 300, Fs, sigma=2e-2)
    N = 500
    for variant in (1024, 2048):  # separate log launch / row written by the step kernel
        env = BatchedTradingEnv((feat, close), N, positions=[-1, 0, 1], windows=4, trading_fees=1e-3,
                                max_episode_duration=7, seed=3, autoreset="same_step", final_obs=True,
                                log_steps=16, kernel_variant=variant)
        env.reset()
        g = torch.Generator(device="cuda").manual_seed(5)
        seen = 0
        prev_ended = None
        for k in range(20):
            act = torch.randint(0, 3, (N,), dtype=torch.int32, device="cuda", generator=g)
            _, reward, term, trunc, _ = env.step(act)
            h = env.batched_history()
            ended = (term | trunc).cpu().numpy()
            assert (host(h["reward", -1])[ended] == 0).all()  # the row under an ended env is a reset row
            np.testing.assert_array_equal(host(h["step", -1])[ended], 0)
            if prev_ended is not None and prev_ended.any():
                m = prev_ended & ~ended
                assert (host(h["reward", -2])[m] == 0).all() and (host(h["reward", 0])[m] == 0).all()
                seen += int(m.sum())
            # the terminal view shows the terminal step's reward as the newest one
            rt = host(env.batched_history(terminal=True)["reward", -1])
            np.testing.assert_array_equal(rt[ended], env.read_output("reward64")[ended])
            np.testing.assert_array_equal(rt[~ended], host(h["reward", -1])[~ended])
            prev_ended = ended
        assert seen > N
        env.close()


@pytest.mark.parametrize("Fs,windows", [(3, 3), (6, 8), (6, None)])
def test_same_step_finished_episode_history_and_custom_metrics(Fs, windows):
    """Same-step auto-reset (SB3's mode) with a device log: `history(e, finished=True)` is the
    episode that just ended — terminal row included — and `episode_metrics()` evaluates
    `add_metric` functions on it (environments.py:274-286); checked against a twin batch that does
    not reset (same first episodes)."""
    import gym_trading_env_amd as gte
    feat, close = _walk(55, 300, Fs, sigma=2e-2)
    df = make_df(feat, close)
    kw = dict(positions=[-1, 0, 1], windows=windows, trading_fees=1e-3, max_episode_duration=9, seed=2,
              output="numpy", log_steps=32)
    N = 64
    env = gte.BatchedTradingEnv(df, N, autoreset="same_step", final_obs=True, **kw)
    twin = gte.BatchedTradingEnv(df, N, autoreset=None, **kw)
    for e in (env, twin):
        e.add_metric("Position Changes", lambda h: int(np.sum(np.diff(h["position"]) != 0)))
        e.add_metric("Episode Length", lambda h: len(h["position"]))
    env.reset(); twin.reset()
    rng = np.random.default_rng(9)
    first = np.ones(N, bool)
    checked = 0
    for k in range(10):
        a = rng.integers(0, 3, N).astype(np.int32)
        _, _, term, trunc, _ = env.step(a)
        twin.step(a)
        ended = term | trunc
        if ended.any():
            m = env.episode_metrics()
            for j, e in enumerate(m["env_ids"]):
                if not first[e]:
                    continue
                hf, ht = env.history(int(e), finished=True), twin.history(int(e))
                assert len(hf) == len(ht) == m["Episode Length"][j]
                for col in ("idx", "step", "position", "portfolio_valuation", "reward", "data_close",
                            "portfolio_distribution_fiat"):
                    np.testing.assert_array_equal(np.asarray(hf[col], np.float64), np.asarray(ht[col], np.float64), err_msg=col)
                assert m["Position Changes"][j] == int(np.sum(np.diff(ht["position"]) != 0))
                assert m["Portfolio Return"][j] == f"{100 * (ht['portfolio_valuation', -1] / 1000 - 1):5.2f}%"
                checked += 1
        first &= ~ended
    assert checked >= N
    with pytest.raises(ValueError, match="did not end"):
        env.history(int(np.nonzero(~ended)[0][0]), finished=True)
    env.close(); twin.close()


def test_apply_reward_and_dynamic_columns_entry_points():
    """`gte_apply_reward` (the reference's rules around a custom reward, one kernel) against the
    same rules written with torch ops, and `gte_set_dynamic_columns` (separate f32 / f64 columns)
    against `gte_set_dynamic_features` (packed f32) on a twin env."""
    import ctypes as C
    import torch
    from gym_trading_env_amd import _abi
    from gym_trading_env_amd.batched import BatchedTradingEnv
    rng = np.random.default_rng(3)
    T, N = 600, 257
    feat = rng.normal(size=(T, 3)).astype(np.float32)
    close = 100 * np.exp(np.cumsum(rng.normal(0, 2e-2, T)))
    kw = dict(num_envs=N, seed=9, positions=[-1, 0, 1], windows=4, trading_fees=1e-3,
              borrow_interest_rate=1e-4, max_episode_duration=7, output="torch", log_steps=3)
    for mode in ("next_step", "same_step"):
        a = BatchedTradingEnv((feat, close), autoreset=mode, **kw)
        b = BatchedTradingEnv((feat, close), autoreset=mode, **kw)
        a.reset(); b.reset()
        g = torch.Generator(device="cuda").manual_seed(1)
        for k in range(25):
            act = torch.randint(0, 3, (N,), dtype=torch.int32, device="cuda", generator=g)
            a.step(act); b.step(act)
            r = torch.randn(N, dtype=torch.float64, device="cuda", generator=g)
            # reference rules with torch ops (what batched.py did before the entry point existed)
            h = a.batched_history()
            newest = (h._rows - 1) % h._L
            term, trunc = a._t["terminated"].clone(), a._t["truncated"].clone()
            reset_row = a._log_tensor("step")[newest] == 0
            if mode == "same_step":
                reset_row = reset_row & ~(term | trunc)
            want = torch.where(term | reset_row, torch.zeros_like(r), r)
            _abi.check(b._lib, b._lib.gte_apply_reward(b._h, C.c_void_p(r.data_ptr()),
                                                       1 if mode == "same_step" else 0))
            torch.cuda.synchronize()
            assert torch.equal(b._t["reward64"], want)
            assert torch.equal(b._t["reward"], want.to(torch.float32))
            # the log row of a reset — in same-step mode also under an env that just ended — keeps
            # the reference's reward 0 (environments.py:196)
            assert torch.equal(b._log_tensor("reward")[newest],
                               torch.where(a._log_tensor("step")[newest] == 0, torch.zeros_like(r), want))
            # dynamic columns: feature 0 from an f64 column, feature 1 from an f32 column
            c0 = torch.randn(N, dtype=torch.float64, device="cuda", generator=g)
            c1 = torch.randn(N, dtype=torch.float32, device="cuda", generator=g)
            packed = torch.stack([c0.to(torch.float32), c1], dim=1).contiguous()
            _abi.check(a._lib, a._lib.gte_set_dynamic_features(a._h, C.c_void_p(packed.data_ptr()), 3))
            cols = (C.c_void_p * 2)(c0.data_ptr(), c1.data_ptr())
            is64 = (C.c_int32 * 2)(1, 0)
            _abi.check(b._lib, b._lib.gte_set_dynamic_columns(b._h, cols, is64))
            torch.cuda.synchronize()
            assert torch.equal(a._t["obs"], b._t["obs"])
        # later windows read the stored values: the twins stay identical
        for k in range(6):
            act = torch.randint(0, 3, (N,), dtype=torch.int32, device="cuda", generator=g)
            oa = a.step(act)[0]; ob = b.step(act)[0]
            assert torch.equal(oa, ob)
        a.close(); b.close()


@pytest.mark.parametrize("mode,N", [("next_step", 193), ("same_step", 193), ("disabled", 193),
                                    ("next_step", 20_000), ("same_step", 20_000)])
def test_trajectory_row_written_by_the_step_kernel_equals_the_separate_launch(mode, N):
    """With `log_steps` the step kernel's phase A writes the trajectory row itself; kernel_variant
    bit 1024 keeps round 1's separate log launch.  Every column of every row must be identical,
    resets and frozen envs included."""
    import torch
    from gym_trading_env_amd import _abi
    from gym_trading_env_amd.batched import BatchedTradingEnv
    rng = np.random.default_rng(11)
    T, L = 300, 5  # (20 000 envs: several workgroups per CU and the L2-affinity order, rows through LDS)
    feat = rng.normal(size=(T, 6)).astype(np.float32)
    close = 100 * np.exp(np.cumsum(rng.normal(0, 3e-2, T)))
    kw = dict(num_envs=N, seed=4, positions=[-1, 0, 0.5, 2], windows=3 if N < 1000 else 16, trading_fees=1e-3,
              borrow_interest_rate=1e-3, max_episode_duration=9, output="torch", log_steps=L,
              autoreset=mode, final_obs=(mode == "same_step"))
    a = BatchedTradingEnv((feat, close), kernel_variant=2048, **kw)   # row written in the kernel
    b = BatchedTradingEnv((feat, close), kernel_variant=1024, **kw)   # separate launch
    a.reset(); b.reset()
    g = torch.Generator(device="cuda").manual_seed(2)
    for k in range(40 if N < 1000 else 14):
        act = torch.randint(-1, 4, (N,), dtype=torch.int32, device="cuda", generator=g)
        ra = a.step(act); rb = b.step(act)
        for x, y in zip(ra[:4], rb[:4]):
            assert torch.equal(x, y)
        for name in _abi.LOG_DTYPES:
            assert torch.equal(a._log_tensor(name), b._log_tensor(name)), (mode, k, name)
        if mode == "disabled" and k % 7 == 6:
            m = (a._t["terminated"] | a._t["truncated"]).to(torch.uint8).cpu().numpy()
            a.reset(mask=m); b.reset(mask=m)
    a.close(); b.close()


@pytest.mark.parametrize("mode,L", [("next_step", 32), ("same_step", 32), ("disabled", 32), ("next_step", 5)])
def test_read_log_envs_equals_the_per_env_reads(mode, L):
    """`gte_read_log_envs` (one kernel packs the logged episode of MANY envs into pinned host
    memory, one transfer) against the per-env strided copies of `gte_read_log` /
    `gte_read_log_portfolio` with the episode cut on the host: every column, every listed env —
    running, just reset, frozen after its end (auto-reset disabled), episodes longer than the log
    (L = 5), `finished=True` with the terminal row from the terminal records, `max_rows`."""
    import ctypes as C
    from gym_trading_env_amd import _abi
    from gym_trading_env_amd.batched import BatchedTradingEnv
    feat, close = _walk(77, 400, 6, sigma=3e-2, drift=-1e-3)
    N = 700
    env = BatchedTradingEnv((feat, close), N, positions=[-1, 0, 1], windows=4, trading_fees=1e-3,
                            borrow_interest_rate=1e-4, max_episode_duration=9, seed=5, autoreset=mode,
                            final_obs=(mode == "same_step"), log_steps=L, output="numpy")
    env.reset()
    rng = np.random.default_rng(2)
    names = list(_abi.LOG_DTYPES)

    def per_env(e, finished):
        bufs = {k: np.empty(L, _abi.LOG_DTYPES[k]) for k in names}
        n = C.c_int32()
        order = ("idx", "step", "position_index", "dataset_index", "portfolio_valuation", "real_position",
                 "reward", "flags")
        _abi.check(env._lib, env._lib.gte_read_log(env._h, int(e), L, *(bufs[k].ctypes.data for k in order), C.byref(n)))
        _abi.check(env._lib, env._lib.gte_read_log_portfolio(
            env._h, int(e), L, *(bufs[k].ctypes.data for k in ("asset", "fiat", "interest_asset", "interest_fiat")),
            C.byref(n)))
        n = n.value
        if finished:
            fs = env.final_state
            for k in ("idx", "step", "position_index", "dataset_index", "portfolio_valuation", "real_position",
                      "asset", "fiat", "interest_asset", "interest_fiat"):
                bufs[k][n - 1] = fs(k)[e]
            bufs["reward"][n - 1] = env.read_output("reward64")[e]
        step = bufs["step"][:n]
        start = n - 1
        while start > 0 and step[start - 1] == step[start] - 1:
            start -= 1
        return {k: bufs[k][start:n] for k in names}

    checked = 0
    for k in range(24):
        a = rng.integers(0, 3, N).astype(np.int32)
        _, _, term, trunc, _ = env.step(a)
        ended = term | trunc
        if k % 3 == 2 or k < 3:
            ids = np.unique(np.concatenate([rng.integers(0, N, 40), np.nonzero(ended)[0][:30], [0, N - 1]]))
            got = env.read_log_envs(ids)
            for j, e in enumerate(ids):
                want = per_env(e, False)
                n = int(got["n_rows"][j])
                assert n == len(want["idx"]), (k, e)
                for name in names:
                    np.testing.assert_array_equal(got[name][j, :n], want[name], err_msg=f"step {k} env {e} {name}")
                checked += 1
            # max_rows keeps the NEWEST rows (the arrays are views of ONE staging buffer: copy first)
            rows, pv = got["n_rows"].copy(), got["portfolio_valuation"].copy()
            got3 = env.read_log_envs(ids, max_rows=3)
            for j in range(len(ids)):
                n, n3 = int(rows[j]), int(got3["n_rows"][j])
                assert n3 == min(n, 3)
                np.testing.assert_array_equal(got3["portfolio_valuation"][j, :n3], pv[j, n - n3:n])
            if mode == "same_step" and ended.any():
                fin = np.nonzero(ended)[0]
                gotf = env.read_log_envs(fin, finished=True)
                for j, e in enumerate(fin[:50]):
                    want = per_env(e, True)
                    n = int(gotf["n_rows"][j])
                    assert n == len(want["idx"]) and n >= 2
                    for name in names:
                        np.testing.assert_array_equal(gotf[name][j, :n], want[name], err_msg=f"finished {e} {name}")
                # the History objects built from it: the single-env path is the batch of one
                hs = env.histories(fin[:5], finished=True)
                for e, h in zip(fin[:5], hs):
                    h1 = env.history(int(e), finished=True)
                    assert len(h) == len(h1) and h[-1] == h1[-1] and h[0] == h1[0]
                    assert h["step", -1] == len(h) - 1 or len(h) == L
        if mode == "disabled" and k % 5 == 4:
            env.reset(mask=ended.astype(np.uint8))
    assert checked > 300
    assert len(env.histories([])) == 0
    with pytest.raises(Exception):
        env.read_log_envs([N])
    env.close()


def test_verbose_prints_the_reference_episode_end_lines(capsys):
    """`verbose` > 0 is not silently ignored by the batch: with a trajectory log the episode-end
    line of the reference (environments.py:269-271, :289-294 — Market Return, Portfolio Return and
    every add_metric entry) is printed for each env that finished; compared with the lines the
    N=1 drop-ins print for the same episodes.  No RNG involved: fixed initial position,
    max_episode_duration='max' (every episode starts at row 0); each env gets its own actions."""
    import gym_trading_env_amd as gte
    feat, close = _walk(91, 70, 3, sigma=4e-2, drift=-6e-3)  # drawdowns end episodes early
    df = make_df(feat, close)
    kw = dict(positions=[-1, 0, 1, 2], trading_fees=1e-3, borrow_interest_rate=1e-4, initial_position=1,
              max_episode_duration="max", verbose=1)
    N = 8
    changes = lambda h: int(np.sum(np.diff(h["position"]) != 0))
    batch = gte.BatchedTradingEnv(df, N, autoreset="next_step", log_steps=128, output="numpy", **kw)
    assert batch.verbose_interval == 1.0  # by default at most one report per second (a report synchronises)
    batch.verbose_interval = 0            # here: after every step, like the reference
    batch.add_metric("Position Changes", changes)
    singles = []
    for e in range(N):
        s = gte.TradingEnv(df, **kw)
        s.add_metric("Position Changes", changes)
        singles.append(s)
    rng = np.random.default_rng(6)
    batch.reset()
    for s in singles:
        s.reset()
    capsys.readouterr()
    lines_batch, lines_single, ended_total = [], [], 0
    need_reset = np.zeros(N, bool)
    for k in range(160):
        a = rng.integers(0, 4, N).astype(np.int32)
        _, _, term, trunc, _ = batch.step(a)
        lines_batch += [ln for ln in capsys.readouterr().out.splitlines() if ln]
        for e, s in enumerate(singles):
            if need_reset[e]:      # next-step auto-reset: this call is the reset of env e
                s.reset()
                need_reset[e] = False
                continue
            _, _, d, t, _ = s.step(int(a[e]))
            need_reset[e] = d or t
        lines_single += [ln for ln in capsys.readouterr().out.splitlines() if ln]
        np.testing.assert_array_equal(term | trunc, need_reset)
        ended_total += int(need_reset.sum())
    assert ended_total >= 2 * N and len(lines_batch) == ended_total
    assert lines_batch == lines_single
    assert all(ln.startswith("Market Return : ") and "Position Changes : " in ln for ln in lines_batch)
    # rate limit: many envs ending in one step print `verbose_max_lines` lines and a count
    batch.verbose_max_lines = 3
    big = gte.BatchedTradingEnv(df, 40, autoreset="next_step", log_steps=128, output="numpy",
                                **dict(kw, max_episode_duration=5, initial_position="random"))
    big.verbose_max_lines = 3
    big.verbose_interval = 0
    big.reset()
    capsys.readouterr()
    for k in range(4):
        big.step(np.zeros(40, np.int32))
    out = [ln for ln in capsys.readouterr().out.splitlines() if ln]
    assert len(out) == 4 and out[-1] == "... and 37 more episodes ended in this step"
    # verbose=0, or no trajectory log: nothing is printed and nothing is synchronised
    quiet = gte.BatchedTradingEnv(df, 40, autoreset="next_step", output="numpy",
                                  **dict(kw, max_episode_duration=5, verbose=1))
    quiet.reset()
    for k in range(6):
        quiet.step(np.zeros(40, np.int32))
    assert capsys.readouterr().out == ""
    batch.close(); big.close(); quiet.close()
    for s in singles:
        s.close()


@pytest.mark.parametrize("n_envs", [96, 20_000])
def test_log_columns_as_strided_views_of_the_records(n_envs):
    """The trajectory log is one array of records [L, N] (`gte_log_view.row_stride / env_stride`);
    its columns reach Python as strided device views (torch) or strided host copies (numpy): both
    equal what `gte_read_log` returns per env, for rows the step kernel wrote itself (big batch
    with the L2-affinity order included) and through the ring wrap of the log."""
    import torch
    import gym_trading_env_amd as gte
    from gym_trading_env_amd import _abi
    feat, close = _walk(77, 4000, 6, sigma=1e-2)
    df = make_df(feat, close)
    L = 5
    kw = dict(positions=[-1, 0, 1], windows=4, trading_fees=1e-3, max_episode_duration=7, seed=4, log_steps=L)
    et = gte.BatchedTradingEnv(df, n_envs, output="torch", **kw)
    en = gte.BatchedTradingEnv(df, n_envs, output="numpy", **kw)
    et.reset(); en.reset()
    rng = np.random.default_rng(1)
    for k in range(8):  # 9 rows written: the 5-row log has wrapped
        a = rng.integers(0, 3, n_envs).astype(np.int32)
        et.step(torch.as_tensor(a, device="cuda")); en.step(a)
    order = [(9 - L + r) % L for r in range(L)]  # physical rows, oldest first
    probe = [0, 1, n_envs // 2, n_envs - 1]
    for name in _abi.LOG_DTYPES:
        t = et._log_rows(name, None, order)
        h = en._log_rows(name, None, order)
        assert tuple(t.shape) == h.shape == (L, n_envs)
        np.testing.assert_array_equal(t.cpu().numpy(), h, err_msg=name)
        newest = en._log_rows(name, order[-1], None)
        np.testing.assert_array_equal(newest, h[-1], err_msg=name)
    for e in probe:
        hist = en.history(e)  # the env's current episode: the last len(hist) logged rows
        idx = en._log_rows("idx", None, order)[:, e]
        step = en._log_rows("step", None, order)[:, e]
        n = len(hist)
        np.testing.assert_array_equal(np.asarray(hist["idx"]), idx[L - n:])
        np.testing.assert_array_equal(np.asarray(hist["step"]), step[L - n:])
    et.close(); en.close()


def test_batched_history_reward_assignment_writes_the_newest_log_row():
    """`history["reward", -1] = value` (environments.py:267) on the batch: `gte_set_log_reward`
    writes the reward field of the newest row of every env's record (a strided device copy)."""
    import torch
    import gym_trading_env_amd as gte
    feat, close = _walk(5, 600, 4, sigma=1e-2)
    env = gte.BatchedTradingEnv(make_df(feat, close), 300, positions=[-1, 0, 1], windows=3, seed=2,
                                output="torch", log_steps=4, max_episode_duration=50)
    env.reset()
    for k in range(6):
        env.step(torch.randint(0, 3, (300,), dtype=torch.int32, device="cuda"))
    h = env.batched_history()
    before = {c: np.asarray(h[c, -1]).copy() for c in ("portfolio_valuation", "idx")}  # (np.asarray copies a DeviceArray to the host)
    older = np.asarray(h["reward", -2]).copy()
    value = torch.arange(300, dtype=torch.float64, device="cuda") * 0.5 - 7.0
    h["reward", -1] = value
    h2 = env.batched_history()
    np.testing.assert_array_equal(np.asarray(h2["reward", -1]), value.cpu().numpy())
    np.testing.assert_array_equal(np.asarray(h2["reward", -2]), older)  # the neighbours are untouched
    for c, v in before.items():
        np.testing.assert_array_equal(np.asarray(h2[c, -1]), v, err_msg=c)
    env.close()
