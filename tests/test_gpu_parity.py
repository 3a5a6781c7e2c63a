"""Parity of the HIP path (through the C ABI) with (a) the golden traces captured
from the reference and (b) the oracle on seeded random inputs.  Needs an MI355X."""
import numpy as np
import pytest

import replay
from gym_trading_env_amd.config import make_config

pytestmark = pytest.mark.gpu


class GpuAdapter:
    """BatchedTradingEnv behind the replay interface; every value read back through
    the C ABI (gte_copy_to_host), no torch involved."""

    def __init__(self, g, tile=1, **over):
        from gym_trading_env_amd.batched import BatchedTradingEnv
        kw = replay.config_kwargs(g, tile, **over)
        n_envs = kw.pop("n_envs")
        kw.pop("n_static"); kw.pop("n_datasets")
        self.env = BatchedTradingEnv(list(g["datasets"]) if len(g["datasets"]) > 1 else g["datasets"][0],
                                     num_envs=n_envs, output="numpy", **kw)

    def reset(self, mask, idx, pos, ds):
        self.env.reset(mask=mask, inject_idx=idx, inject_position_index=pos, inject_dataset=ds)

    def set_autoreset_injection(self, idx, pos, ds):
        self.env.set_autoreset_injection(idx, pos, ds)

    def step(self, actions):
        self.env.step(np.asarray(actions, np.int32))

    def add_limit_orders(self, pos, limit, persistent):
        self.env.add_limit_order(pos, limit, persistent)

    obs = lambda s: s.env.read_output("obs")
    reward64 = lambda s: s.env.read_output("reward64")
    terminated = lambda s: s.env.read_output("terminated")
    truncated = lambda s: s.env.read_output("truncated")

    def state(self):
        names = ("idx", "step", "position_index", "dataset_index", "asset", "fiat",
                 "interest_asset", "interest_fiat", "portfolio_valuation", "real_position")
        return {n: self.env.state(n) for n in names}


@pytest.mark.parametrize("name", replay.golden_names())
def test_hip_matches_reference_trace(name):
    """idx/step/position/done/truncated bit-exact; fp64 portfolio state, rewards and
    observations against the reference's own outputs (north-star tolerance: 1e-6
    relative; measured: state bit-exact, reward <= 1 ulp of f64)."""
    g = replay.load(name)
    if g["op"].shape[0] > 700:  # keep per-call D2H round trips bounded
        for k in list(g):
            if isinstance(g[k], np.ndarray) and g[k].ndim >= 2 and g[k].shape[0] == g["op"].shape[0]:
                g[k] = g[k][:700]
    a = GpuAdapter(g)
    worst = replay.replay(a, g, rtol=1e-12)
    assert worst <= 1e-12
    a.env.close()


@pytest.mark.parametrize("name,epw", [("c3_window20", 1), ("c3_window20", 4), ("c2_nowindow", 64),
                                      ("drawdown_done", 2), ("multidataset_k1", 8),
                                      ("c3_window20", 13), ("limit_orders", 7), ("c2_nowindow", 5)])
def test_hip_trace_tiled_across_waves(name, epw):
    """Same traces, envs tiled 67x so that they span many wavefronts / workgroups and a
    ragged last wave, for several envs-per-wave geometries."""
    g = replay.load(name)
    K = min(g["op"].shape[0], 160)
    for k in list(g):
        if isinstance(g[k], np.ndarray) and g[k].ndim >= 2 and g[k].shape[0] == g["op"].shape[0]:
            g[k] = g[k][:K]
    a = GpuAdapter(g, tile=67, envs_per_wave=epw)
    replay.replay(a, g, tile=67, rtol=1e-12)
    a.env.close()


def _synthetic(seed, T, n_static, sigma=5e-3, drift=0.0):
    rng = np.random.default_rng(seed)
    close = 100.0 * np.exp(np.cumsum(rng.normal(drift, sigma, T)))
    feat = rng.normal(0, 1, (T, n_static)).astype(np.float32)
    return feat, close


def _compare_with_oracle(oracle_mod, datasets, n_envs, steps, seed, check_every=1, p_order=0.0,
                         **kw):
    """Drive the HIP env and the oracle with the same config, the same device-RNG
    seed (no injection: Philox draws on both sides) and the same actions."""
    from gym_trading_env_amd.batched import BatchedTradingEnv
    env = BatchedTradingEnv(datasets if len(datasets) > 1 else datasets[0], num_envs=n_envs,
                            output="numpy", seed=seed, **kw)
    n_dyn = env.cfg.n_dyn
    staged = []
    for ds in datasets:
        f = ds[0]
        full = np.zeros((f.shape[0], f.shape[1] + n_dyn), np.float32)
        full[:, :f.shape[1]] = f
        staged.append((full,) + tuple(ds[1:]))
    ora = oracle_mod.OracleEnv(env.cfg, staged)
    env.reset()
    ora.reset()
    rng = np.random.default_rng(seed + 1)
    P = len(env.positions)
    names_i = ("idx", "step", "position_index", "dataset_index", "start_idx", "episode", "needs_reset")
    names_f = ("asset", "fiat", "interest_asset", "interest_fiat", "portfolio_valuation", "real_position")
    n_term = 0
    for k in range(steps + 1):
        if k > 0:
            a = rng.integers(-1, P, n_envs).astype(np.int32)
            if p_order > 0:  # limit orders, persistent and not, on a random subset of envs
                pi = np.where(rng.random(n_envs) < p_order, rng.integers(0, P, n_envs), -1).astype(np.int32)
                idx_now = ora.state()["idx"]
                px = np.array([datasets[d][1][i] for d, i in zip(ora.state()["dataset_index"], idx_now)])
                lim = px * (1 + rng.normal(0, 0.01, n_envs))
                per = (rng.random(n_envs) < 0.5).astype(np.uint8)
                env.add_limit_order(pi, lim, per)
                ora.add_limit_orders(pi, lim, per)
            env.step(a)
            ora.step(a, threads=8)
        if k % check_every and k != steps:
            continue
        so = ora.state()
        for n in names_i:
            np.testing.assert_array_equal(env.state(n), so[n], err_msg=f"step {k} {n}")
        for n in names_f:
            np.testing.assert_allclose(env.state(n), so[n], rtol=1e-12, atol=0, err_msg=f"step {k} {n}")
        np.testing.assert_array_equal(env.read_output("terminated"), ora.terminated, err_msg=f"step {k}")
        np.testing.assert_array_equal(env.read_output("truncated"), ora.truncated, err_msg=f"step {k}")
        # device log() vs libm log(): <= 1 ulp of f64 each
        np.testing.assert_allclose(env.read_output("reward64"), ora.reward64, rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(env.read_output("reward"), ora.reward, rtol=1e-6, atol=1e-12)
        np.testing.assert_array_equal(env.read_output("obs"), ora.obs, err_msg=f"step {k} obs")
        if k > 0:
            np.testing.assert_array_equal(env.terminal_ids(), np.sort(ora.term_ids))
            n_term += len(ora.term_ids)
    env.close()
    ora.close()
    return n_term


@pytest.mark.parametrize("autoreset", ["next_step", "same_step", None])
def test_hip_vs_oracle_c3_shape_random_resets(oracle_mod, autoreset):
    ds = [_synthetic(11, 700, 30, sigma=2e-2)]
    n = _compare_with_oracle(oracle_mod, ds, n_envs=3000, steps=150, seed=5, windows=20,
                             positions=[-1, 0, 1], trading_fees=1e-4, borrow_interest_rate=3e-6,
                             max_episode_duration=40, autoreset=autoreset)
    assert n > 3000  # every env ended at least once


def test_hip_vs_oracle_c2_shape(oracle_mod):
    ds = [_synthetic(12, 500, 14)]
    _compare_with_oracle(oracle_mod, ds, n_envs=4096, steps=120, seed=6, positions=[-1, 0, 1],
                         trading_fees=1e-4, borrow_interest_rate=3e-6, max_episode_duration=50,
                         autoreset="next_step")


@pytest.mark.parametrize("n_static,windows", [(3, 7), (1, None), (17, 3), (0, 5), (30, 20)])
def test_hip_vs_oracle_odd_shapes(oracle_mod, n_static, windows):
    """F_obs not a multiple of 4 takes the 4-byte path; ragged last wave (N=1001)."""
    ds = [_synthetic(13, 300, n_static, sigma=3e-2, drift=-2e-3)]
    _compare_with_oracle(oracle_mod, ds, n_envs=1001, steps=100, seed=7, windows=windows,
                         positions=[-2, -1, 0, 0.5, 1, 2], trading_fees=1e-3,
                         borrow_interest_rate=1e-4, autoreset="next_step")


@pytest.mark.parametrize("switch,persist", [(1, False), (2, True)])
def test_hip_vs_oracle_multidataset(oracle_mod, switch, persist):
    ds = [_synthetic(100 + d, 150 + 13 * d, 6) for d in range(9)]
    _compare_with_oracle(oracle_mod, ds, n_envs=777, steps=160, seed=8, windows=4,
                         positions=[-1, 0, 1], trading_fees=1e-4, max_episode_duration=25,
                         episodes_between_dataset_switch=switch, dyn_persist=persist,
                         autoreset="next_step")


def test_hip_vs_oracle_persist_single_dataset(oracle_mod):
    ds = [_synthetic(21, 200, 2)]
    _compare_with_oracle(oracle_mod, ds, n_envs=300, steps=200, seed=9, windows=6,
                         positions=[-1, 0, 1], max_episode_duration=20, dyn_persist=True,
                         autoreset="same_step")


def test_hip_masked_reset_and_disabled_autoreset(oracle_mod):
    """Caller-driven resets with a mask; other envs must be untouched."""
    from gym_trading_env_amd.batched import BatchedTradingEnv
    f, c = _synthetic(31, 120, 4)
    kw = dict(positions=[0, 1], windows=3, max_episode_duration=30, autoreset=None, seed=3)
    env = BatchedTradingEnv((f, c), num_envs=500, output="numpy", **kw)
    full = np.zeros((120, 6), np.float32); full[:, :4] = f
    ora = oracle_mod.OracleEnv(env.cfg, [(full, c)])
    env.reset(); ora.reset()
    rng = np.random.default_rng(0)
    for k in range(80):
        a = rng.integers(0, 2, 500).astype(np.int32)
        env.step(a); ora.step(a)
        if k % 10 == 9:
            mask = (ora.truncated | ora.terminated).astype(np.uint8)
            if mask.any():
                env.reset(mask=mask); ora.reset(mask=mask)
        np.testing.assert_array_equal(env.read_output("obs"), ora.obs)
        np.testing.assert_array_equal(env.state("idx"), ora.state()["idx"])
        np.testing.assert_array_equal(env.state("episode"), ora.state()["episode"])
    env.close()


def test_hip_vs_oracle_limit_orders(oracle_mod):
    """Persistent and non-persistent limit orders on random envs, several datasets."""
    ds = []
    for d in range(3):
        f, c = _synthetic(400 + d, 220 + 10 * d, 5, sigma=1e-2)
        r = np.random.default_rng(500 + d)
        ds.append((f, c, c * (1 + np.abs(r.normal(0, 8e-3, len(c)))), c * (1 - np.abs(r.normal(0, 8e-3, len(c))))))
    _compare_with_oracle(oracle_mod, ds, n_envs=1500, steps=150, seed=21, windows=3, p_order=0.2,
                         positions=[-1, -0.5, 0, 1, 2], trading_fees=1e-3, borrow_interest_rate=1e-4,
                         max_episode_duration=60, autoreset="next_step")


def test_limit_orders_need_high_low():
    from gym_trading_env_amd import GteError
    from gym_trading_env_amd.batched import BatchedTradingEnv
    f, c = _synthetic(1, 100, 2)
    env = BatchedTradingEnv((f, c), num_envs=8, output="numpy")
    env.reset()
    with pytest.raises(GteError, match="high/low"):
        env.add_limit_order(np.zeros(8, np.int32), c[:8], True)
    env.close()


def test_affinity_order_does_not_change_results(oracle_mod):
    """The L2-affinity processing order is rebuilt every step here (period 1); outputs must
    equal the identity order and the oracle."""
    ds = [_synthetic(61, 5000, 30, sigma=5e-3)]
    kw = dict(windows=20, positions=[-1, 0, 1], trading_fees=1e-4, borrow_interest_rate=3e-6,
              max_episode_duration=30, autoreset="next_step")
    _compare_with_oracle(oracle_mod, ds, n_envs=8192, steps=70, seed=31, check_every=7,
                         affinity_period=1, **kw)
    _compare_with_oracle(oracle_mod, ds, n_envs=8192, steps=70, seed=31, check_every=7,
                         affinity_period=-1, **kw)


def test_out_of_range_device_action_is_a_hold():
    import torch
    from gym_trading_env_amd.batched import BatchedTradingEnv
    f, c = _synthetic(62, 200, 2)
    env = BatchedTradingEnv((f, c), num_envs=256, positions=[0, 1], output="torch", seed=1)
    env.reset()
    before = env.state("position_index").copy()
    bad = torch.full((256,), 1000, dtype=torch.int32, device="cuda")
    env.step(bad)
    np.testing.assert_array_equal(env.state("position_index"), before)
    with pytest.raises(IndexError):
        env.step(np.full(256, 2, np.int32))       # host actions are range-checked like :234
    env.close()


def test_injection_is_validated():
    from gym_trading_env_amd import GteError
    from gym_trading_env_amd.batched import BatchedTradingEnv
    f, c = _synthetic(63, 100, 2)
    env = BatchedTradingEnv((f, c), num_envs=4, positions=[0, 1], windows=5,
                            max_episode_duration=20, output="numpy")
    with pytest.raises(GteError, match="start row"):
        env.reset(inject_idx=[10, 10, 99, 10])
    with pytest.raises(GteError, match="position index"):
        env.reset(inject_position_index=[0, 5, 0, 0])
    env.reset(inject_idx=[4, 50, 98, 10], inject_position_index=[0, 1, 0, 1])
    info_keys = env.step(np.zeros(4, np.int32))[4]
    assert info_keys["_portfolio_valuation"].all() and "idx" in info_keys
    env.close()


SHAPES = [
    # (n_envs, n_static, windows, n_dyn kinds, positions, T, max_dur)
    (1, 2, None, 2, [0, 1], 50, "max"),
    (1, 30, 20, 2, [-1, 0, 1], 300, 40),
    (63, 6, 1, 2, [-1, 0, 1], 80, 20),              # windows=1 is still a window: obs (1, F)
    (65, 2, 2, 2, [0, 1], 60, 10),
    (130, 8, 5, 0, [0, 0.5, 1], 90, 25),            # no dynamic features at all
    (200, 3, 3, 1, [-1, 1], 70, "max"),             # one dynamic feature, F_obs = 4
    (77, 510, 3, 2, [-1, 0, 1], 40, 12),            # wide rows: F_obs = 512
    (40, 2, 512, 2, [-1, 0, 1], 1300, 60),          # long windows: 512 x 4
    (300, 5, 64, 3, [-1, 0, 1], 400, 50),           # F_obs = 8, three dynamic features
    (999, 12, 7, 4, list(np.linspace(-1, 2, 32)), 120, 30),  # 4 dynamic features, 32 positions
    (5000, 29, 20, 2, [-1, 0, 1], 400, 35),         # F_obs = 31: 4-byte path at size
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: f"N{s[0]}_Fs{s[1]}_W{s[2]}_nd{s[3]}")
@pytest.mark.parametrize("autoreset", ["next_step", "same_step"])
def test_hip_vs_oracle_shape_sweep(oracle_mod, shape, autoreset):
    n_envs, n_static, windows, nd, positions, T, max_dur = shape
    kinds = ["last_position_taken", "real_position", "real_position", "last_position_taken"][:nd]
    ds = [_synthetic(900 + n_static, T, n_static, sigma=2e-2, drift=-1e-3)]
    _compare_with_oracle(oracle_mod, ds, n_envs=n_envs, steps=45, seed=17, check_every=3,
                         windows=windows, positions=positions, dynamic_feature_functions=kinds,
                         trading_fees=1e-3, borrow_interest_rate=1e-4,
                         max_episode_duration=max_dur, autoreset=autoreset)


@pytest.mark.parametrize("variant,store", [(0, 1), (0, 2), (0, 0), (1, 2), (2, 2), (3, 1), (64, 2),
                                           (64, 1), (8192, 2), (8192 + 1, 2), (4096, 2)])
def test_hip_vs_oracle_kernel_variants(oracle_mod, variant, store):
    """Every selectable kernel structure (isolated / shared-TU hot kernel, per-wave phase A, no
    LDS staging, record stored by the stepping lane instead of through LDS, generic copy loop) and
    store policy (plain / nt / sc1) gives the same results."""
    ds = [_synthetic(71, 3000, 30, sigma=1e-2)]
    _compare_with_oracle(oracle_mod, ds, n_envs=6000, steps=60, seed=41, check_every=6,
                         windows=20, positions=[-1, 0, 1], trading_fees=1e-4,
                         borrow_interest_rate=3e-6, max_episode_duration=25,
                         autoreset="next_step", kernel_variant=variant, nontemporal_obs=store)


def test_persistent_dynamic_columns_beyond_row_32768(oracle_mod):
    """dyn_persist keeps a T-deep dynamic column per env; the window's first ROW used to travel
    in 15 bits of the LDS job record, so rows >= 32 768 aliased (the reference's own
    BTC_USD-Hourly.csv has 33 259 rows).  T = 40 000, starts forced past row 32 768 + W."""
    from gym_trading_env_amd.batched import BatchedTradingEnv
    T, Fs, N, W = 40_000, 6, 384, 20
    f, c = _synthetic(5, T, Fs, sigma=4e-3)
    kw = dict(windows=W, positions=[-1, 0, 1], trading_fees=1e-4, borrow_interest_rate=3e-6,
              max_episode_duration=12, autoreset="next_step", dyn_persist=True, seed=3)
    env = BatchedTradingEnv((f, c), num_envs=N, output="numpy", **kw)
    full = np.zeros((T, Fs + 2), np.float32); full[:, :Fs] = f
    ora = oracle_mod.OracleEnv(env.cfg, [(full, c)])
    rng = np.random.default_rng(8)
    # a third of the envs below the old limit, the rest across and beyond it; the later
    # auto-resets land close to the first start so that stale rows are revisited
    start = np.where(np.arange(N) % 3 == 0, rng.integers(W - 1, 30_000, N),
                     rng.integers(32_768 - W, 39_000, N)).astype(np.int32)
    q = np.clip(start[:, None] + rng.integers(-6, 7, (N, 6)), W - 1, T - 40).astype(np.int32)
    env.set_autoreset_injection(q, None, None)
    ora.set_autoreset_injection(q, None, None)
    env.reset(inject_idx=start)
    ora.reset(inj_idx=start)
    for k in range(60):
        a = rng.integers(-1, 3, N).astype(np.int32)
        env.step(a); ora.step(a)
        np.testing.assert_array_equal(env.state("idx"), ora.state()["idx"])
        np.testing.assert_array_equal(env.read_output("obs"), ora.obs, err_msg=f"step {k}")
    assert (env.state("idx") > 32_768 + W).sum() > N // 2
    env.close()


def test_window_limit_and_dataset_replacement():
    """gte_create refuses windows >= 32 768 (the job record's zero-row count has 15 bits);
    gte_upload_dataset after a reset refuses a SHORTER table (running envs could be past its
    end) and swaps a same-length one in without disturbing the env."""
    from gym_trading_env_amd import _abi
    from gym_trading_env_amd.batched import BatchedTradingEnv
    from gym_trading_env_amd import staging
    f, c = _synthetic(6, 70_000, 1)
    with pytest.raises(_abi.GteError, match="32768"):
        BatchedTradingEnv((f, c), num_envs=4, windows=32_768, dynamic_feature_functions=[],
                          output="numpy")
    f, c = _synthetic(7, 500, 3)
    env = BatchedTradingEnv((f, c), num_envs=64, windows=4, output="numpy", autoreset="next_step",
                            max_episode_duration=30)
    env.reset()
    for _ in range(5):
        env.step(np.zeros(64, np.int32))
    with pytest.raises(_abi.GteError, match="cannot replace"):
        env.upload_dataset(0, staging.stage_arrays(f[:200], c[:200]))
    f2 = f + 1.0
    env.upload_dataset(0, staging.stage_arrays(f2, c))
    obs, *_ = env.step(np.zeros(64, np.int32))
    idx = env.state("idx")
    np.testing.assert_array_equal(obs[:, -1, :3], f2[idx])
    env.close()


@pytest.mark.parametrize("windows,persist,output", [(20, False, "numpy"), (None, False, "torch"),
                                                    (5, True, "numpy"), (3, False, "torch")])
def test_same_step_final_observation(oracle_mod, windows, persist, output):
    """Same-step auto-reset keeps the terminal observation (Gymnasium final_observation /
    SB3 terminal_observation) of every env that ends."""
    from gym_trading_env_amd.batched import BatchedTradingEnv
    f, c = _synthetic(81, 400, 6, sigma=2e-2, drift=-2e-3)
    kw = dict(windows=windows, positions=[-2, -1, 0, 1, 2], trading_fees=1e-3,
              borrow_interest_rate=1e-4, max_episode_duration=14, autoreset="same_step",
              final_obs=True, dyn_persist=persist, seed=9)
    N = 700
    env = BatchedTradingEnv((f, c), num_envs=N, output=output, **kw)
    full = np.zeros((400, 8), np.float32); full[:, :6] = f
    ora = oracle_mod.OracleEnv(env.cfg, [(full, c)])
    env.reset(); ora.reset()
    rng = np.random.default_rng(2)
    seen = 0
    for k in range(60):
        a = rng.integers(-1, 5, N).astype(np.int32)
        env.step(a); ora.step(a)
        ids, fin = env.final_observations()
        if output == "torch":
            fin = fin.cpu().numpy()
        np.testing.assert_array_equal(ids, np.sort(ora.term_ids))
        np.testing.assert_array_equal(fin, ora.final_obs[ids], err_msg=f"step {k}")
        obs = env.read_output("obs") if output == "numpy" else env._t["obs"].cpu().numpy()
        np.testing.assert_array_equal(obs, ora.obs, err_msg=f"step {k}")
        seen += len(ids)
    assert seen > N
    env.close()
    with pytest.raises(ValueError, match="same_step"):
        BatchedTradingEnv((f, c), num_envs=4, autoreset="next_step", final_obs=True, output="numpy")


@pytest.mark.parametrize("seed", range(24))
def test_hip_vs_oracle_random_configurations(oracle_mod, seed):
    """Random position sets, fees, interest rates, initial values, window lengths, feature
    counts, durations, auto-reset modes and price volatility (calm to violent: the 0.7 drawdown
    rule fires), several hundred envs each, against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    T = int(rng.integers(80, 500))
    n_static = int(rng.integers(1, 40))
    sigma = float(rng.choice([1e-3, 1e-2, 5e-2, 0.12]))
    ds = [_synthetic(3000 + seed, T, n_static, sigma=sigma, drift=-sigma / 4)]
    P = int(rng.integers(2, 9))
    positions = sorted(set(np.round(rng.uniform(-2.5, 3.5, P), 2).tolist() + [0.0]))
    windows = None if rng.random() < 0.25 else int(rng.integers(1, 30))
    first = 0 if windows is None else windows - 1
    room = T - 2 * first
    if room < 12:
        windows, first, room = 3, 2, T - 4
    max_dur = "max" if rng.random() < 0.3 else int(rng.integers(3, max(4, room // 2)))
    nd = int(rng.integers(0, 5))
    kinds = [str(rng.choice(["last_position_taken", "real_position"])) for _ in range(nd)]
    n = _compare_with_oracle(
        oracle_mod, ds, n_envs=int(rng.integers(1, 700)), steps=40, seed=seed, check_every=4,
        positions=positions, windows=windows, dynamic_feature_functions=kinds,
        trading_fees=float(rng.choice([0.0, 1e-4, 1e-3, 1e-2])),
        borrow_interest_rate=float(rng.choice([0.0, 3e-6, 1e-4, 1e-3])),
        portfolio_initial_value=float(rng.choice([1000.0, 1.0, 1e6])),
        initial_position="random" if rng.random() < 0.7 else positions[int(rng.integers(len(positions)))],
        max_episode_duration=max_dur,
        autoreset=[None, "next_step", "same_step"][int(rng.integers(3))],
        dyn_persist=bool(rng.random() < 0.2))
    assert n >= 0
