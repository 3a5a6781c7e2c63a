"""gte_rollout (K steps fused into one launch, csrc/gte_rollout.hip: windows resident in LDS,
one new table row per env and step) against K single
gte_step calls on a twin env, bit for bit: per-step rewards / flags / observations /
valuations, the state afterwards, the terminal list, and the steps that follow (the
dynamic-feature rings the fused kernel kept in LDS must have reached HBM unchanged)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STATE = ("idx", "step", "position_index", "dataset_index", "start_idx", "episode", "needs_reset",
         "asset", "fiat", "interest_asset", "interest_fiat", "portfolio_valuation", "real_position")


def _data(seed, T, Fs, sigma=1e-2):
    rng = np.random.default_rng(seed)
    close = 100 * np.exp(np.cumsum(rng.normal(0, sigma, T)))
    feat = rng.normal(0, 1, (T, Fs)).astype(np.float32)
    return feat, close, close * 1.004, close * 0.996


def _twins(data, N, **kw):
    from gym_trading_env_amd.batched import BatchedTradingEnv
    a = BatchedTradingEnv(data, num_envs=N, **kw)
    b = BatchedTradingEnv(data, num_envs=N, **kw)
    a.reset()
    b.reset()
    return a, b


def _same_state(a, b, tag):
    for f in STATE:
        np.testing.assert_array_equal(a.state(f), b.state(f), err_msg=f"{tag}: {f}")


def _check(a, b, acts, keep_obs, tag):
    """env a: single steps; env b: one rollout over the same actions."""
    import torch
    K = acts.shape[0]
    out = b.rollout(acts, keep_obs=keep_obs, valuation=True, reward64=True)
    for k in range(K):
        obs, reward, term, trunc, _ = a.step(acts[k])
        np.testing.assert_array_equal(out["reward"][k].cpu().numpy(), reward.cpu().numpy(),
                                      err_msg=f"{tag} step {k} reward")
        np.testing.assert_array_equal(out["reward64"][k].cpu().numpy(), a.read_output("reward64"))
        np.testing.assert_array_equal(out["terminated"][k].cpu().numpy(), term.cpu().numpy())
        np.testing.assert_array_equal(out["truncated"][k].cpu().numpy(), trunc.cpu().numpy())
        np.testing.assert_array_equal(out["valuation"][k].cpu().numpy(),
                                      a.state("portfolio_valuation"), err_msg=f"{tag} step {k} pv")
        if keep_obs:
            np.testing.assert_array_equal(out["obs"][k].cpu().numpy(), obs.cpu().numpy(),
                                          err_msg=f"{tag} step {k} obs")
    if not keep_obs:
        np.testing.assert_array_equal(out["obs"].cpu().numpy(), obs.cpu().numpy(),
                                      err_msg=f"{tag} last obs")
    # the env's own return buffers and terminal list describe the last step
    np.testing.assert_array_equal(b.read_output("reward"), a.read_output("reward"))
    np.testing.assert_array_equal(b.read_output("terminated"), a.read_output("terminated"))
    np.testing.assert_array_equal(b.read_output("truncated"), a.read_output("truncated"))
    np.testing.assert_array_equal(b.terminal_ids(), a.terminal_ids())
    _same_state(a, b, tag)
    return int(out["terminated"].sum() + out["truncated"].sum())


CASES = {
    # headline layout: 16-byte vectors, cooperative phase A, rings staged in LDS -> fused kernel
    "window20": dict(data=lambda: _data(1, 700, 30)[:2], N=1000,
                     kw=dict(positions=[-1, 0, 1], windows=20, trading_fees=1e-4,
                             borrow_interest_rate=3e-6, max_episode_duration=40)),
    "nowindow": dict(data=lambda: _data(2, 500, 14)[:2], N=777,
                     kw=dict(positions=[-1, 0, 1], windows=None, trading_fees=1e-4,
                             borrow_interest_rate=3e-6, max_episode_duration=25)),
    "drawdown_same_step": dict(data=lambda: _data(3, 400, 6, sigma=4e-2)[:2], N=512,
                               kw=dict(positions=[-2, -1, 0, 1, 2, 3], windows=4, trading_fees=1e-3,
                                       borrow_interest_rate=1e-3, autoreset="same_step")),
    "multidataset": dict(data=lambda: [_data(10 + d, 300 + 20 * d, 6)[:2] for d in range(5)], N=640,
                         kw=dict(positions=[-1, 0, 1], windows=5, trading_fees=1e-4,
                                 max_episode_duration=20, episodes_between_dataset_switch=2)),
    "no_autoreset": dict(data=lambda: _data(4, 300, 2)[:2], N=300,
                         kw=dict(positions=[-1, 0, 1], windows=3, trading_fees=1e-3,
                                 max_episode_duration=30, autoreset=None)),
    # edge shapes of the window-resident kernel: a one-row LDS ring, wide rows (few envs per
    # workgroup, two newest-row vectors per owner thread), a ragged last workgroup
    "window2": dict(data=lambda: _data(12, 400, 6)[:2], N=333,
                    kw=dict(positions=[-1, 0, 1], windows=2, trading_fees=1e-4, max_episode_duration=12)),
    "wide_rows": dict(data=lambda: _data(13, 300, 126)[:2], N=210,
                      kw=dict(positions=[-1, 0, 1], windows=6, trading_fees=1e-4,
                              borrow_interest_rate=3e-6, max_episode_duration=16)),
    # the gather-per-step fused kernel (what shapes too big for LDS residency run)
    "streaming_kernel": dict(data=lambda: _data(1, 700, 30)[:2], N=1000,
                             kw=dict(positions=[-1, 0, 1], windows=20, trading_fees=1e-4,
                                     borrow_interest_rate=3e-6, max_episode_duration=40,
                                     kernel_variant=256)),
    # shapes the fused kernels do not cover: K launches of the step kernel, same results
    "persist_fallback": dict(data=lambda: _data(5, 200, 2)[:2], N=256,
                             kw=dict(positions=[-1, 0, 1], windows=6, trading_fees=1e-4,
                                     max_episode_duration=25, dyn_persist=True)),
    "scalar_layout_fallback": dict(data=lambda: _data(6, 300, 3)[:2], N=300,
                                   kw=dict(positions=[0, 1], windows=7, max_episode_duration=30)),
    "forced_fallback": dict(data=lambda: _data(7, 400, 6)[:2], N=400,
                            kw=dict(positions=[-1, 0, 1], windows=5, max_episode_duration=30,
                                    kernel_variant=128)),
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("keep_obs", [True, False])
def test_rollout_equals_single_steps(name, keep_obs):
    import torch
    c = CASES[name]
    N, kw = c["N"], dict(c["kw"], seed=11)
    a, b = _twins(c["data"](), N, **kw)
    P = len(kw["positions"])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    ended = 0
    for K in (1, 7, 33):  # rollouts follow one another, single steps in between
        acts = torch.randint(-1, P, (K, N), dtype=torch.int32, device="cuda", generator=gen)
        ended += _check(a, b, acts, keep_obs, f"{name} K={K}")
        for _ in range(3):
            one = torch.randint(-1, P, (N,), dtype=torch.int32, device="cuda", generator=gen)
            oa = a.step(one)
            ob = b.step(one)
            for x, y in zip(oa[:4], ob[:4]):
                np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy())
        _same_state(a, b, f"{name} after singles")
    if name != "no_autoreset":
        assert ended > N // 4  # episodes really ended and restarted inside the rollouts
    a.close()
    b.close()


def test_rollout_with_limit_orders_and_oracle(oracle_mod):
    """Pending limit orders fill inside a rollout exactly as in single steps, and the whole
    thing equals the oracle."""
    import torch
    feat, close, high, low = _data(8, 400, 6, sigma=1.5e-2)
    N, K = 384, 40
    kw = dict(positions=[-1, 0, 1], windows=4, trading_fees=1e-3, borrow_interest_rate=1e-4,
              max_episode_duration=60, seed=3)
    from gym_trading_env_amd.batched import BatchedTradingEnv
    env = BatchedTradingEnv((feat, close, high, low), num_envs=N, **kw)
    full = np.zeros((400, 8), np.float32)
    full[:, :6] = feat
    ora = oracle_mod.OracleEnv(env.cfg, [(full, close, high, low)])
    env.reset()
    ora.reset()
    rng = np.random.default_rng(0)
    idx = env.state("idx")
    pos = rng.integers(0, 3, N).astype(np.int32)
    limit = close[idx] * rng.uniform(0.99, 1.01, N)
    env.add_limit_order(pos, limit, np.ones(N, np.uint8))
    ora.add_limit_orders(pos, limit, np.ones(N, np.uint8))
    acts = rng.integers(-1, 3, (K, N)).astype(np.int32)
    out = env.rollout(torch.from_numpy(acts).cuda(), keep_obs=True, valuation=True)
    for k in range(K):
        ora.step(acts[k])
        np.testing.assert_array_equal(out["obs"][k].cpu().numpy(), ora.obs)
        np.testing.assert_array_equal(out["reward"][k].cpu().numpy(), ora.reward)
        np.testing.assert_array_equal(out["terminated"][k].cpu().numpy(), ora.terminated.astype(bool))
        np.testing.assert_array_equal(out["truncated"][k].cpu().numpy(), ora.truncated.astype(bool))
        np.testing.assert_allclose(out["valuation"][k].cpu().numpy(), ora.state()["portfolio_valuation"],
                                   rtol=1e-12)
    st = ora.state()
    np.testing.assert_array_equal(env.state("idx"), st["idx"])
    np.testing.assert_array_equal(env.state("position_index"), st["position_index"])
    env.close()


def test_rollout_argument_errors():
    import torch
    from gym_trading_env_amd.batched import BatchedTradingEnv
    feat, close = _data(9, 200, 2)[:2]
    env = BatchedTradingEnv((feat, close), num_envs=64, positions=[0, 1], windows=4)
    with pytest.raises(Exception):
        env.rollout(torch.zeros((3, 64), dtype=torch.int32, device="cuda"))  # before reset
    env.reset()
    with pytest.raises(ValueError):
        env.rollout(torch.zeros((3, 63), dtype=torch.int32, device="cuda"))
    with pytest.raises(IndexError):
        env.rollout(np.full((2, 64), 5))
    out = env.rollout([[None] * 64, [1] * 64])  # None = hold, like step()
    assert out["reward"].shape == (2, 64)
    env.close()


@pytest.mark.parametrize("seed", range(16))
def test_rollout_random_shapes_equal_single_steps(seed):
    """Random window lengths (one-row LDS rings to 40 rows), row widths (1 to 40 vectors), 1-4
    dynamic features, env counts that leave ragged workgroups, auto-reset modes, several datasets,
    violent prices (drawdown terminations): a fused rollout — with every step's observations and
    without — equals single steps bit for bit, and so do the steps that follow."""
    import torch
    rng = np.random.default_rng(7000 + seed)
    nd = int(rng.integers(1, 5))
    fv = int(rng.integers(1, 41))                 # vectors per row
    Fs = 4 * fv - nd
    if Fs < 0:
        fv += 1
        Fs = 4 * fv - nd
    W = int(rng.choice([2, 3, 5, 8, 13, 20, 40]))
    D = int(rng.choice([1, 1, 3]))
    T = int(rng.integers(3 * W + 40, 3 * W + 400))
    sigma = float(rng.choice([2e-3, 2e-2, 6e-2]))
    data = [_data(9000 + 10 * seed + d, T + 7 * d, Fs, sigma=sigma)[:2] for d in range(D)]
    kinds = [str(rng.choice(["last_position_taken", "real_position"])) for _ in range(nd)]
    autoreset = [None, "next_step", "same_step"][int(rng.integers(3))]
    N = int(rng.integers(1, 2500))
    kw = dict(positions=sorted(set(np.round(rng.uniform(-2, 3, 4), 1).tolist() + [0.0])), windows=W,
              dynamic_feature_functions=kinds, trading_fees=float(rng.choice([0, 1e-4, 1e-2])),
              borrow_interest_rate=float(rng.choice([0, 3e-6, 1e-3])),
              max_episode_duration=int(rng.integers(4, 30)), autoreset=autoreset,
              episodes_between_dataset_switch=int(rng.integers(1, 3)), seed=seed)
    a, b = _twins(data if D > 1 else data[0], N, **kw)
    P = len(kw["positions"])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    for K, keep in ((int(rng.integers(2, 30)), True), (int(rng.integers(2, 30)), False), (1, True)):
        acts = torch.randint(-1, P, (K, N), dtype=torch.int32, device="cuda", generator=gen)
        _check(a, b, acts, keep, f"seed {seed} W={W} Fobs={Fs + nd} nd={nd} N={N} K={K} {autoreset}")
        one = torch.randint(-1, P, (N,), dtype=torch.int32, device="cuda", generator=gen)
        for x, y in zip(a.step(one)[:4], b.step(one)[:4]):
            np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy())
    a.close()
    b.close()


def test_long_rollouts_equal_single_steps():
    """Hundreds of fused steps (many episodes per env, dataset switches, the LDS ring wrapping
    dozens of times, prices and state carried in registers all the way): the state afterwards,
    sampled rows on the way and the steps that follow equal single steps."""
    import torch
    data = [_data(300 + d, 900 + 13 * d, 14, sigma=1.5e-2)[:2] for d in range(3)]
    N = 1500
    kw = dict(positions=[-1, 0, 0.5, 1, 2], windows=10, trading_fees=1e-3, borrow_interest_rate=1e-4,
              max_episode_duration=37, seed=5)
    a, b = _twins(data, N, **kw)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)
    for K, keep in ((400, True), (700, False)):
        acts = torch.randint(-1, 5, (K, N), dtype=torch.int32, device="cuda", generator=gen)
        out = b.rollout(acts, keep_obs=keep, valuation=True)
        for k in range(K):
            obs, reward, term, trunc, _ = a.step(acts[k])
            if k % 53 == 0 or k == K - 1:
                assert torch.equal(out["reward"][k], reward), k
                assert torch.equal(out["terminated"][k], term) and torch.equal(out["truncated"][k], trunc)
                np.testing.assert_array_equal(out["valuation"][k].cpu().numpy(), a.state("portfolio_valuation"))
                if keep:
                    assert torch.equal(out["obs"][k], obs), k
        if not keep:
            assert torch.equal(out["obs"], obs)
        _same_state(a, b, f"after K={K}")
        del out
    one = torch.randint(-1, 5, (N,), dtype=torch.int32, device="cuda", generator=gen)
    for x, y in zip(a.step(one)[:4], b.step(one)[:4]):
        assert torch.equal(x, y)
    a.close()
    b.close()
