"""BASELINE.json's full sizes on the GPU: a few steps against the OpenMP oracle on the
same seeds (device Philox resets, no injection) and size-independent invariants over a
longer run."""
import numpy as np
import pytest

from test_gpu_parity import _compare_with_oracle, _synthetic

pytestmark = pytest.mark.gpu

C3 = dict(windows=20, positions=[-1, 0, 1], trading_fees=1e-4, borrow_interest_rate=3e-6,
          autoreset="next_step")


def test_config3_full_size_vs_oracle(oracle_mod):
    """65 536 envs x obs (20, 32), T = 100 000; short episodes so that resets happen."""
    ds = [_synthetic(1234, 100_000, 30, sigma=1e-3)]
    n = _compare_with_oracle(oracle_mod, ds, n_envs=65_536, steps=24, seed=11, check_every=8,
                             max_episode_duration=9, **C3)
    assert n > 0


def test_config2_full_size_vs_oracle(oracle_mod):
    """4 096 envs, 16-feature obs, no window, T = 100 000."""
    ds = [_synthetic(1234, 100_000, 14, sigma=1e-3)]
    _compare_with_oracle(oracle_mod, ds, n_envs=4_096, steps=60, seed=12, check_every=10,
                         positions=[-1, 0, 1], trading_fees=1e-4, borrow_interest_rate=3e-6,
                         max_episode_duration=25, autoreset="next_step")


def test_config5_shape_many_datasets_vs_oracle(oracle_mod):
    """1 024 resident datasets (config 5's count; T shortened to keep host memory small),
    32 768 envs = one GPU's share of 262 144, per-env dataset indirection."""
    ds = [_synthetic(2000 + d, 1200 + (d % 7) * 10, 30, sigma=2e-3) for d in range(1024)]
    _compare_with_oracle(oracle_mod, ds, n_envs=32_768, steps=40, seed=13, check_every=10,
                         max_episode_duration=12, **C3)


def test_config3_invariants_long_run():
    """600 steps at full size: properties that hold whatever the size."""
    from gym_trading_env_amd.batched import BatchedTradingEnv
    import torch
    feat, close = _synthetic(1234, 100_000, 30, sigma=1e-3)
    N = 65_536
    env = BatchedTradingEnv((feat, close), num_envs=N, max_episode_duration=500, seed=3,
                            output="torch", **C3)
    obs, _ = env.reset()
    dev = obs.device
    table = torch.from_numpy(feat).to(dev)
    acts = torch.randint(-1, 3, (64, N), dtype=torch.int32, device=dev)
    ended_total = 0
    for k in range(600):
        obs, reward, term, trunc, info = env.step(acts[k % 64])
        ended_total += int((term | trunc).sum())
        if k % 100 == 99 or k < 3:
            idx = torch.from_numpy(env.state("idx")).to(dev).long()
            # static columns of the newest window row == the table row at idx (all envs)
            assert torch.equal(obs[:, -1, :30], table[idx])
            # ... and of the oldest window row == table row idx-19
            assert torch.equal(obs[:, 0, :30], table[idx - 19])
            # dynamic column 0 of the newest row == the position taken
            pos = torch.tensor([-1.0, 0.0, 1.0], device=dev)[
                torch.from_numpy(env.state("position_index")).to(dev).long()]
            assert torch.equal(obs[:, -1, 30], pos)
            st = {n: env.state(n) for n in ("idx", "step", "start_idx", "needs_reset", "episode")}
            np.testing.assert_array_equal(st["idx"] - st["start_idx"], st["step"])
            flags = (term | trunc).cpu().numpy()
            np.testing.assert_array_equal(st["needs_reset"].astype(bool), flags)
            np.testing.assert_array_equal(env.terminal_ids(), np.nonzero(flags)[0])
            pv = env.state("portfolio_valuation")
            assert np.isfinite(pv).all() and (pv > 0).all()
            r = reward.cpu().numpy()
            assert np.isfinite(r).all() and np.abs(r).max() < 0.1
    # random starts in [19, 99481): every env hits the 500-step truncation once in 600 steps
    assert ended_total >= N
    assert (env.state("episode") >= 2).all()
    env.close()


def test_config4_total_size_on_one_gpu_vs_oracle(oracle_mod):
    """262 144 envs (config 4's node total) on one GPU: 671 MB of observations per step;
    checks 64-bit offsets end to end."""
    ds = [_synthetic(1234, 100_000, 30, sigma=1e-3)]
    _compare_with_oracle(oracle_mod, ds, n_envs=262_144, steps=12, seed=14, check_every=6,
                         max_episode_duration=7, **C3)


def test_config3_full_size_rollout_paths_agree():
    """65 536 envs x obs (20, 32): a rollout that keeps every observation (168 MB per step,
    fused kernel, streaming stores), one that keeps only the last, one forced through separate
    launches (kernel_variant 128) and plain single steps must agree bit for bit, episodes
    ending and restarting on the way."""
    import torch
    from gym_trading_env_amd.batched import BatchedTradingEnv
    ds = _synthetic(1234, 100_000, 30, sigma=1e-3)
    N, K = 65_536, 12
    kw = dict(num_envs=N, seed=21, max_episode_duration=7, **C3)
    envs = [BatchedTradingEnv(ds, **kw), BatchedTradingEnv(ds, **kw),
            BatchedTradingEnv(ds, kernel_variant=128, **kw), BatchedTradingEnv(ds, **kw)]
    for e in envs:
        e.reset()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(3)
    acts = torch.randint(-1, 3, (K, N), dtype=torch.int32, device="cuda", generator=gen)
    full = envs[0].rollout(acts, keep_obs=True)
    lean = envs[1].rollout(acts, keep_obs=False, valuation=True)
    forced = envs[2].rollout(acts, keep_obs=True)
    for k in range(K):
        obs, reward, term, trunc, _ = envs[3].step(acts[k])
        for out in (full, lean, forced):
            assert torch.equal(out["reward"][k], reward)
            assert torch.equal(out["terminated"][k], term) and torch.equal(out["truncated"][k], trunc)
        assert torch.equal(full["obs"][k], obs) and torch.equal(forced["obs"][k], obs)
    assert torch.equal(lean["obs"], obs)
    assert int(full["truncated"].sum()) >= N  # every env finished an episode and restarted
    for f in ("idx", "step", "episode", "asset", "fiat", "portfolio_valuation"):
        ref = envs[3].state(f)
        for e in envs[:3]:
            np.testing.assert_array_equal(e.state(f), ref, err_msg=f)
    np.testing.assert_array_equal(lean["valuation"][-1].cpu().numpy(), envs[3].state("portfolio_valuation"))
    for e in envs:
        e.close()


@pytest.mark.parametrize("n_envs,stores", [(65_536, "sc1"), (81_920, "non-temporal"), (32_768, "sc1")])
def test_automatic_geometry_and_store_policy(oracle_mod, n_envs, stores):
    """The launch geometry makes every workgroup resident at once and fills the slots (the
    hardware packs workgroups onto CUs, so empty slots mean idle CUs), the store policy follows
    the observation buffer size; results stay those of the oracle."""
    import torch
    from gym_trading_env_amd.batched import BatchedTradingEnv
    ds = _synthetic(1234, 20_000, 30, sigma=1e-3)
    env = BatchedTradingEnv(ds, num_envs=n_envs, seed=5, max_episode_duration=9, **C3)
    info = env.launch_info()
    env.close()
    slots = info["resident_workgroups_per_cu"] * torch.cuda.get_device_properties(0).multi_processor_count
    assert info["obs_stores"] == stores, info
    assert info["resident_workgroups_per_cu"] >= 4 and 0.85 * slots <= info["n_blocks"] <= slots, info
    _compare_with_oracle(oracle_mod, [ds], n_envs=n_envs, steps=12, seed=5, check_every=6,
                         max_episode_duration=9, **C3)
