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
    """The launch geometry makes every workgroup resident at once, with at least 16 waves on every
    CU and the envs dealt evenly over the CUs (the launch ends when the busiest CU is done), the
    store policy follows the observation buffer size; results stay those of the oracle."""
    import torch
    from gym_trading_env_amd.batched import BatchedTradingEnv
    ds = _synthetic(1234, 20_000, 30, sigma=1e-3)
    env = BatchedTradingEnv(ds, num_envs=n_envs, seed=5, max_episode_duration=9, **C3)
    info = env.launch_info()
    env.close()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    slots = info["resident_workgroups_per_cu"] * cus
    assert info["obs_stores"] == stores, info
    assert info["resident_workgroups_per_cu"] >= 4 and 4 * cus <= info["n_blocks"] <= slots, info
    busiest = -(-info["n_blocks"] // cus) * info["envs_per_wave"] * (info["threads_per_block"] // 64)
    assert busiest <= 1.05 * n_envs / cus, (busiest, info)
    _compare_with_oracle(oracle_mod, [ds], n_envs=n_envs, steps=12, seed=5, check_every=6,
                         max_episode_duration=9, **C3)


def test_config5_full_T_invariants():
    """BASELINE config 5 at its real per-GPU size — 128 resident datasets x T = 100 000 rows
    (1.6 GB of tables), 32 768 envs, dataset switch at every episode — where no oracle run is
    affordable: size-independent properties of `MultiDatasetTradingEnv` semantics
    (environments.py:365-400) and of `_get_obs` (:152-160).
      * every window row's static columns are the rows idx-W+1..idx of the env's OWN dataset;
      * idx - start_idx == step for running envs;
      * dataset picks are "uniform among the least used" (:383-388) == every dataset exactly
        once per round of D picks: over 2 rounds of short episodes each env's picks 1..D-1 are
        distinct and picks D..2D-1 are a permutation of all D datasets (pick 0, the constructor's,
        is consumed by no episode when the switch period is 1 — the reference's counter quirk)."""
    import torch
    from gym_trading_env_amd.batched import BatchedTradingEnv
    D, T, Fs, N, W = 128, 100_000, 30, 32_768, 20
    rng = np.random.default_rng(77)
    data = []
    for d in range(D):  # cheap but distinct tables: dataset d, row t, column c are recoverable
        base = rng.normal(0, 1, (T, 1)).astype(np.float32)
        feat = base + np.arange(Fs, dtype=np.float32)[None, :] * 0.25 + np.float32(d)
        close = 100.0 * np.exp(np.cumsum(rng.normal(0, 1e-3, T)))
        data.append((feat, close))
    env = BatchedTradingEnv(data, num_envs=N, max_episode_duration=8, seed=21, output="torch",
                            episodes_between_dataset_switch=1, **C3)
    obs, _ = env.reset()
    dev = obs.device
    sample = torch.from_numpy(np.sort(rng.choice(N, 1536, replace=False))).to(dev)
    sample_h = sample.cpu().numpy()
    acts = torch.randint(-1, 3, (16, N), dtype=torch.int32, device=dev)
    picks = [[] for _ in sample_h]            # datasets per episode of the sampled envs
    last_episode = np.zeros(len(sample_h), np.int64)
    steps = 8 * (2 * D + 2)
    for k in range(steps):
        obs, reward, term, trunc, info = env.step(acts[k % 16])
        ep = env.state("episode")[sample_h]
        ds = env.state("dataset_index")[sample_h]
        for j in np.nonzero(ep != last_episode)[0]:
            picks[j].append(int(ds[j]))
        last_episode = ep
        if k % 257 == 0 or k == steps - 1:
            idx, start, step = (env.state(n) for n in ("idx", "start_idx", "step"))
            need = env.state("needs_reset")
            run = need == 0
            np.testing.assert_array_equal((idx - start)[run], step[run])
            o = obs[sample].cpu().numpy()
            for j in range(0, len(sample_h), 7):
                e = sample_h[j]
                rows = data[ds[j]][0][idx[e] - W + 1: idx[e] + 1]
                np.testing.assert_array_equal(o[j, :, :Fs], rows, err_msg=f"step {k} env {e}")
    n_ok = 0
    for p in picks:
        if len(p) < 2 * D:
            continue
        first, second = p[:D - 1], p[D - 1:2 * D - 1]   # picks 1..D-1, then picks D..2D-1
        assert len(set(first)) == D - 1, "a dataset was visited twice within its first round"
        assert sorted(second) == list(range(D)), "the second round is not a permutation"
        n_ok += 1
    assert n_ok > len(picks) * 0.9
    env.close()


def test_bound_outputs_do_not_pay_for_a_second_observation_buffer():
    """The torch path binds its own tensors before the first reset: the library then never
    allocates its own [N, W, F_obs] buffer (it did in round 1: 168 MB at the headline shape, 671 MB
    at 262 144 envs).  Measured as the device memory the env takes."""
    import torch
    from gym_trading_env_amd.batched import BatchedTradingEnv
    ds = _synthetic(77, 20_000, 30, sigma=1e-3)
    n_envs = 65_536
    obs_bytes = n_envs * 20 * 32 * 4
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info()
    env = BatchedTradingEnv(ds, num_envs=n_envs, seed=5, max_episode_duration=50, **C3)
    env.reset()
    env.step(torch.zeros(n_envs, dtype=torch.int32, device="cuda"))
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    env.close()
    taken = free0 - free1
    assert obs_bytes < taken < 1.7 * obs_bytes, (taken / 1e6, obs_bytes / 1e6)
