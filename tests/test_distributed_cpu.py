"""The N>1 path on CPU: world_size-2 `gloo` processes, each owning a contiguous env
shard (the oracle stands in for the HIP kernels as the data source — tests only), and
the return all-gather of gym_trading_env_amd.distributed.  The sharded run must equal
the unsharded one bit for bit (reset draws are keyed by global env id)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, STEPS = 96, 60
KW = dict(positions=[-1, 0, 1], windows=4, trading_fees=1e-4, borrow_interest_rate=3e-6,
          max_episode_duration=12, autoreset="next_step", seed=77)


def _data():
    rng = np.random.default_rng(3)
    close = 100 * np.exp(np.cumsum(rng.normal(0, 1e-2, 80)))
    feat = np.zeros((80, 5), np.float32)
    feat[:, :3] = rng.normal(0, 1, (80, 3))
    return feat, close


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gym_trading_env_amd.config import make_config
    from gym_trading_env_amd.distributed import ReturnGather, pack_returns, shard_range
    from oracle import oracle
    first, n = shard_range(G, world, rank)
    env = oracle.OracleEnv(make_config(n_envs=n, n_static=3, env_id_base=first, **KW), [_data()])
    env.reset()
    rg = ReturnGather(n, "cpu", obs_shape=(4, 5))
    actions = np.random.default_rng(5).integers(-1, 3, (STEPS, G)).astype(np.int32)
    rec = []
    for k in range(STEPS):
        env.step(actions[k, first:first + n])
        packed = pack_returns(torch.from_numpy(env.reward.copy()),
                              torch.from_numpy(env.terminated.copy()).bool(),
                              torch.from_numpy(env.truncated.copy()).bool())
        reward, term, trunc = rg.gather(packed)
        obs = rg.gather_obs(torch.from_numpy(env.obs.copy()))
        rec.append((reward.reshape(-1).numpy().copy(), term.reshape(-1).numpy().copy(),
                    trunc.reshape(-1).numpy().copy(), obs.numpy().copy()))
    if rank == 0:
        np.savez(os.path.join(out_dir, "gathered.npz"),
                 reward=np.stack([r[0] for r in rec]), term=np.stack([r[1] for r in rec]),
                 trunc=np.stack([r[2] for r in rec]), obs=np.stack([r[3] for r in rec]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_all_envs():
    from gym_trading_env_amd.distributed import shard_range
    for g, w in ((262144, 8), (10, 3), (7, 8), (65536, 1)):
        parts = [shard_range(g, w, r) for r in range(w)]
        assert parts[0][0] == 0 and sum(c for _, c in parts) == g
        for (a, ca), (b, _) in zip(parts, parts[1:]):
            assert a + ca == b


def test_two_rank_sharded_run_equals_unsharded(tmp_path, oracle_mod):
    from gym_trading_env_amd.config import make_config
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered.npz")
    env = oracle_mod.OracleEnv(make_config(n_envs=G, n_static=3, env_id_base=0, **KW), [_data()])
    env.reset()
    actions = np.random.default_rng(5).integers(-1, 3, (STEPS, G)).astype(np.int32)
    ends = 0
    for k in range(STEPS):
        env.step(actions[k])
        np.testing.assert_array_equal(got["reward"][k], env.reward)
        np.testing.assert_array_equal(got["term"][k], env.terminated.astype(bool))
        np.testing.assert_array_equal(got["trunc"][k], env.truncated.astype(bool))
        np.testing.assert_array_equal(got["obs"][k], env.obs)
        ends += int(env.truncated.sum() + env.terminated.sum())
    assert ends > G  # episodes ended and were re-drawn from the per-global-id streams
