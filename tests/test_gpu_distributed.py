"""The sharded env through RCCL on the one GPU of the test box (world_size 1: the
collective degenerates to a copy but runs the real nccl code path and the packed
output binding); the world_size-2 logic is covered on CPU by test_distributed_cpu."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_sharded_env_single_rank_nccl(oracle_mod):
    import torch
    import torch.distributed as dist
    from gym_trading_env_amd.distributed import ShardedTradingEnv
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 1000))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rng = np.random.default_rng(0)
        T, Fs, G = 300, 6, 640
        close = 100 * np.exp(np.cumsum(rng.normal(0, 1e-2, T)))
        feat = rng.normal(0, 1, (T, Fs)).astype(np.float32)
        kw = dict(positions=[-1, 0, 1], windows=5, trading_fees=1e-4, max_episode_duration=20,
                  autoreset="next_step", seed=4)
        env = ShardedTradingEnv((feat, close), G, gather_obs=True, **kw)
        full = np.zeros((T, Fs + 2), np.float32)
        full[:, :Fs] = feat
        ora = oracle_mod.OracleEnv(env.env.cfg, [(full, close)])
        env.reset()
        ora.reset()
        for k in range(40):
            a = rng.integers(-1, 3, G).astype(np.int32)
            obs, reward, term, trunc, _ = env.step(torch.from_numpy(a).cuda())
            ora.step(a)
            assert reward.shape == (1, G) and obs.shape == (G, 5, 8)
            np.testing.assert_array_equal(obs.cpu().numpy(), ora.obs)
            np.testing.assert_array_equal(reward.cpu().numpy().reshape(-1), ora.reward)
            np.testing.assert_array_equal(term.cpu().numpy().reshape(-1), ora.terminated.astype(bool))
            np.testing.assert_array_equal(trunc.cpu().numpy().reshape(-1), ora.truncated.astype(bool))
        env.close()
    finally:
        dist.destroy_process_group()
