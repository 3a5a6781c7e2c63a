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
        env = ShardedTradingEnv((feat, close), G, gather_obs=True, pipeline=3, **kw)
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
        # pipelined: up to 2 all-gathers in flight on RCCL's stream while the next steps run
        # (3 rotating return buffers); each handle is consumed two steps late
        handles, expect = [], []
        for k in range(40):
            a = rng.integers(-1, 3, G).astype(np.int32)
            _, pending, _ = env.step_async(torch.from_numpy(a).cuda())
            ora.step(a)
            handles.append(pending)
            expect.append((ora.reward.copy(), ora.terminated.astype(bool), ora.truncated.astype(bool)))
            if k >= 2:
                reward, term, trunc = handles[k - 2].wait()
                np.testing.assert_array_equal(reward.cpu().numpy().reshape(-1), expect[k - 2][0])
                np.testing.assert_array_equal(term.cpu().numpy().reshape(-1), expect[k - 2][1])
                np.testing.assert_array_equal(trunc.cpu().numpy().reshape(-1), expect[k - 2][2])
        env.drain()
        env.close()
        # block form: one all-gather per 4 steps, 2 blocks in rotation
        env = ShardedTradingEnv((feat, close), G, pipeline=2, block=4, **kw)
        ora = oracle_mod.OracleEnv(env.env.cfg, [(full, close)])
        env.reset()
        ora.reset()
        expect, seen = [], 0
        for k in range(22):
            a = rng.integers(-1, 3, G).astype(np.int32)
            obs, pending, _ = env.step_async(torch.from_numpy(a).cuda())
            ora.step(a)
            np.testing.assert_array_equal(obs.cpu().numpy(), ora.obs)
            expect.append((ora.reward.copy(), ora.terminated.astype(bool)))
            assert (pending is not None) == (k % 4 == 3)
            if pending is not None:
                reward, term, _ = pending.wait()
                assert reward.shape == (1, 4, G)
                for j in range(4):
                    np.testing.assert_array_equal(reward[0, j].cpu().numpy(), expect[k - 3 + j][0])
                    np.testing.assert_array_equal(term[0, j].cpu().numpy(), expect[k - 3 + j][1])
                seen += 4
        tail = env._pipe.flush()  # steps 20, 21: the unfinished block
        reward, _, _ = tail.wait()
        np.testing.assert_array_equal(reward[0, 0].cpu().numpy(), expect[20][0])
        np.testing.assert_array_equal(reward[0, 1].cpu().numpy(), expect[21][0])
        assert seen == 20
        env.close()
    finally:
        dist.destroy_process_group()


def test_c_abi_allgather_single_rank_rccl(oracle_mod):
    """gte_comm_unique_id / gte_comm_init / gte_allgather_returns / gte_allgather_obs /
    gte_allgather (include/gte.h): the return exchange through libgte's own RCCL communicator,
    driven through the C ABI alone (ctypes, no torch on the path), 1-rank group on the test box:
    the collective degenerates to a copy but runs RCCL's real code path.  Mode 0 (env stream)
    and mode 1 (communication stream + gte_comm_wait) against the oracle."""
    import ctypes as C
    from gym_trading_env_amd import _abi
    from gym_trading_env_amd.batched import BatchedTradingEnv
    rng = np.random.default_rng(1)
    T, Fs, N = 300, 6, 900
    close = 100 * np.exp(np.cumsum(rng.normal(0, 1e-2, T)))
    feat = rng.normal(0, 1, (T, Fs)).astype(np.float32)
    env = BatchedTradingEnv((feat, close), num_envs=N, positions=[-1, 0, 1], windows=5, trading_fees=1e-4,
                            max_episode_duration=20, autoreset="next_step", seed=4, output="numpy")
    lib, h = env._lib, env._h
    full = np.zeros((T, Fs + 2), np.float32); full[:, :Fs] = feat
    ora = oracle_mod.OracleEnv(env.cfg, [(full, close)])
    ident = (C.c_uint8 * _abi.GTE_COMM_ID_BYTES)()
    with pytest.raises(_abi.GteError, match="gte_comm_init"):
        _abi.check(lib, lib.gte_allgather_returns(h, None, 0, None))
    _abi.check(lib, lib.gte_comm_unique_id(ident))
    assert any(ident)
    _abi.check(lib, lib.gte_comm_init(h, ident, 0, 1))
    with pytest.raises(_abi.GteError, match="already"):
        _abi.check(lib, lib.gte_comm_init(h, ident, 0, 1))
    env.reset(); ora.reset()
    out = C.c_void_p()
    for k in range(45):
        a = rng.integers(-1, 3, N).astype(np.int32)
        env._launch_step(a); ora.step(a)
        mode = k % 2
        _abi.check(lib, lib.gte_allgather_returns(h, None, mode, C.byref(out)))
        if mode == 1:
            _abi.check(lib, lib.gte_comm_wait(h, 0))  # the env's stream now follows the gather
        packed = env._to_host(out.value, np.uint8, 6 * N)
        np.testing.assert_array_equal(packed[:4 * N].view(np.float32), ora.reward)
        np.testing.assert_array_equal(packed[4 * N:5 * N], ora.terminated)
        np.testing.assert_array_equal(packed[5 * N:], ora.truncated)
    # observations, and the generic form on an arbitrary device buffer (here: reward64)
    env2 = BatchedTradingEnv((feat, close), num_envs=N, positions=[-1, 0, 1], windows=5, output="numpy")
    env2.reset()  # a second env only lends device memory to receive into
    _abi.check(lib, lib.gte_allgather_obs(h, C.c_void_p(env2._out.obs), 0))
    np.testing.assert_array_equal(env2._to_host(env2._out.obs, np.float32, N * 5 * 8).reshape(N, 5, 8), ora.obs)
    _abi.check(lib, lib.gte_allgather(h, C.c_void_p(env._out.reward64), C.c_void_p(env2._out.reward64), 8 * N, 1))
    _abi.check(lib, lib.gte_comm_synchronize(h))
    np.testing.assert_array_equal(env2._to_host(env2._out.reward64, np.float64, N), ora.reward64)
    _abi.check(lib, lib.gte_comm_destroy(h))
    _abi.check(lib, lib.gte_comm_destroy(h))  # idempotent
    env2.close()
    env.close()


def test_sharded_env_native_gather_single_rank(oracle_mod):
    """ShardedTradingEnv(native_gather=True): step() gathers through libgte's communicator
    (NativeReturnGather) instead of torch.distributed; same results."""
    import torch
    import torch.distributed as dist
    from gym_trading_env_amd.distributed import ShardedTradingEnv
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 1000))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=0, world_size=1)  # only the id hand-off uses it
    try:
        rng = np.random.default_rng(0)
        T, Fs, G = 300, 6, 640
        close = 100 * np.exp(np.cumsum(rng.normal(0, 1e-2, T)))
        feat = rng.normal(0, 1, (T, Fs)).astype(np.float32)
        env = ShardedTradingEnv((feat, close), G, gather_obs=True, native_gather=True, device=0,
                                positions=[-1, 0, 1], windows=5, trading_fees=1e-4,
                                max_episode_duration=20, autoreset="next_step", seed=4)
        full = np.zeros((T, Fs + 2), np.float32); full[:, :Fs] = feat
        ora = oracle_mod.OracleEnv(env.env.cfg, [(full, close)])
        env.reset(); ora.reset()
        for k in range(30):
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


def test_return_slots_keep_older_returns_intact(oracle_mod):
    """return_slots=3: step t writes packed buffer t % 3; the buffers of steps t-1 and t-2
    still hold those steps' returns (what an all-gather in flight reads)."""
    import torch
    from gym_trading_env_amd.batched import BatchedTradingEnv
    rng = np.random.default_rng(2)
    T, Fs, N = 400, 3, 700
    close = 100 * np.exp(np.cumsum(rng.normal(0, 2e-2, T)))
    feat = rng.normal(0, 1, (T, Fs)).astype(np.float32)
    kw = dict(positions=[-1, 0, 1], windows=None, trading_fees=1e-3, borrow_interest_rate=1e-4,
              max_episode_duration=15, autoreset="same_step", seed=9)
    env = BatchedTradingEnv((feat, close), num_envs=N, return_slots=3, **kw)
    full = np.zeros((T, Fs + 2), np.float32)
    full[:, :Fs] = feat
    ora = oracle_mod.OracleEnv(env.cfg, [(full, close)])
    env.reset()
    ora.reset()
    kept = []
    for k in range(25):
        a = rng.integers(-1, 3, N).astype(np.int32)
        # the slot rotates whichever way the actions arrive: device tensor, numpy, list, CPU tensor
        arg = (torch.from_numpy(a).cuda(), a, a.tolist(), torch.from_numpy(a))[k % 4]
        _, reward, term, trunc, _ = env.step(arg)
        ora.step(a)
        kept.append((env.packed_returns, reward, term, trunc, ora.reward.copy(),
                     ora.terminated.astype(bool), ora.truncated.astype(bool)))
        for packed, r, t, u, er, et, eu in kept[-3:]:  # this step and the two before it
            np.testing.assert_array_equal(r.cpu().numpy(), er)
            np.testing.assert_array_equal(t.cpu().numpy(), et)
            np.testing.assert_array_equal(u.cpu().numpy(), eu)
            np.testing.assert_array_equal(packed[:4 * N].view(torch.float32).cpu().numpy(), er)
        np.testing.assert_array_equal(env.read_output("reward"), ora.reward)
        np.testing.assert_array_equal(env.read_output("terminated").astype(bool),
                                      ora.terminated.astype(bool))
    assert len({k[0].data_ptr() for k in kept}) == 3
    with pytest.raises(ValueError):
        BatchedTradingEnv((feat, close), num_envs=N, return_slots=2, output="numpy", **kw)
    env.close()


def _rank_main(rank, world, port, out_dir):
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gym_trading_env_amd.distributed import ShardedTradingEnv
        torch.cuda.set_device(0)
        rng = np.random.default_rng(0)
        T, Fs, G = 300, 6, 1024
        close = 100 * np.exp(np.cumsum(rng.normal(0, 1e-2, T)))
        feat = rng.normal(0, 1, (T, Fs)).astype(np.float32)
        kw = dict(positions=[-1, 0, 1], windows=5, trading_fees=1e-4, max_episode_duration=20,
                  autoreset="next_step", seed=4)
        env = ShardedTradingEnv((feat, close), G, device=0, gather_obs=True, pipeline=2, **kw)
        env.reset()
        acts = np.random.default_rng(1).integers(-1, 3, (30, G)).astype(np.int32)
        rec = []
        prev = None
        for k in range(30):
            a = torch.from_numpy(acts[k, env.first:env.first + env.n_local]).cuda()
            if k < 15:
                obs, reward, term, trunc, _ = env.step(a)
                rec.append((obs.cpu().numpy().copy(), reward.cpu().numpy().reshape(-1).copy(),
                            term.cpu().numpy().reshape(-1).copy(),
                            trunc.cpu().numpy().reshape(-1).copy()))
                continue
            obs, pending, _ = env.step_async(a)  # returns consumed one step late
            obs = env.returns.gather_obs(obs).cpu().numpy().copy()
            if prev is not None:
                reward, term, trunc = prev[0].wait()
                rec.append((prev[1], reward.cpu().numpy().reshape(-1).copy(),
                            term.cpu().numpy().reshape(-1).copy(),
                            trunc.cpu().numpy().reshape(-1).copy()))
            prev = (pending, obs)
        reward, term, trunc = prev[0].wait()
        rec.append((prev[1], reward.cpu().numpy().reshape(-1).copy(),
                    term.cpu().numpy().reshape(-1).copy(), trunc.cpu().numpy().reshape(-1).copy()))
        if rank == 0:
            np.savez(os.path.join(out_dir, "g.npz"), obs=np.stack([r[0] for r in rec]),
                     reward=np.stack([r[1] for r in rec]), term=np.stack([r[2] for r in rec]),
                     trunc=np.stack([r[3] for r in rec]), acts=acts, feat=feat, close=close)
        env.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_unsharded_oracle(tmp_path, oracle_mod):
    """world_size 2 for real: both ranks run their shard through the HIP kernels on the
    box's single GPU; the return gather goes through gloo (RCCL refuses two ranks on one
    device).  The gathered (obs, reward, flags) must equal the unsharded oracle."""
    import torch.multiprocessing as mp
    from gym_trading_env_amd.config import make_config
    port = 29700 + os.getpid() % 1000
    mp.spawn(_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = np.load(tmp_path / "g.npz")
    full = np.zeros((300, 8), np.float32)
    full[:, :6] = g["feat"]
    cfg = make_config(n_envs=1024, n_static=6, positions=[-1, 0, 1], windows=5, trading_fees=1e-4,
                      max_episode_duration=20, autoreset="next_step", seed=4)
    ora = oracle_mod.OracleEnv(cfg, [(full, g["close"])])
    ora.reset()
    for k in range(30):
        ora.step(g["acts"][k])
        np.testing.assert_array_equal(g["obs"][k], ora.obs)
        np.testing.assert_array_equal(g["reward"][k], ora.reward)
        np.testing.assert_array_equal(g["term"][k], ora.terminated.astype(bool))
        np.testing.assert_array_equal(g["trunc"][k], ora.truncated.astype(bool))


def test_bench_gpus_2_from_a_bare_shell_launches_its_own_ranks():
    """`python3 bench.py --gpus 2` without a launcher around it: the parent starts the two ranks
    as fresh child processes (torch.distributed.run) before touching the GPU, relays rank 0's ONE
    JSON line and exits 0.  Rehearsal configuration of a one-GPU box: both ranks share device 0
    and exchange their returns over gloo (RCCL refuses two ranks on one device)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GTE_BENCH_BACKEND="gloo", GTE_BENCH_SHARE_DEVICE="1")
    for workload, extra in (("c3", ["--envs", "8192"]), ("c4", ["--envs", "4096"])):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20",
                            "--warmup", "5", "--workload", workload, "--no-cpu-baseline", "--no-pmc"] + extra,
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, r.stdout
        out = json.loads(lines[0])
        assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 5
        assert out["config"]["global_envs"] == 2 * out["config"]["envs_per_gpu"]
        assert out["value"] > 0 and out["scaling"] == "weak"
        if workload == "c4":  # BASELINE config 4: with the observation all-gather, and without beside it
            assert "observations all-gathered" in out["config"]["workload"]
            assert out["config"]["without_obs_gather"]["value"] >= out["value"] * 0.5
