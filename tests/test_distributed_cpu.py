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
    rg = ReturnGather(n, "cpu", obs_shape=(4, 5), depth=2)
    actions = np.random.default_rng(5).integers(-1, 3, (STEPS, G)).astype(np.int32)
    rec = []
    slots = [torch.empty(6 * n, dtype=torch.uint8) for _ in range(2)]  # rotating sources
    pending = None  # (handle, obs) of the previous step: the pipelined form, depth 2
    for k in range(STEPS):
        env.step(actions[k, first:first + n])
        packed = pack_returns(torch.from_numpy(env.reward.copy()),
                              torch.from_numpy(env.terminated.copy()).bool(),
                              torch.from_numpy(env.truncated.copy()).bool(), out=slots[k % 2])
        obs = rg.gather_obs(torch.from_numpy(env.obs.copy())).numpy().copy()
        if k < STEPS // 2:  # synchronous gather
            reward, term, trunc = rg.gather(packed)
            rec.append((reward.reshape(-1).numpy().copy(), term.reshape(-1).numpy().copy(),
                        trunc.reshape(-1).numpy().copy(), obs))
            continue
        handle = rg.gather_async(packed)  # step k's gather is consumed during step k+1
        if pending is not None:
            reward, term, trunc = pending[0].wait()
            rec.append((reward.reshape(-1).numpy().copy(), term.reshape(-1).numpy().copy(),
                        trunc.reshape(-1).numpy().copy(), pending[1]))
        pending = (handle, obs)
    reward, term, trunc = pending[0].wait()
    rec.append((reward.reshape(-1).numpy().copy(), term.reshape(-1).numpy().copy(),
                trunc.reshape(-1).numpy().copy(), pending[1]))
    if rank == 0:
        np.savez(os.path.join(out_dir, "gathered.npz"),
                 reward=np.stack([r[0] for r in rec]), term=np.stack([r[1] for r in rec]),
                 trunc=np.stack([r[2] for r in rec]), obs=np.stack([r[3] for r in rec]))
    dist.barrier()
    dist.destroy_process_group()


class _SlotEnv:
    """The return-slot interface of BatchedTradingEnv (return_slots, return_slot,
    return_block, packed_returns) over the oracle, for the ReturnPipeline logic on CPU."""

    def __init__(self, ora, slots):
        self.ora, self.return_slots = ora, slots
        self.n = ora.cfg.n_envs
        self._packed_all = torch.zeros((slots, 6 * self.n), dtype=torch.uint8)
        self._ret_slot = slots - 1

    @property
    def return_slot(self):
        return (self._ret_slot + 1) % self.return_slots

    def return_block(self, first, count):
        return self._packed_all[first:first + count]

    def step(self, a):
        from gym_trading_env_amd.distributed import pack_returns
        self._ret_slot = self.return_slot
        self.ora.step(a)
        self.packed_returns = pack_returns(
            torch.from_numpy(self.ora.reward.copy()), torch.from_numpy(self.ora.terminated.copy()).bool(),
            torch.from_numpy(self.ora.truncated.copy()).bool(), out=self._packed_all[self._ret_slot])


def _block_worker(rank, world, port, out_dir, block, depth, steps):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gym_trading_env_amd.config import make_config
    from gym_trading_env_amd.distributed import ReturnGather, ReturnPipeline, shard_range
    from oracle import oracle
    first, n = shard_range(G, world, rank)
    ora = oracle.OracleEnv(make_config(n_envs=n, n_static=3, env_id_base=first, **KW), [_data()])
    ora.reset()
    env = _SlotEnv(ora, block * depth)
    pipe = ReturnPipeline(env, ReturnGather(n, "cpu", depth=depth, block=block), block, depth)
    actions = np.random.default_rng(5).integers(-1, 3, (steps, G)).astype(np.int32)
    rows = []  # gathered (reward, term, trunc) per step, in step order
    handles = []

    def collect(h, count):
        reward, term, trunc = h.wait()
        if block == 1:
            reward, term, trunc = reward[:, None], term[:, None], trunc[:, None]
        for j in range(count):
            rows.append((reward[:, j].reshape(-1).numpy().copy(), term[:, j].reshape(-1).numpy().copy(),
                         trunc[:, j].reshape(-1).numpy().copy()))

    for k in range(steps):
        pipe.before_step()
        env.step(actions[k, first:first + n])
        h = pipe.after_step()
        if h is not None:
            handles.append(h)
        if len(handles) == depth:  # consume as late as the rotation allows
            collect(handles.pop(0), block)
    for h in handles:
        collect(h, block)
    tail = pipe.flush()
    if tail is not None:
        collect(tail, steps % block)
    assert len(rows) == steps
    if rank == 0:
        np.savez(os.path.join(out_dir, "blocks.npz"), reward=np.stack([r[0] for r in rows]),
                 term=np.stack([r[1] for r in rows]), trunc=np.stack([r[2] for r in rows]))
    dist.barrier()
    dist.destroy_process_group()


class _HostCommLib:
    """Stand-in for libgte's communicator entry points (include/gte.h, gte_comm_*) over HOST
    memory and gloo, so that NativeReturnGather's id exchange, call sequence and view layout run
    without a GPU.  `handle` is this object's env stub; pointers are host addresses."""

    def __init__(self, env):
        self.env, self.rank, self.world, self.id_seen = env, None, None, None

    def gte_last_error(self):
        return b"stub"

    def gte_comm_unique_id(self, ident):
        for i in range(128):
            ident[i] = (i * 7 + 3) % 251  # what rank 0 "generated"
        return 0

    def gte_comm_init(self, handle, ident, rank, world):
        self.id_seen, self.rank, self.world = bytes(ident), rank, world
        return 0

    def _into(self, ptr, nbytes, src):
        import ctypes
        parts = [torch.empty_like(src) for _ in range(self.world)]
        dist.all_gather(parts, src)
        flat = torch.cat([x.view(torch.uint8).reshape(-1) for x in parts])
        dst = np.frombuffer((ctypes.c_char * nbytes).from_address(ptr.value), dtype=np.uint8)
        dst[:] = flat.numpy()

    def gte_allgather_returns(self, handle, dst, mode, out):
        self._into(dst, self.world * 6 * self.env.num_envs, self.env.packed_returns)
        return 0

    def gte_allgather_obs(self, handle, dst, mode):
        self._into(dst, self.world * self.env.obs.numel() * 4, self.env.obs)
        return 0

    def gte_comm_destroy(self, handle):
        return 0


def _native_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gym_trading_env_amd.config import make_config
    from gym_trading_env_amd.distributed import NativeReturnGather, ReturnGather, pack_returns, shard_range
    from oracle import oracle
    first, n = shard_range(G, world, rank)
    ora = oracle.OracleEnv(make_config(n_envs=n, n_static=3, env_id_base=first, **KW), [_data()])
    ora.reset()

    class Env:  # what NativeReturnGather reads from a BatchedTradingEnv
        num_envs, obs_shape = n, (4, 5)
        packed_returns = torch.zeros(6 * n, dtype=torch.uint8)
        obs = torch.zeros((n, 4, 5), dtype=torch.float32)
    env = Env()
    lib = _HostCommLib(env)
    native = NativeReturnGather(env, with_obs=True, lib=lib, handle=0)
    assert lib.rank == rank and lib.world == world
    assert lib.id_seen == bytes((i * 7 + 3) % 251 for i in range(128))  # rank 0's id reached every rank
    ref = ReturnGather(n, "cpu", obs_shape=(4, 5))
    actions = np.random.default_rng(5).integers(-1, 3, (STEPS, G)).astype(np.int32)
    rec = []
    for k in range(STEPS):
        ora.step(actions[k, first:first + n])
        pack_returns(torch.from_numpy(ora.reward.copy()), torch.from_numpy(ora.terminated.copy()).bool(),
                     torch.from_numpy(ora.truncated.copy()).bool(), out=env.packed_returns)
        env.obs.copy_(torch.from_numpy(ora.obs))
        reward, term, trunc = native.gather()
        r2, t2, u2 = ref.gather(env.packed_returns)  # the torch.distributed wrapper: same views
        assert torch.equal(reward, r2) and torch.equal(term, t2) and torch.equal(trunc, u2)
        obs = native.gather_obs()
        assert torch.equal(obs, ref.gather_obs(env.obs))
        rec.append((reward.reshape(-1).numpy().copy(), term.reshape(-1).numpy().copy(),
                    trunc.reshape(-1).numpy().copy(), obs.numpy().copy()))
    native.close()
    if rank == 0:
        np.savez(os.path.join(out_dir, "native.npz"), reward=np.stack([r[0] for r in rec]),
                 term=np.stack([r[1] for r in rec]), trunc=np.stack([r[2] for r in rec]),
                 obs=np.stack([r[3] for r in rec]))
    dist.barrier()
    dist.destroy_process_group()


def test_native_return_gather_wrapper_two_ranks(tmp_path, oracle_mod):
    """NativeReturnGather (the Python face of gte_comm_init / gte_allgather_returns /
    gte_allgather_obs): rank 0's communicator id reaches every rank, the gathered views have
    the layout of the torch.distributed wrapper, and the sharded run equals the unsharded one.
    The library calls are served by a host/gloo stand-in here; the real RCCL path runs in
    tests/test_gpu_distributed.py."""
    from gym_trading_env_amd.config import make_config
    port = 29700 + os.getpid() % 200
    mp.spawn(_native_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "native.npz")
    env = oracle_mod.OracleEnv(make_config(n_envs=G, n_static=3, env_id_base=0, **KW), [_data()])
    env.reset()
    actions = np.random.default_rng(5).integers(-1, 3, (STEPS, G)).astype(np.int32)
    for k in range(STEPS):
        env.step(actions[k])
        np.testing.assert_array_equal(got["reward"][k], env.reward)
        np.testing.assert_array_equal(got["term"][k], env.terminated.astype(bool))
        np.testing.assert_array_equal(got["trunc"][k], env.truncated.astype(bool))
        np.testing.assert_array_equal(got["obs"][k], env.obs)


@pytest.mark.parametrize("block,depth,steps", [(4, 2, 30), (1, 3, 20), (8, 2, 16)])
def test_block_pipeline_two_ranks_equals_unsharded(tmp_path, oracle_mod, block, depth, steps):
    """ReturnPipeline: returns gathered a block at a time, `depth` blocks in rotation, handles
    consumed as late as allowed, unfinished block flushed — equals the unsharded run."""
    from gym_trading_env_amd.config import make_config
    port = 29850 + os.getpid() % 100 + block
    mp.spawn(_block_worker, args=(2, port, str(tmp_path), block, depth, steps), nprocs=2, join=True)
    got = np.load(tmp_path / "blocks.npz")
    env = oracle_mod.OracleEnv(make_config(n_envs=G, n_static=3, env_id_base=0, **KW), [_data()])
    env.reset()
    actions = np.random.default_rng(5).integers(-1, 3, (steps, G)).astype(np.int32)
    for k in range(steps):
        env.step(actions[k])
        np.testing.assert_array_equal(got["reward"][k], env.reward)
        np.testing.assert_array_equal(got["term"][k], env.terminated.astype(bool))
        np.testing.assert_array_equal(got["trunc"][k], env.truncated.astype(bool))


def test_shard_datasets_partitions_the_symbols():
    from gym_trading_env_amd.distributed import shard_datasets
    for n, w in ((1024, 8), (10, 3), (8, 8)):
        blocks = [list(shard_datasets(n, w, r)) for r in range(w)]
        assert sum(blocks, []) == list(range(n)) and all(blocks)
    assert list(shard_datasets(1024, 8, 3)) == list(range(384, 512))
    with pytest.raises(ValueError):
        shard_datasets(3, 4, 3)


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


def test_bench_gpus_n_launches_its_own_ranks_and_relays_their_failure():
    """`python3 bench.py --gpus 2` from a bare shell (WORLD_SIZE unset) starts two rank
    processes under torch.distributed.run as CHILDREN, before touching the GPU itself.  Here
    (no GPU) every rank refuses to run — there is no CPU fallback — and the parent must relay
    that: non-zero exit, no JSON line on stdout."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU host: the working path is tests/test_gpu_distributed.py")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--no-pmc", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "--gpus 2 without a launcher" in r.stderr
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node=2" in r.stderr
    assert r.stderr.count("bench.py needs an MI355X") >= 1  # the ranks started and said why
    assert '{"metric"' not in r.stdout
