"""Sharding the environments over the GPUs of a node (one process per GPU).

Environments are independent (no cross-env term anywhere in `TradingEnv.step`,
reference environments.py:233-272), so the shard is a contiguous env range per rank
and the data path needs no collective.  The only exchange is the RETURN of a step: an
RCCL all-gather (torch.distributed backend "nccl" over xGMI; "gloo" in the CPU tests)
of the packed per-env records (reward f32 | terminated u8 | truncated u8 = 6 bytes per
env) and, on request, of the observations.  Reset draws are keyed by GLOBAL env id
(`env_id_base`), so a sharded run reproduces the unsharded one bit for bit.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(global_envs: int, world: int, rank: int) -> tuple[int, int]:
    """(first global env id, count) of `rank`'s contiguous shard."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    base, rem = divmod(int(global_envs), int(world))
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def shard_datasets(n_datasets: int, world: int, rank: int) -> range:
    """Indices of the datasets `rank` keeps resident when the symbols are PARTITIONED over the
    ranks (config 5: 1 024 symbols, 128 per GPU): a contiguous block, like the env shards, so
    that envs and the tables they read sit on the same GPU and dataset switching stays local."""
    first, count = shard_range(n_datasets, world, rank)
    if count == 0:
        raise ValueError(f"{n_datasets} datasets cannot be partitioned over {world} ranks")
    return range(first, first + count)


def packed_layout(n: int) -> dict:
    """Byte ranges of the packed per-step return of a shard of n envs."""
    return {"reward": (0, 4 * n), "terminated": (4 * n, 5 * n), "truncated": (5 * n, 6 * n),
            "bytes": 6 * n}


def pack_returns(reward: torch.Tensor, terminated: torch.Tensor, truncated: torch.Tensor,
                 out: torch.Tensor | None = None) -> torch.Tensor:
    """reward f32 [n], flags bool/u8 [n] -> uint8 [6n].  (BatchedTradingEnv with
    output="torch" already lets the kernel write straight into this layout:
    `env.packed_returns`.)"""
    n = reward.numel()
    lay = packed_layout(n)
    if out is None:
        out = torch.empty(lay["bytes"], dtype=torch.uint8, device=reward.device)
    out[lay["reward"][0]:lay["reward"][1]].copy_(reward.contiguous().view(torch.uint8))
    out[lay["terminated"][0]:lay["terminated"][1]].copy_(terminated.view(torch.uint8))
    out[lay["truncated"][0]:lay["truncated"][1]].copy_(truncated.view(torch.uint8))
    return out


class ReturnGather:
    """All-gather of the per-step returns of equally sized shards.

    `gather(packed)` -> (reward f32 [world, n], terminated bool [world, n],
    truncated bool [world, n]) as zero-copy views of the gathered buffer; row r is
    rank r's shard, i.e. global env ids r*n .. r*n+n-1.  With `block=K > 1` one call moves
    the returns of K consecutive steps (uint8 [K, 6n] in, views [world, K, n] out)."""

    def __init__(self, n_local: int, device, group=None, obs_shape=None, depth=1, block=1):
        self.group = group
        self.world = dist.get_world_size(group)
        self.n = int(n_local)
        self.block = max(1, int(block))
        self.lay = packed_layout(self.n)
        self.nbytes = self.block * self.lay["bytes"]
        # `depth` gathered buffers: gather_async() may have that many collectives in flight
        self.bufs = [torch.empty(self.world * self.nbytes, dtype=torch.uint8, device=device)
                     for _ in range(max(1, int(depth)))]
        self.buf = self.bufs[0]
        self._turn = 0
        self.obs_buf = None
        if obs_shape is not None:
            self.obs_buf = torch.empty((self.world * self.n,) + tuple(obs_shape),
                                       dtype=torch.float32, device=device)

    def _check(self, packed):
        if packed.numel() != self.nbytes or packed.dtype != torch.uint8 or not packed.is_contiguous():
            raise ValueError(f"packed must be contiguous uint8 [{self.block} x 6*n_local]")

    def gather(self, packed: torch.Tensor):
        self._check(packed)
        dist.all_gather_into_tensor(self.buf, packed.view(-1), group=self.group)
        return self._views(self.buf)

    def _views(self, buf):
        rows = buf.view(self.world, self.block, self.lay["bytes"])
        if self.block == 1:
            rows = rows[:, 0]
        r0, r1 = self.lay["reward"]
        t0, t1 = self.lay["terminated"]
        u0, u1 = self.lay["truncated"]
        return (rows[..., r0:r1].view(torch.float32), rows[..., t0:t1].view(torch.bool),
                rows[..., u0:u1].view(torch.bool))

    def gather_async(self, packed: torch.Tensor) -> "PendingReturns":
        """Start the all-gather and return at once; `.wait()` on the result gives the views.

        The collective runs on the backend's own stream (RCCL), ordered after everything
        already enqueued on the current stream, so the NEXT step's kernel overlaps it.
        `packed` must not be rewritten before `.wait()`: use BatchedTradingEnv(return_slots=K)
        and keep at most K-1 gathers pending while stepping (K = `depth` here)."""
        self._check(packed)
        buf = self.bufs[self._turn]
        self._turn = (self._turn + 1) % len(self.bufs)
        work = dist.all_gather_into_tensor(buf, packed.view(-1), group=self.group, async_op=True)
        return PendingReturns(self, buf, work)

    def gather_obs(self, obs: torch.Tensor) -> torch.Tensor:
        """obs f32 [n, ...] -> [world*n, ...] (config 4: xGMI-bound, 2 560 B per env)."""
        if self.obs_buf is None:
            raise ValueError("constructed without obs_shape")
        dist.all_gather_into_tensor(self.obs_buf, obs.contiguous(), group=self.group)
        return self.obs_buf


class PendingReturns:
    """An all-gather in flight (ReturnGather.gather_async)."""

    def __init__(self, owner: ReturnGather, buf, work):
        self._owner, self._buf, self._work = owner, buf, work

    def wait(self):
        """Order the current stream (GPU backends) / the host (gloo) after the collective
        and return (reward [world, n], terminated [world, n], truncated [world, n])."""
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self._owner._views(self._buf)


class NativeReturnGather:
    """The per-step return all-gather through libgte's OWN RCCL communicator
    (`gte_comm_init` / `gte_allgather_returns`, csrc/gte_comm.hip): the collective is enqueued by
    the library on the env's stream right behind the step kernel — no torch.distributed call, no
    host synchronisation, no cross-stream event per step.  torch.distributed is used ONCE, as
    the host channel that hands rank 0's communicator id to the other ranks (any backend).

    `gather()` -> (reward f32 [world, n], terminated bool [world, n], truncated bool [world, n])
    views of the gathered buffer, stream-ordered on the env's stream (torch's current stream for
    an env built with output="torch").  `lib` / `handle` let the id exchange and the view layout
    be exercised without a GPU (tests/test_distributed_cpu.py)."""

    def __init__(self, env, group=None, with_obs=False, lib=None, handle=None, mode=0):
        import ctypes as C
        from . import _abi
        self._C, self._abi = C, _abi
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.lib = lib if lib is not None else env._lib
        self.handle = handle if handle is not None else env._h
        self.n = int(env.num_envs)
        self.mode = int(mode)
        self.lay = packed_layout(self.n)
        ident = (C.c_uint8 * _abi.GTE_COMM_ID_BYTES)()
        if self.rank == 0:
            _abi.check(self.lib, self.lib.gte_comm_unique_id(ident))
        box = [bytes(ident)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0,
                                   group=group)
        ident = (C.c_uint8 * _abi.GTE_COMM_ID_BYTES).from_buffer_copy(box[0])
        _abi.check(self.lib, self.lib.gte_comm_init(self.handle, ident, self.rank, self.world))
        dev = env.packed_returns.device
        # mode 1 keeps two gathers in flight: one destination per rotating return buffer
        self.bufs = [torch.empty(self.world * self.lay["bytes"], dtype=torch.uint8, device=dev)
                     for _ in range(2 if self.mode == 1 else 1)]
        self.buf = self.bufs[0]
        self._turn = 0
        self.obs_buf = None
        if with_obs:
            self.obs_buf = torch.empty((self.world * self.n,) + tuple(env.obs_shape),
                                       dtype=torch.float32, device=dev)
        self._closed = False

    def gather(self):
        """All-gather the packed returns of the step just enqueued."""
        out = self._C.c_void_p()
        self.buf = self.bufs[self._turn % len(self.bufs)]
        self._turn += 1
        self._abi.check(self.lib, self.lib.gte_allgather_returns(
            self.handle, self._C.c_void_p(self.buf.data_ptr()), self.mode, self._C.byref(out)))
        return self.views()

    def views(self, buf=None):
        rows = (self.buf if buf is None else buf).view(self.world, self.lay["bytes"])
        r0, r1 = self.lay["reward"]
        t0, t1 = self.lay["terminated"]
        u0, u1 = self.lay["truncated"]
        return (rows[:, r0:r1].view(torch.float32), rows[:, t0:t1].view(torch.bool),
                rows[:, u0:u1].view(torch.bool))

    def gather_obs(self):
        if self.obs_buf is None:
            raise ValueError("constructed without with_obs=True")
        self._abi.check(self.lib, self.lib.gte_allgather_obs(
            self.handle, self._C.c_void_p(self.obs_buf.data_ptr()), self.mode))
        return self.obs_buf

    def wait(self, back: int = 0):
        """mode 1: order the env's stream after the overlapped gather issued `back` gathers ago
        (0 = the last one).  With an env built with return_slots=2, `wait(1)` before every step
        keeps one gather overlapping the next step while never letting a step rewrite a buffer a
        gather still reads; views returned by gather() then describe the step after `wait(0)`."""
        self._abi.check(self.lib, self.lib.gte_comm_wait(self.handle, int(back)))

    def close(self):
        if not self._closed:
            self._closed = True
            self._abi.check(self.lib, self.lib.gte_comm_destroy(self.handle))


class ReturnPipeline:
    """Block-wise, overlapped all-gather of a BatchedTradingEnv's returns.

    The env is built with `return_slots = depth * block`: step t writes row t % slots of one
    [slots, 6n] buffer.  Every `block` steps the finished block (contiguous rows) is
    all-gathered asynchronously on the collective's stream while the following steps fill the
    next block; a block is waited for only when its rows are about to be rewritten, `depth`
    blocks later.  Per step this costs no cross-stream dependency and no host work except on
    block boundaries (measured on one MI355X with a 1-rank RCCL group, config 3: synchronous
    per-step gather 48 us/step, per-step asynchronous 58-68 us/step — host and cross-stream
    bound —, block=16 41.7 us/step, no gather 39.1 us/step; profiles/r01_gather_rehearsal.md).
    block=1, depth>=2 is the per-step form."""

    def __init__(self, env, returns: ReturnGather, block: int, depth: int):
        if depth < 2 or block < 1:
            raise ValueError("ReturnPipeline needs depth >= 2 and block >= 1")
        if env.return_slots != block * depth or returns.block != block or len(returns.bufs) < depth:
            raise ValueError("env.return_slots must be block*depth and the ReturnGather must "
                             "be built with the same block and depth")
        self.env, self.returns, self.block, self.depth = env, returns, block, depth
        self._pending = [None] * depth

    def before_step(self):
        s = self.env.return_slot
        if s % self.block == 0:  # the step is about to rewrite the first row of block b
            b = s // self.block
            if self._pending[b] is not None:
                self._pending[b].wait()
                self._pending[b] = None

    def after_step(self):
        """-> PendingReturns when the step completed a block, else None."""
        s = self.env._ret_slot
        if s % self.block != self.block - 1:
            return None
        b = s // self.block
        h = self.returns.gather_async(self.env.return_block(b * self.block, self.block))
        self._pending[b] = h
        return h

    def drain(self):
        for b, h in enumerate(self._pending):
            if h is not None:
                h.wait()
                self._pending[b] = None

    def flush(self):
        """Gather the block in progress too (its unwritten rows travel as they are), then
        wait for everything: after it, every step taken so far has had its returns gathered.
        -> PendingReturns of the partial block, or None when the last step closed a block."""
        h = None
        s = self.env._ret_slot
        if s % self.block != self.block - 1:
            b = s // self.block
            if self._pending[b] is not None:
                self._pending[b].wait()
            h = self.returns.gather_async(self.env.return_block(b * self.block, self.block))
            self._pending[b] = h
        self.drain()
        return h


class ShardedTradingEnv:
    """This rank's shard of a `global_envs`-environment BatchedTradingEnv.

    step() returns the LOCAL observations (device resident) and the GLOBAL reward /
    terminated / truncated ([world, n_local] views); `gather_obs=True` also returns the
    global observations instead of the local ones.  Datasets are replicated on every rank
    unless `partition_datasets=True` (then rank r keeps block r of the list, and its envs
    only ever visit those)."""

    def __init__(self, df, global_envs: int, *, group=None, device=None, gather_obs=False,
                 pipeline=1, block=1, partition_datasets=False, native_gather=False, **kw):
        from .batched import BatchedTradingEnv
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if partition_datasets:  # config 5: each rank keeps only its block of the symbols
            if not isinstance(df, (list, tuple)) or len(df) < self.world:
                raise ValueError("partition_datasets needs a list of at least `world` datasets")
            self.dataset_indices = shard_datasets(len(df), self.world, self.rank)
            df = [df[i] for i in self.dataset_indices]
        if global_envs % self.world:
            raise ValueError("global_envs must be a multiple of the world size")
        self.first, self.n_local = shard_range(global_envs, self.world, self.rank)
        dev_index = torch.cuda.current_device() if device is None else device
        self.pipeline, self.block = max(1, int(pipeline)), max(1, int(block))
        if self.block > 1 and self.pipeline < 2:
            raise ValueError("block > 1 needs pipeline >= 2")
        self.env = BatchedTradingEnv(df, num_envs=self.n_local, env_id_base=self.first,
                                     device=dev_index, output="torch",
                                     return_slots=self.pipeline * self.block, **kw)
        self.gather_obs = gather_obs
        dev = self.env.packed_returns.device
        obs_shape = self.env.obs_shape if gather_obs else None
        # step(): synchronous per-step gather; step_async(): the pipeline (own buffers)
        self.returns = ReturnGather(self.n_local, dev, group, obs_shape=obs_shape)
        # native_gather=True: step() gathers through libgte's own RCCL communicator, enqueued
        # by the library behind the step kernel (GPU backends only)
        self.native = (NativeReturnGather(self.env, group, with_obs=gather_obs)
                       if native_gather else None)
        self._pipe = None
        if self.pipeline >= 2:
            self._pipe = ReturnPipeline(self.env, ReturnGather(self.n_local, dev, group,
                                                               depth=self.pipeline, block=self.block),
                                        self.block, self.pipeline)

    def reset(self, **kw):
        obs, info = self.env.reset(**kw)
        return (self.returns.gather_obs(obs) if self.gather_obs else obs), info

    def step(self, local_actions):
        if self._pipe is not None:
            self._pipe.before_step()  # never rewrite rows a pending block gather still reads
        obs, _, _, _, info = self.env.step(local_actions)
        if self.native is not None:
            reward, term, trunc = self.native.gather()
            if self.gather_obs:
                obs = self.native.gather_obs()
            return obs, reward, term, trunc, info
        reward, term, trunc = self.returns.gather(self.env.packed_returns)
        if self.gather_obs:
            obs = self.returns.gather_obs(obs)
        return obs, reward, term, trunc, info

    def step_async(self, local_actions):
        """step() whose global returns arrive later: -> (local obs, pending | None, info).

        Needs `pipeline=P >= 2`.  Every `block`-th call returns a PendingReturns for the block
        of steps just completed (None otherwise); its all-gather runs on the collective's
        stream while the following steps execute — each GPU acts on its LOCAL observations,
        the global returns are what the learner consumes a rollout block at a time.
        `.wait()` gives (reward, terminated, truncated) as [world, n_local] (block=1) or
        [world, block, n_local]; those views are reused P blocks later."""
        if self._pipe is None:
            raise ValueError("step_async needs ShardedTradingEnv(pipeline >= 2)")
        self._pipe.before_step()
        obs, _, _, _, info = self.env.step(local_actions)
        return obs, self._pipe.after_step(), info

    def drain(self):
        """Wait for every all-gather started by step_async."""
        if self._pipe is not None:
            self._pipe.drain()

    def close(self):
        self.drain()
        if self.native is not None:
            self.native.close()
        self.env.close()
