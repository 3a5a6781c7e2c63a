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
    rank r's shard, i.e. global env ids r*n .. r*n+n-1."""

    def __init__(self, n_local: int, device, group=None, obs_shape=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.n = int(n_local)
        self.lay = packed_layout(self.n)
        self.buf = torch.empty(self.world * self.lay["bytes"], dtype=torch.uint8, device=device)
        self.obs_buf = None
        if obs_shape is not None:
            self.obs_buf = torch.empty((self.world * self.n,) + tuple(obs_shape),
                                       dtype=torch.float32, device=device)

    def gather(self, packed: torch.Tensor):
        if packed.numel() != self.lay["bytes"] or packed.dtype != torch.uint8:
            raise ValueError("packed must be uint8 [6*n_local]")
        dist.all_gather_into_tensor(self.buf, packed, group=self.group)
        rows = self.buf.view(self.world, self.lay["bytes"])
        r0, r1 = self.lay["reward"]
        t0, t1 = self.lay["terminated"]
        u0, u1 = self.lay["truncated"]
        return (rows[:, r0:r1].view(torch.float32), rows[:, t0:t1].view(torch.bool),
                rows[:, u0:u1].view(torch.bool))

    def gather_obs(self, obs: torch.Tensor) -> torch.Tensor:
        """obs f32 [n, ...] -> [world*n, ...] (config 4: xGMI-bound, 2 560 B per env)."""
        if self.obs_buf is None:
            raise ValueError("constructed without obs_shape")
        dist.all_gather_into_tensor(self.obs_buf, obs.contiguous(), group=self.group)
        return self.obs_buf


class ShardedTradingEnv:
    """This rank's shard of a `global_envs`-environment BatchedTradingEnv.

    step() returns the LOCAL observations (device resident) and the GLOBAL reward /
    terminated / truncated ([world, n_local] views); `gather_obs=True` also returns the
    global observations instead of the local ones."""

    def __init__(self, df, global_envs: int, *, group=None, device=None, gather_obs=False, **kw):
        from .batched import BatchedTradingEnv
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if global_envs % self.world:
            raise ValueError("global_envs must be a multiple of the world size")
        self.first, self.n_local = shard_range(global_envs, self.world, self.rank)
        dev_index = torch.cuda.current_device() if device is None else device
        self.env = BatchedTradingEnv(df, num_envs=self.n_local, env_id_base=self.first,
                                     device=dev_index, output="torch", **kw)
        self.gather_obs = gather_obs
        self.returns = ReturnGather(self.n_local, self.env.packed_returns.device, group,
                                    obs_shape=self.env.obs_shape if gather_obs else None)

    def reset(self, **kw):
        obs, info = self.env.reset(**kw)
        return (self.returns.gather_obs(obs) if self.gather_obs else obs), info

    def step(self, local_actions):
        obs, _, _, _, info = self.env.step(local_actions)
        reward, term, trunc = self.returns.gather(self.env.packed_returns)
        if self.gather_obs:
            obs = self.returns.gather_obs(obs)
        return obs, reward, term, trunc, info

    def close(self):
        self.env.close()
