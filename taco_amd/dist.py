"""Env-sharded multi-GPU layer (one process per GPU, torch.distributed over RCCL/xGMI).

The reference is single-process (no NCCL / torch.distributed call site anywhere), so nothing
here has a reference counterpart; it follows SURVEY.md section 8(e): envs are independent, so each rank owns a
contiguous slice of GLOBAL env ids (which key the random streams and the FpvMix thirds -- results do not depend on the
number of ranks), and one all-gather per step publishes the packed per-rank block [obs stack | reward | done | time-out].
The step kernel fills that block itself (taco_bind_gather_block), so a step is one kernel launch + one collective.
"""
import torch

from . import _lib


def shard_bounds(n_global, world_size, rank):
    """Contiguous slice [lo, hi) of global env ids owned by `rank`; the first n_global % world_size ranks get one more."""
    base, rem = divmod(int(n_global), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def block_row(len_obs):
    """floats per env in the gather block: obs stack + reward + done + time-out, padded to whole 128-byte lines
    (= taco_gather_row_floats of the C ABI)"""
    return (len_obs * 26 + 3 + 31) // 32 * 32


def unpack_block(block, len_obs):
    """[n, len_obs*26+3] f32 -> (obs [n,len_obs,26] f32, rew [n] f32, done [n] i64, time_outs [n] bool)"""
    n = block.shape[0]
    row = len_obs * 26
    obs = block[:, :row].reshape(n, len_obs, 26)
    return obs, block[:, row], block[:, row + 1].to(torch.long), block[:, row + 2] != 0


def pack_block(obs, rew, done, timeout):
    """Host-side packer with the kernel's block layout (used by the CPU tests, where no kernel fills the block)."""
    n = obs.shape[0]
    row = obs.shape[1] * 26
    blk = torch.zeros((n, block_row(obs.shape[1])), dtype=torch.float32, device=obs.device)
    blk[:, :row] = obs.reshape(n, -1)
    blk[:, row] = rew.float()
    blk[:, row + 1] = done.float()
    blk[:, row + 2] = timeout.float()
    return blk


def all_gather_blocks(block, n_global, world_size, group=None):
    """ONE collective: every rank's [n_r, row] block -> [n_global, row] in global env order.  Equal shards use
    all_gather_into_tensor directly; ragged shards are padded to the largest shard and compacted afterwards."""
    import torch.distributed as dist
    row = block.shape[1]
    sizes = [hi - lo for lo, hi in (shard_bounds(n_global, world_size, r) for r in range(world_size))]
    if len(set(sizes)) == 1:
        out = torch.empty((n_global, row), dtype=block.dtype, device=block.device)
        dist.all_gather_into_tensor(out, block, group=group)
        return out
    m = max(sizes)
    padded = torch.zeros((m, row), dtype=block.dtype, device=block.device)
    padded[: block.shape[0]] = block
    out = torch.empty((world_size * m, row), dtype=block.dtype, device=block.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * m: r * m + sizes[r]] for r in range(world_size)], dim=0)


class ShardedEnv:
    """This rank's slice of a `numEnvs`-wide job on its own MI355X."""

    def __init__(self, cfg, rank, world_size, device, gather=True, group=None):
        from .vec_env import FpvBase
        self.rank, self.world_size, self.group, self.gather = rank, world_size, group, gather
        self.n_global = int(cfg["env"]["numEnvs"])
        self.lo, self.hi = shard_bounds(self.n_global, world_size, rank)
        self.env = FpvBase(cfg, rl_device=str(device), sim_device=str(device), env_offset=self.lo, num_envs_local=self.hi - self.lo,
                           copy_outputs=False)
        self.len_obs = self.env.len_obs
        self.block = torch.zeros((self.hi - self.lo, block_row(self.len_obs)), dtype=torch.float32, device=self.env.device)
        _lib.check(self.env.lib.taco_bind_gather_block(self.env._h, self.block.data_ptr()))
        self.gathered = None

    def step_gathered(self, local_actions):
        """One step of this rank's envs; returns the all-gathered block (or the local one if gather is off)."""
        self.env.step_raw(local_actions)
        if not self.gather or self.world_size == 1:
            return self.block
        self.gathered = all_gather_blocks(self.block, self.n_global, self.world_size, self.group)
        return self.gathered

    def step(self, local_actions):
        """-> (obs [N_global,len_obs,26], rew, done, time_outs) for the whole job, plus this rank's critic states."""
        blk = self.step_gathered(local_actions)
        obs, rew, done, tmo = unpack_block(blk, self.len_obs)
        return {"obs": obs, "states": self.env.states_buf}, rew, done, {"time_outs": tmo}
