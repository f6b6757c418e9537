"""Env-sharded multi-GPU layer (one process per GPU, torch.distributed over RCCL/xGMI).

The reference is single-process (no NCCL / torch.distributed call site anywhere), so nothing
here has a reference counterpart; it follows SURVEY.md section 8(e): envs are independent, so each rank owns a
contiguous slice of GLOBAL env ids (which key the random streams and the FpvMix thirds -- results do not depend on the
number of ranks), and one all-gather per step publishes the packed per-rank block [obs stack | reward | done | time-out].
The step kernel fills that block itself (taco_bind_gather_block), so a step is one kernel launch + one collective.

Layout of the gathered result: `world_size` slabs of `m = max shard size` rows each (the PADDED layout: every rank
contributes m rows, the first `hi - lo` of them live), i.e. rank r's envs are rows [r * m, r * m + size_r).  Equal
shards make that the global env order; ragged shards (numEnvs not a multiple of the world size) are exposed as views
(`GatheredBlocks.rank_rows`, `.global_rows()`), never re-copied.

Overlap (the gather of ~12-25 us is comparable to the step itself): `ShardedEnv.step_async` issues the
collective with async_op=True -- RCCL runs it on the process group's own stream, ordered behind the step kernel -- and
returns a handle; the next step's launch does not wait for it.  Two block / result buffers alternate, and a step that
is about to refill a block first waits (stream-side) for the gather that last read it.  An overlapped loop passes
`before_gather` = "wait for the previous step's gather": the wait then sits between this step's kernel launch and this
step's collective, so the kernel runs under the previous gather and at most one collective is outstanding when the next is issued.  `step_gathered` = step_async +
wait is the serial form a single learner that needs obs(t) before action(t+1) uses.
"""
import torch

from . import _lib


def shard_bounds(n_global, world_size, rank):
    """Contiguous slice [lo, hi) of global env ids owned by `rank`; the first n_global % world_size ranks get one more."""
    base, rem = divmod(int(n_global), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n_global, world_size):
    return [hi - lo for lo, hi in (shard_bounds(n_global, world_size, r) for r in range(world_size))]


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


def pack_block(obs, rew, done, timeout, rows=None):
    """Host-side packer with the kernel's block layout (used by the CPU tests, where no kernel fills the block).
    `rows` >= n allocates the padded block of the gather layout (rows beyond n stay zero)."""
    n = obs.shape[0]
    row = obs.shape[1] * 26
    blk = torch.zeros((rows or n, block_row(obs.shape[1])), dtype=torch.float32, device=obs.device)
    blk[:n, :row] = obs.reshape(n, -1)
    blk[:n, row] = rew.float()
    blk[:n, row + 1] = done.float()
    blk[:n, row + 2] = timeout.float()
    return blk


class GatheredBlocks:
    """The result of one all-gather in the padded layout [world_size * m, row]; `wait()` makes the CURRENT stream (and, for
    CPU backends, the host) wait for the collective."""

    def __init__(self, out, sizes, work=None):
        self.out, self.sizes, self.work = out, sizes, work
        self.m = out.shape[0] // len(sizes)

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self

    def rank_rows(self, r):
        """view of rank r's live rows"""
        return self.out[r * self.m: r * self.m + self.sizes[r]]

    def global_rows(self):
        """[n_global, row] in global env order: the buffer itself for equal shards (no copy), one index_select for ragged shards"""
        if len(set(self.sizes)) == 1:
            return self.out
        idx = torch.cat([torch.arange(r * self.m, r * self.m + s, device=self.out.device) for r, s in enumerate(self.sizes)])
        return self.out.index_select(0, idx)


def all_gather_blocks(block, n_global, world_size, group=None, out=None, async_op=False):
    """ONE collective: every rank's [m, row] block (m = the largest shard; a rank with fewer envs leaves its last row unused) ->
    GatheredBlocks over [world_size * m, row].  A block with exactly this rank's shard size is accepted too and padded here
    (tests, callers that do not pre-allocate)."""
    import torch.distributed as dist
    sizes = shard_sizes(n_global, world_size)
    m = max(sizes)
    if block.shape[0] != m:
        padded = torch.zeros((m, block.shape[1]), dtype=block.dtype, device=block.device)
        padded[: block.shape[0]] = block
        block = padded
    if out is None:
        out = torch.empty((world_size * m, block.shape[1]), dtype=block.dtype, device=block.device)
    work = dist.all_gather_into_tensor(out, block, group=group, async_op=async_op)
    return GatheredBlocks(out, sizes, work if async_op else None)


class ShardedEnv:
    """This rank's slice of a `numEnvs`-wide job on its own MI355X."""

    def __init__(self, cfg, rank, world_size, device, gather=True, group=None, collective_when_alone=False, direct=False):
        """direct: False (default) -- the per-step all-gather is the process group's all_gather_into_tensor; True (opt-in) -- when the group's backend
        is RCCL ("nccl") it is ONE ncclAllGather called directly on a comm stream (taco_amd/rccl.py), after a self-check every rank agrees on.
        Measured with the one rank a one-GPU box allows (profiles/r06_l_gather_host_cost.txt): 23.3 us of host time per overlapped step against
        29.4 through torch.distributed (5 without a collective).  Opt-in because its multi-rank initialisation has never run on real hardware
        (RCCL refuses two ranks on one device, and no multi-GPU node was available to the builder).  `direct_path` / `direct_reason` tell which one runs."""
        from .vec_env import FpvBase
        self.rank, self.world_size, self.group, self.gather = rank, world_size, group, gather
        # a 1-rank job needs no collective; collective_when_alone issues it anyway -- what lets a ONE-GPU box execute the RCCL branch for real
        # (RCCL refuses two ranks on one device): tests/test_api_gpu.py::test_rccl_backend_runs_the_gather_path_on_one_rank
        self.collective_when_alone = bool(collective_when_alone)
        self.n_global = int(cfg["env"]["numEnvs"])
        self.lo, self.hi = shard_bounds(self.n_global, world_size, rank)
        self.sizes = shard_sizes(self.n_global, world_size)
        self.m = max(self.sizes)
        self.env = FpvBase(cfg, rl_device=str(device), sim_device=str(device), env_offset=self.lo, num_envs_local=self.hi - self.lo,
                           copy_outputs=False, states_ring=True)   # (in place on the obs buffer; a state stack lives in the frame ring)
        self.len_obs = self.env.len_obs
        row = block_row(self.len_obs)
        # two block / result pairs alternate so that step t + 1 can fill its block while the gather of step t is still reading the other
        self.blocks = [torch.zeros((self.m, row), dtype=torch.float32, device=self.env.device) for _ in range(2)]
        self.outs = [torch.empty((world_size * self.m, row), dtype=torch.float32, device=self.env.device) for _ in range(2)]
        self.pending = [None, None]
        self.k = 0
        self.block = self.blocks[0]
        self._bound = None
        self._bind(self.block)
        self.direct, self.direct_reason = None, "not asked for"
        if direct and (world_size > 1 or self.collective_when_alone):
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                self.direct_reason = "no process group"
            elif dist.get_backend(group) != "nccl":
                self.direct_reason = f"the process group's backend is {dist.get_backend(group)}, not RCCL"   # (gloo rehearsals: several ranks share a device)
            else:
                from . import rccl
                self.direct, self.direct_reason = rccl.make_direct_comm(rank, world_size, self.env.device, group)

    @property
    def direct_path(self):
        return self.direct is not None

    def _bind(self, block):
        """the block the step kernel fills besides its outputs (None: none -- the launch a single-GPU VecTask.step() makes)"""
        ptr = block.data_ptr() if block is not None else None
        if ptr != self._bound:
            _lib.check(self.env.lib.taco_bind_gather_block(self.env._h, ptr))
            self._bound = ptr

    def step_local(self, local_actions):
        """This rank's slice through VecTask.step() with NO gather block bound -- exactly the launch a single-GPU env makes; for callers
        that need no global view of the step (a data-parallel learner: every rank acts on its own envs).  -> (obs dict, rew, done, extras)
        of the local envs."""
        self._bind(None)
        return self.env.step(local_actions)

    def step_async(self, local_actions, before_gather=None):
        """One step of this rank's envs + the all-gather of its block, issued without waiting for it: returns a GatheredBlocks whose
        `wait()` orders the current stream behind the collective.  (gather off / one rank: the local block, nothing pending)
        before_gather: called between the kernel launch and the collective -- where an overlapped loop waits for the PREVIOUS step's gather
        (the step kernel is already in flight under it; at most one collective of this env is outstanding when the next is issued)."""
        k = self.k
        if self.pending[k] is not None:  # the gather that last read this block / wrote this result must be done before both are reused
            self.pending[k].wait()
            self.pending[k] = None
        self.block = self.blocks[k]
        self._bind(self.block)
        self.env.step_raw(local_actions)
        self.k = 1 - k
        if before_gather is not None:
            before_gather()
        if not self.gather or (self.world_size == 1 and not self.collective_when_alone):
            return GatheredBlocks(self.block[: self.hi - self.lo], [self.hi - self.lo])
        if self.direct is not None:   # one ncclAllGather on the comm stream, ordered behind the step kernel by an event
            g = GatheredBlocks(self.outs[k], self.sizes, self.direct.all_gather(self.outs[k], self.block))
        else:
            g = all_gather_blocks(self.block, self.n_global, self.world_size, self.group, out=self.outs[k], async_op=True)
        self.pending[k] = g
        return g

    def step_gathered(self, local_actions):
        """step_async + wait: the gathered block of THIS step in global env order (or the local one if gather is off)."""
        g = self.step_async(local_actions).wait()
        self.gathered = g
        return g.global_rows()

    def drain(self):
        for k in range(2):
            if self.pending[k] is not None:
                self.pending[k].wait()
                self.pending[k] = None

    def step(self, local_actions):
        """-> (obs [N_global,len_obs,26], rew, done, time_outs) for the whole job, plus this rank's critic states."""
        blk = self.step_gathered(local_actions)
        obs, rew, done, tmo = unpack_block(blk, self.len_obs)
        return {"obs": obs, "states": self.env.states_buf}, rew, done, {"time_outs": tmo}
