"""The N > 1 path on CPU: two (and eight) `gloo` ranks, each stepping its contiguous slice of global env ids and publishing it with
ONE all-gather of the packed block.  There is no GPU here, so each rank's producer is the CPU oracle (a checker used as a
stand-in data source inside tests/ only); what is under test is taco_amd.dist: shard bounds, block layout, the
collective, and that the gathered result equals a single-process run of all envs (results independent of world size)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist_
import torch.multiprocessing as mp

from taco_amd import config, dist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_global, idx, steps, q):
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist_.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = config.baseline_config(idx, num_envs=n_global)
        lo, hi = dist.shard_bounds(n_global, world, rank)
        flat = config.flat_cfg(cfg, env_offset=lo, num_envs_local=hi - lo)
        env = O.OracleEnv(flat)
        rng = np.random.default_rng(5)
        outs = []
        for t in range(steps):
            a = np.clip(0.3 * rng.standard_normal((n_global, 4)), -1, 1).astype(np.float32)  # same stream on every rank
            obs, st, rew, done, tmo = env.step(a[lo:hi])
            m = max(dist.shard_sizes(n_global, world))
            blk = dist.pack_block(torch.from_numpy(obs.copy()), torch.from_numpy(rew.copy()), torch.from_numpy(done.copy()), torch.from_numpy(tmo.copy()),
                                  rows=m if t % 2 else None)   # pre-padded block (what ShardedEnv binds) and exact-size block alike
            g = dist.all_gather_blocks(blk, n_global, world, async_op=bool(t % 3))   # the async form (handle + wait) and the blocking one
            g.wait()
            for r in range(world):   # padded layout: rank r's live rows are a VIEW of the result
                lo_r, hi_r = dist.shard_bounds(n_global, world, r)
                assert g.rank_rows(r).shape[0] == hi_r - lo_r and g.rank_rows(r).data_ptr() == g.out[r * g.m:].data_ptr()
            full = g.global_rows()
            if len(set(g.sizes)) == 1:
                assert full.data_ptr() == g.out.data_ptr()   # equal shards: no copy at all
            outs.append(full.numpy().copy())
        if rank == 0:
            q.put(np.stack(outs))
    finally:
        dist_.destroy_process_group()


# equal shards; ragged shards + mix thirds + len_obs 1; and the rank count the driver's scaling run ends with: EIGHT ranks, ragged (75 = 3 x 10 + 5 x 9
# envs: the mix thirds 25 / 50 fall inside ranks 2 and 5)
@pytest.mark.parametrize("n_global,idx,world", [(64, 1, 2), (75, 4, 2), (75, 4, 8)])
def test_gather_over_gloo_ranks_equals_single_process(n_global, idx, world):
    from oracle import oracle as O
    steps = 25 if world == 2 else 12
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_global, idx, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cfg = config.baseline_config(idx, num_envs=n_global)
    env = O.OracleEnv(config.flat_cfg(cfg))
    rng = np.random.default_rng(5)
    for t in range(steps):
        a = np.clip(0.3 * rng.standard_normal((n_global, 4)), -1, 1).astype(np.float32)
        obs, st, rew, done, tmo = env.step(a)
        o, r, d, tm = dist.unpack_block(torch.from_numpy(got[t]), 1)
        np.testing.assert_array_equal(o.numpy(), obs)
        np.testing.assert_array_equal(r.numpy(), rew)
        np.testing.assert_array_equal(d.numpy(), done)
        np.testing.assert_array_equal(tm.numpy(), tmo != 0)
