"""The N > 1 path on CPU: two gloo ranks shard a batch, advance their shards with the ORACLE
(the HIP kernels need a GPU; here the checker stands in for them to exercise the host-side
sharding + reduction logic), and all-reduce the episodic-return record.  The sharded result
must equal the single-process result: the noise is keyed by the global env index."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rollout_record(offset, count, seed, T, Tmax):
    """Oracle rollout of envs [offset, offset+count): fishing-v1 f32, random policy, auto-reset."""
    sys.path.insert(0, ROOT)
    from oracle import fishing_oracle as fo
    env = np.arange(offset, offset + count, dtype=np.uint64)
    obs = fo.reset_obs(fo.MODEL_V1, 0.75, np.full(count, 1.0, np.float32), np.float32)
    t = np.zeros(count, np.int32)
    ep = np.zeros(count, np.float32)
    rec = np.zeros(4)
    for s in range(T):
        a = fo.policy_random_action(fo.MODEL_V1, seed, env, s)
        z = fo.noise_normal(seed, env, s)
        o, r, d, t2, _ = fo.step(fo.MODEL_V1, obs, t, a, z, 0.3, 1.0, 0.1, Tmax=Tmax, dtype=np.float32)
        ep = (ep + r).astype(np.float32)
        m = d.astype(bool)
        rec += [ep[m].astype(np.float64).sum(), (ep[m].astype(np.float64) ** 2).sum(), m.sum(), t2[m].sum()]
        ep = np.where(m, np.float32(0), ep)
        obs, t, _, _ = fo.auto_reset(fo.MODEL_V1, o, d, t2, 1.0, 0.3, 0.75, dtype=np.float32)
    return rec, obs


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from gym_fishing_amd import sharding
    assert sharding.dist_info()[:2] == (rank, world)
    off, cnt = sharding.shard_range(total, rank, world)
    rec, obs = _rollout_record(off, cnt, seed=77, T=12, Tmax=4)
    t = torch.tensor(rec, dtype=torch.float64)
    sharding.all_reduce_record(t)
    gathered = [torch.zeros(total, dtype=torch.float32) for _ in range(world)] if rank == 0 else None
    pad = torch.zeros(total, dtype=torch.float32)
    pad[off:off + cnt] = torch.from_numpy(obs)
    dist.gather(pad, gathered, dst=0)
    if rank == 0:
        q.put((t.numpy().copy(), sum(gathered).numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo_sharded_rollout_matches_single_process():
    total, world = 1030, 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    rec2, obs2 = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rec1, obs1 = _rollout_record(0, total, seed=77, T=12, Tmax=4)
    assert rec2[2] == rec1[2] and rec2[3] == rec1[3] and rec1[2] > 0
    assert np.allclose(rec2[:2], rec1[:2], rtol=1e-12)
    assert np.array_equal(obs2, obs1)          # per-env trajectories identical for any world size
    from gym_fishing_amd import sharding
    s = sharding.summarize_record(torch.tensor(rec2))
    assert 0 < s["mean_length"] <= 5
