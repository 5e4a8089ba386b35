"""N > 1 path on CPU: two (and eight) gloo ranks, envs sharded by global index, observation gather / all_gather == one unsharded batch,
also after every rank has called inject_novelty() on its shard.  The local envs are the oracle-backed stand-in of
tests/ngw_testlib.py (a subclass of the product class with its device hooks replaced); the product path is
VecNovelGridworld on each GPU (tests/test_multi_gpu_rehearsal.py runs that on the GPU box)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ngw_testlib as T
from gym_novel_gridworlds_amd.dist import shard_range

N, STEPS = 96, 40


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, cfg, q, inject):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from gym_novel_gridworlds_amd import inject_novelty
        env_id, S, nov = T.CFGS[cfg]
        if inject:                                       # the shard is built plain; every rank injects the novelty on its own shard
            from gym_novel_gridworlds_amd import make_spec
            env = T.oracle_sharded(global_num_envs=N, spec=make_spec(env_id, S), seed=3, autoreset=True, horizon=12)
            first = env.first
            same = inject_novelty(env, *nov)
            assert same is env and env.first == first and env.local.o.base == first      # same object, same global env indices
        else:
            env = T.oracle_sharded(global_num_envs=N, spec=T.build_spec(cfg), seed=3, autoreset=True, horizon=12)
        spec = env.spec
        assert (env.first, env.num_envs) == shard_range(N, world, rank)
        env.reset()
        rs = np.random.RandomState(0)
        for t in range(STEPS):
            a = rs.randint(0, len(spec.actions_id), size=N).astype(np.int32)      # same global action batch on every rank
            env.step(a[env.first:env.first + env.num_envs])
        got = env.gather_observation(dst=0)
        every = env.all_gather_observation()                                       # ... and the same stack on every rank
        if rank == 0:
            assert all((got[k] == every[k]).all() for k in got)
            q.put({k: v.numpy() for k, v in got.items()})
        else:
            assert got is None
            q.put({k: v.numpy() for k, v in every.items()})
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('cfg,inject,world', [('pogo10', False, 2), ('axe10', False, 2), ('axe10', True, 2), ('add12m', True, 2),
                                              ('axe10', True, 8)])     # world 8: the node's rank count (BASELINE configs 4-5), 12 envs per rank
def test_sharding_and_gather_match_single_batch(cfg, inject, world):
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cfg, q, inject)) for r in range(world)]
    for p in procs:
        p.start()
    gots = [q.get() for _ in range(world)]
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    spec = T.build_spec(cfg)
    ref = T.OracleVec(spec, N, seed=3, autoreset=True, horizon=12)
    ref.reset()
    rs = np.random.RandomState(0)
    for t in range(STEPS):
        ref.step(rs.randint(0, len(spec.actions_id), size=N).astype(np.int32))
    st = ref.o.st
    S = spec.map_size
    for got in gots:                                     # rank 0's gather and every other rank's all_gather
        assert (got['map'] == st.map.reshape(N, S, S)).all() and (got['agent_location'] == st.loc).all()
        assert (got['agent_facing_id'] == st.facing).all() and (got['inventory_items_quantity'] == st.inv).all()
        assert (got['reward'] == ref.o.reward).all() and (got['done'] == ref.o.done.astype(bool)).all()
        assert (got['info'].view(np.uint32) == ref.o.info).all()
    assert st.episode.max() >= 3
    assert len({tuple(st.loc[i]) for i in range(N)}) > 8                  # (ranks that all restarted at global env 0 would repeat each other)
    assert not (st.map[:N // 2] == st.map[N // 2:]).all()


def test_shard_range_requires_even_split():
    assert shard_range(65536, 8, 3) == (3 * 8192, 8192)
    with pytest.raises(ValueError):
        shard_range(10, 4, 0)
