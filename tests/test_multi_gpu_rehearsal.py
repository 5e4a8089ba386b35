"""The N > 1 path on real hardware, rehearsed on ONE GPU: two ranks (gloo rendezvous on 127.0.0.1, both on cuda:0), envs
sharded by global index, HIP step kernels on each rank, the observation stack through ngw_pack_obs -> gather ->
ngw_unpack_obs, checked against one unsharded oracle batch; and `python bench.py --gpus 2` started as the driver starts
it (plain python: it launches its own ranks)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import ngw_testlib as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, STEPS = 192, 45


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, cfg, q, inject):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    import torch
    import torch.distributed as dist
    from gym_novel_gridworlds_amd import inject_novelty, make_spec
    from gym_novel_gridworlds_amd.dist import ShardedVecNovelGridworld, shard_range
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        env_id, S, nov = T.CFGS[cfg]
        if inject:                                       # every rank injects the novelty on its own shard, lidar and prepared episodes set before
            env = ShardedVecNovelGridworld(global_num_envs=N, spec=make_spec(env_id, S), seed=3, autoreset=True, horizon=12, device=0,
                                           reset_prefetch=5, reset_prefetch_depth=2)
            env.local.lidar_configure(num_beams=4, fused=True)
            same = inject_novelty(env, *nov)
            assert same is env and env.local.env_index_base == env.first == shard_range(N, world, rank)[0]
            assert env.local.reset_prefetch == 5 and env.local.reset_prefetch_depth == 2 and env.local.lidar_fused
        else:
            env = ShardedVecNovelGridworld(global_num_envs=N, spec=T.build_spec(cfg), seed=3, autoreset=True, horizon=12, device=0)
        spec = env.spec
        assert (env.first, env.num_envs) == shard_range(N, world, rank)
        env.reset()
        rs = np.random.RandomState(0)
        for t in range(STEPS):
            a = rs.randint(0, len(spec.actions_id), size=N).astype(np.int32)      # same global action batch on every rank
            env.step(a[env.first:env.first + env.num_envs])
        got = env.gather_observation(dst=0)
        got2 = env.gather_observation(dst=0)                                       # buffers are reused: same answer
        every = env.all_gather_observation()
        if rank == 0:
            assert all((got[k] == got2[k]).all() for k in got) and all((got[k] == every[k]).all() for k in got)
            q.put({k: v.cpu().numpy() for k, v in got.items()})
        else:
            assert got is None
            q.put({k: v.cpu().numpy() for k, v in every.items()})
        dist.barrier()
        env.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('cfg,inject', [('pogo10', False), ('axe10', False), ('bow20', False), ('axe10', True), ('add12m', True)])
def test_two_ranks_on_one_gpu_match_one_oracle_batch(cfg, inject):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, cfg, q, inject)) for r in range(2)]
    for p in procs:
        p.start()
    gots = [q.get(), q.get()]
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
    for got in gots:                                     # rank 0's gather and rank 1's all_gather
        assert (got['map'] == st.map.reshape(N, S, S)).all() and (got['agent_location'] == st.loc).all()
        assert (got['agent_facing_id'] == st.facing).all() and (got['inventory_items_quantity'] == st.inv).all()
        assert (got['reward'] == ref.o.reward).all() and (got['done'] == ref.o.done.astype(bool)).all()
        assert (got['info'].view(np.uint32) == ref.o.info).all()
    assert st.episode.max() >= 3


def _rccl_worker(port, q):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    os.environ['RANK'], os.environ['WORLD_SIZE'], os.environ['LOCAL_RANK'] = '0', '1', '0'
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch
    from gym_novel_gridworlds_amd.dist import ShardedVecNovelGridworld, init_process_group
    torch.cuda.set_device(0)
    dist = init_process_group('nccl', 0)                       # RCCL communicator + probe all_reduce, loud on failure
    try:
        spec = T.build_spec('axe10')
        env = ShardedVecNovelGridworld(global_num_envs=N, spec=spec, seed=3, autoreset=True, horizon=12, device=0, exchange_always=True)
        env.reset()
        rs = np.random.RandomState(0)
        for t in range(STEPS):
            env.step(rs.randint(0, len(spec.actions_id), size=N).astype(np.int32))
        got = env.gather_observation(dst=0)                    # torch.distributed.gather of a DEVICE payload over RCCL
        every = env.all_gather_observation()                   # all_gather_into_tensor, same
        torch.cuda.synchronize()
        assert all((got[k] == every[k]).all() for k in got)
        q.put({k: v.cpu().numpy() for k, v in got.items()})
        env.close()
    finally:
        dist.destroy_process_group()


def test_one_rank_rccl_group_runs_the_device_collectives():
    """The RCCL branch of the observation stack on the one GPU a box has: a one-rank `nccl` process group (torch's nccl backend IS
    RCCL on ROCm), device payloads through torch.distributed.gather / all_gather_into_tensor, pack and unpack launches ordered
    against torch's stream by events - the same calls N ranks make, minus the xGMI hop.  (Two RCCL ranks cannot share one GPU.)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    p.join(240)
    assert p.exitcode == 0
    got = q.get()
    spec = T.build_spec('axe10')
    ref = T.OracleVec(spec, N, seed=3, autoreset=True, horizon=12)
    ref.reset()
    rs = np.random.RandomState(0)
    for t in range(STEPS):
        ref.step(rs.randint(0, len(spec.actions_id), size=N).astype(np.int32))
    st = ref.o.st
    assert (got['map'] == st.map.reshape(N, spec.map_size, spec.map_size)).all() and (got['agent_location'] == st.loc).all()
    assert (got['inventory_items_quantity'] == st.inv).all() and (got['reward'] == ref.o.reward).all()
    assert (got['done'] == ref.o.done.astype(bool)).all() and (got['info'].view(np.uint32) == ref.o.info).all()


def test_world_8_unpack_on_one_gpu_matches_one_oracle_batch():
    """The root side of an EIGHT-rank gather on the one GPU a box has (no 8-GPU node reaches this build): eight shards of one global
    batch (BASELINE config 4's split: axe-medium, 8 x 4 096 envs, shard r = global envs [4096 r, 4096 r + 4096)) are stepped by eight
    handles, every shard packs its payload with ngw_pack_obs one after another into ONE buffer - what the gather delivers to the
    root - and a single ngw_unpack_obs(world = 8) (56 regions in one launch) scatters them into global arrays, compared with one
    unsharded oracle batch of 32 768 envs."""
    import torch
    from gym_novel_gridworlds_amd import VecNovelGridworld
    from gym_novel_gridworlds_amd.dist import shard_range
    spec = T.build_spec('axe10')
    world, n, H, steps = 8, 4096, 12, 30
    Ng, S, K, A = world * n, spec.map_size, len(spec.items_id), len(spec.actions_id)
    shards = [VecNovelGridworld(spec=spec, num_envs=n, seed=3, autoreset=True, horizon=H, device=0, env_index_base=shard_range(Ng, world, r)[0])
              for r in range(world)]
    ref = T.OracleVec(spec, Ng, seed=3, autoreset=True, horizon=H)
    ref.reset()
    for v in shards:
        v.reset()
    rs = np.random.RandomState(5)
    for t in range(steps):
        a = rs.randint(0, A, size=Ng).astype(np.int32)
        ref.step(a)
        for r, v in enumerate(shards):
            v.step(a[r * n:(r + 1) * n])
    offs = shards[0].pack_layout()
    assert all(v.pack_layout() == offs for v in shards)                 # equal shards: equal payloads
    payloads = torch.zeros(world * offs[7], dtype=torch.uint8, device='cuda:0')
    for r, v in enumerate(shards):
        v.pack_obs(payloads.data_ptr() + r * offs[7])
        v.sync()
    dev = 'cuda:0'
    out = {'map': torch.zeros((Ng, S, S), dtype=torch.int8, device=dev), 'agent_location': torch.zeros((Ng, 2), dtype=torch.int32, device=dev),
           'agent_facing_id': torch.zeros(Ng, dtype=torch.int32, device=dev), 'inventory_items_quantity': torch.zeros((Ng, K), dtype=torch.int32, device=dev),
           'reward': torch.zeros(Ng, dtype=torch.int32, device=dev), 'done': torch.zeros(Ng, dtype=torch.uint8, device=dev),
           'info': torch.zeros(Ng, dtype=torch.int32, device=dev)}
    torch.cuda.synchronize()
    root = shards[0]
    root.unpack_obs(payloads.data_ptr(), world, [out[k].data_ptr() for k in ('map', 'agent_location', 'agent_facing_id', 'inventory_items_quantity', 'reward', 'done', 'info')])
    root.sync()
    st = ref.o.st
    got = {k: v.cpu().numpy() for k, v in out.items()}
    assert (got['map'] == st.map.reshape(Ng, S, S)).all() and (got['agent_location'] == st.loc).all()
    assert (got['agent_facing_id'] == st.facing).all() and (got['inventory_items_quantity'] == st.inv).all()
    assert (got['reward'] == ref.o.reward).all() and (got['done'] == ref.o.done).all()
    assert (got['info'].view(np.uint32) == ref.o.info).all()
    assert st.episode.max() >= 2
    for v in shards:
        v.close()


def test_dist_module_has_no_device_wide_synchronisation():
    src = open(os.path.join(ROOT, 'gym_novel_gridworlds_amd', 'dist.py')).read()
    assert 'cuda.synchronize' not in src and '.sync()' not in src.split('# ------------------------------------------------------------------ the one collective')[1]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` outside torchrun: the parent spawns the ranks before it touches a GPU; rank 0's line
    reports both ranks, resets inside the timed region and the gather leg."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dist-backend', 'gloo', '--single-device', '--workload', 'C4',
           '--steps', '40', '--warmup', '5', '--no-cpu-baseline', '--no-side']
    env = dict(os.environ)
    env.pop('RANK', None)
    env.pop('WORLD_SIZE', None)
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['config']['global_envs'] == 2 * 32768
    assert line['resets_in_timed_region'] >= 1
    assert line['gather']['ranks'] == 2 and line['gather']['ms'] > 0 and line['gather']['payload_bytes_per_rank'] >= 32768 * (152 + 9)
    assert line['value'] > 0 and 0 < line['roofline']['frac'] < 1
