#!/usr/bin/env python3
"""bench.py - env-steps/sec of the batched step()/reset() hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--workload C2|C3|C4|C5] [--mode step|rollout]

A "step" is ONE batched env step over the whole resident batch: the step kernel advances every env, resets the ones
whose episode ended (done, or horizon H = 100) and writes the new observation batch.  Inputs (state + int32 actions)
are resident in HBM when the timed region starts.  Default workload C2 = BASELINE.json configs[1]:
NovelGridworld-Pogostick-v1, 65 536 envs per GPU, 10x10 map.  N > 1: one process per GPU (torchrun), envs sharded by
global env index, no data-path collective (envs are independent) -> weak scaling.

The JSON line also carries
  roofline      algorithmic bytes per env-step (2*S*S + 12*K + 45) * envs per launch / the step
                kernel's average duration measured with HIP events on the kernel's own stream, vs 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (oracle/ngw_oracle.c, a port - the Python reference cannot travel to the GPU box)
                timed on this box's host cores on a bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)

POGO, BOW = 'NovelGridworld-Pogostick-v1', 'NovelGridworld-Bow-v1'
WORKLOADS = {   # name -> (env id, map size, novelty, envs per GPU, description)
    'C2': (POGO, 10, None, 65536, 'NovelGridworld-Pogostick-v1, 65536 envs/GPU, 10x10'),
    'C3': (BOW, 20, None, 65536, 'NovelGridworld-Bow-v1, 65536 envs/GPU, 20x20'),
    'C4': (POGO, 10, ('axe', 'medium', 'wooden', ''), 32768, "Pogostick-v1 + inject_novelty('axe','medium','wooden'), 32768 envs/GPU, 10x10"),
    'C5': (POGO, 32, ('additem', 'hard', 'arrow', ''), 65536, "Pogostick-v1 + inject_novelty('additem','hard','arrow'), 65536 envs/GPU, 32x32"),
}
HORIZON = 100                  # per-episode step cap of the reference's evaluation scripts (tests/test.py:30, enjoy.py:107)
ACTION_SEED = 1234


def algorithmic_bytes(S, K):
    return 2 * S * S + 12 * K + 45


def cpu_baseline(spec, budget_s=12.0):
    """Oracle timed on the host cores: same workload shape (uniform actions, autoreset, H = 100), bounded sample."""
    from oracle import ngw_oracle as orc
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:                                                    # the box's CPU SHARE (cgroup quota), not the host's thread count:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]     # more threads than that only oversubscribe
        if quota != 'max':
            cores = max(1, min(cores, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    out = {}
    for label, threads in (('1core', 1), ('allcores', cores)):
        used = orc.set_threads(threads)
        n = 4096 * max(1, used)
        o = orc.Oracle(spec.compile(), n, seed=0, autoreset=True, horizon=HORIZON)
        o.reset()
        o.rollout(5, ACTION_SEED, 0)                       # warm-up
        t0, steps, T = time.perf_counter(), 0, 500
        while True:                                        # about budget_s / 2 seconds of CPU work per leg
            o.rollout(T, ACTION_SEED, 5 + steps)
            steps += T
            dt = time.perf_counter() - t0
            if dt > budget_s / 2:
                break
        out[label] = dict(value=n * steps / dt, cores=used, sample='%d envs x %d steps in %.1f s' % (n, steps, dt))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1000)
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--workload', default='C2', choices=sorted(WORKLOADS))
    ap.add_argument('--mode', default='step', choices=['step', 'rollout'])
    ap.add_argument('--envs', type=int, default=0, help='envs per GPU (default: the workload\'s)')
    ap.add_argument('--launch', default='graph', choices=['graph', 'eager'],
                    help='step mode: replay the K step launches from one hipGraph (default) or launch them one by one')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-stagger', action='store_true', help='skip the staggered-episode-ends side measurement')
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                    help='gloo + --single-device: rehearse the N > 1 code path on a one-GPU box (every rank on cuda:0)')
    ap.add_argument('--single-device', action='store_true')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("--gpus %d needs one process per GPU: launch with python -m torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))   # RCCL on ROCm
        else:
            dist.init_process_group('gloo')

    from gym_novel_gridworlds_amd import VecNovelGridworld, apply_novelty, make_spec
    env_id, S, nov, n_default, desc = WORKLOADS[args.workload]
    n = args.envs or n_default
    spec = make_spec(env_id, S)
    if nov:
        apply_novelty(spec, *nov)
    K, A = len(spec.items_id), len(spec.actions_id)
    # headline: every env hits the horizon in the same step, where preparing next episodes ahead buys nothing (the refill
    # launch costs what the inline resets cost); the staggered side measurement below switches it on
    v = VecNovelGridworld(spec=spec, num_envs=n, device=local_rank, seed=0, autoreset=True, horizon=HORIZON,
                          env_index_base=rank * n, reset_prefetch=0)
    v.reset()
    steps, warmup = args.steps, args.warmup

    GRAPH_MAX = 2048                      # kernel nodes per graph; longer runs replay it (its action rows repeat)
    use_graph = args.mode == 'step' and args.launch == 'graph' and steps >= 2
    if args.mode == 'step':
        # i.i.d. uniform int32 actions over len(actions_id), seed 1234 (+rank), resident in HBM before the timed region
        g = torch.Generator(device='cuda')
        g.manual_seed(ACTION_SEED + rank)
        rows = warmup + (min(steps, GRAPH_MAX) if use_graph else min(steps, 1024))
        acts = torch.randint(0, A, (rows, n), dtype=torch.int32, device='cuda', generator=g)
        ptrs = [acts[i].data_ptr() for i in range(rows)]
        torch.cuda.synchronize()
    g_steps = 0
    if use_graph:
        g_steps = min(steps, GRAPH_MAX) & ~1
        v.graph_build(ptrs[warmup], n, g_steps)

    def run_eager(k, row0):
        for i in range(k):
            v.step_device(ptrs[warmup + (row0 + i) % (len(ptrs) - warmup)] if row0 >= 0 else ptrs[i])

    def run_timed():
        """exactly `steps` batched steps"""
        if args.mode == 'rollout':
            v.rollout(steps, ACTION_SEED, warmup)
        elif use_graph:
            v.graph_launch(steps // g_steps)
            run_eager(steps % g_steps, 0)
        else:
            run_eager(steps, 0)

    def fence():
        v.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if args.mode == 'rollout':
        if warmup > 0:
            v.rollout(warmup, ACTION_SEED, 0)
    else:
        run_eager(warmup, -1)
    fence()
    v.timing_begin()                      # HIP event pair on the kernel's own stream, around the timed launches
    t0 = time.perf_counter()
    run_timed()
    dev_ms = v.timing_end()
    fence()
    dt = time.perf_counter() - t0
    assert v.error_flags() == 0
    if world > 1:
        t = torch.tensor([dt, dev_ms], dtype=torch.float64, device='cuda' if args.dist_backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dev_ms = float(t[0].item()), float(t[1].item())

    # roofline: algorithmic bytes per launch / average launch duration (device time of the timed region / launches)
    launches = 1 if args.mode == 'rollout' else steps
    steps_per_launch = steps if args.mode == 'rollout' else 1
    launch_ms = dev_ms / launches
    B = algorithmic_bytes(S, K)
    achieved = B * n * steps_per_launch / (launch_ms * 1e-3) / 1e9
    roofline = {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': None, 'kernel': 'ngw_kernel',
                'kernel_ms_avg': round(launch_ms, 6), 'launches_timed': launches,
                'algorithmic_bytes_per_env_step': B, 'env_steps_per_launch': n * steps_per_launch,
                'timing': 'HIP event pair on the kernel stream around the timed launches (includes inter-launch gaps)'}
    pmc_file = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    pmc = json.load(open(pmc_file)) if os.path.exists(pmc_file) else {}

    def add_traffic(rf, mode, env_steps_per_launch):
        """HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc cannot run inside this process)."""
        rec = pmc.get('%s_%s' % (args.workload, mode))
        if rec and n == 65536 or (rec and args.workload == 'C4' and n == 32768):
            rf['traffic'] = round(rec['hbm_bytes_per_env_step'] * env_steps_per_launch)
            rf['traffic_source'] = rec['source']

    add_traffic(roofline, args.mode, n * steps_per_launch)

    # the other mode of the same workload, reported beside the headline (fused T-step rollout: SURVEY.md §8(d))
    fused = None
    if args.mode == 'step' and world == 1:
        if warmup > 0:
            v.rollout(warmup, ACTION_SEED, 10 ** 6)
        fence()
        v.timing_begin()
        t1 = time.perf_counter()
        v.rollout(steps, ACTION_SEED, 10 ** 6 + warmup)
        f_ms = v.timing_end()
        fence()
        f_dt = time.perf_counter() - t1
        f_ach = B * n * steps / (f_ms * 1e-3) / 1e9
        obs_bytes = S * S + 4 * K + 12 + 9            # what the fused kernel must write per env-step (state stays on chip)
        fused = {'value': round(n * steps / f_dt, 1), 'unit': 'env-steps/s', 'ms_per_step': round(f_dt / steps * 1e3, 6),
                 'what': 'ngw_rollout: all %d steps in ONE launch, uniform actions generated in-kernel, state kept in LDS, '
                         'observation batch written to HBM every step' % steps,
                 'roofline': {'bound': 'hbm', 'achieved': round(f_ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                              'frac': round(f_ach / HBM_PEAK_GBS, 4), 'kernel_ms': round(f_ms, 4),
                              'algorithmic_bytes_per_env_step': B, 'min_hbm_bytes_per_env_step': obs_bytes,
                              'frac_of_peak_on_min_bytes': round(obs_bytes * n * steps / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              'traffic': None}}
        add_traffic(fused['roofline'], 'rollout', n * steps)
        # the same fused launch taking the caller's action rows from HBM (ngw_rollout_actions) instead of the in-kernel policy
        rows_a = min(steps, len(ptrs) - warmup)
        if rows_a >= 2:
            v.rollout_actions(ptrs[warmup], n, rows_a)
            fence()
            v.timing_begin()
            v.rollout_actions(ptrs[warmup], n, rows_a)
            a_ms = v.timing_end()
            fence()
            fused['with_supplied_actions'] = {'value': round(n * rows_a / (a_ms * 1e-3), 1), 'unit': 'env-steps/s',
                                              'ms_per_step': round(a_ms / rows_a, 6), 'steps': rows_a,
                                              'what': 'ngw_rollout_actions: one launch, step t reads the [t, :] int32 action row resident in HBM'}
        assert v.error_flags() == 0

    # side measurement: the SAME workload with episode ends spread over the batch (step_count offset e * 7919 % H: about
    # n / H envs reset in every batched step, a few per wavefront) - the regime of a training loop; inline placement
    # loops vs prepared next episodes (ngw_set_reset_prefetch).  Not the headline: BASELINE's loop resets all envs together.
    stag = None
    if args.mode == 'step' and world == 1 and use_graph and not args.no_stagger:
        import numpy as np
        stag = {'what': 'episode ends staggered over the batch (~%d of %d envs reset per batched step), hipGraph replay' % (n // HORIZON, n)}
        for key, every in (('inline_resets', 0), ('prepared_next_episodes_every_32', 32)):
            v.set_reset_prefetch(every)
            v.reset()
            v.set_state(0, step_count=(np.arange(n) * 7919 % HORIZON).astype(np.int32))
            gs = max(2, min(64, len(ptrs) - warmup))          # the action rows resident in HBM bound the captured graph
            v.graph_build(ptrs[warmup], n, gs)
            v.graph_launch(max(1, 256 // gs))
            fence()
            v.timing_begin()
            v.graph_launch(max(1, 512 // gs))
            s_ms = v.timing_end() / (max(1, 512 // gs) * gs)
            fence()
            stag[key] = {'ms_per_step': round(s_ms, 6), 'value': round(n / (s_ms * 1e-3), 1), 'unit': 'env-steps/s'}
        v.set_reset_prefetch(0)
        assert v.error_flags() == 0

    if rank == 0:
        total = n * world * steps
        line = {
            'metric': 'env-steps/sec', 'value': round(total / dt, 1), 'unit': 'env-steps/s', 'n_gpus': world,
            'steps': steps, 'warmup': warmup, 'ms_per_step': round(dt / steps * 1e3, 6), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'int8/int32', 'data': 'synthetic',
            'config': {'workload': desc, 'name': args.workload, 'envs_per_gpu': n, 'global_envs': n * world,
                       'map_size': S, 'n_items': K, 'n_actions': A, 'horizon': HORIZON, 'autoreset': 'same-step', 'reset_prefetch': 0,
                       'mode': ('one launch per batched step (%s), actions in HBM' % ('hipGraph replay' if use_graph else 'eager')) if args.mode == 'step'
                               else 'fused rollout: all steps in one launch, actions generated in-kernel',
                       'parallelism': 'envs sharded x%d, no collective' % world},
            'roofline': roofline,
        }
        if fused:
            line['fused_rollout'] = fused
        if stag:
            line['staggered_resets'] = stag
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(spec)
            line['cpu_baseline'] = {'value': round(cb['allcores']['value'], 1), 'unit': 'env-steps/s',
                                    'cores': cb['allcores']['cores'], 'kind': 'port', 'sample': cb['allcores']['sample'],
                                    'one_core_value': round(cb['1core']['value'], 1)}
        print(json.dumps(line), flush=True)
    v.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
