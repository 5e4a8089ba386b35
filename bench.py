#!/usr/bin/env python3
"""bench.py - env-steps/sec of the batched step()/reset() hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--workload C2|C3|C4|C5] [--mode step|rollout]

A "step" is ONE batched env step over the whole resident batch: the step kernel advances every env, resets the ones
whose episode ended (done, or horizon H = 100) and writes the new observation batch.  Inputs (state + int32 actions)
are resident in HBM when the timed region starts.  Default workload C2 = BASELINE.json configs[1]:
NovelGridworld-Pogostick-v1, 65 536 envs per GPU, 10x10 map.

N > 1: one process per GPU, envs sharded by global env index, no data-path collective (envs are independent) -> weak
scaling.  Called as plain `python bench.py --gpus N` this script STARTS ITS OWN RANKS (python -m torch.distributed.run,
rendezvous on 127.0.0.1) before it touches a GPU and relays rank 0's JSON line; launched under torchrun by someone
else it reads RANK / LOCAL_RANK / WORLD_SIZE from the environment.

Timed region: barrier + torch.cuda.synchronize(), clock, K steps, torch.cuda.synchronize(), clock, barrier; `value` uses the MAX
over ranks of that span (= until the slowest rank had finished its K steps).

The JSON line carries, besides the contract's fields,
  roofline        algorithmic bytes per env-step (2*S*S + 12*K + 45) * envs per launch / the step kernel's
                  average launch duration, measured with a HIP event pair on the kernel's own stream, vs 8 TB/s HBM peak
                  (`frac`), and the same on the wall-clock ms_per_step the driver can check (`frac_wall`)
  repeats         the K-step region run R = 5 more times after the contract's one: median / min / max (device and wall)
  cold_region     the same W + K steps BEFORE the device-clock warm-up (a fresh process finds the GPU in a low power state)
  lidar           the step with the fused LidarInFront observation (what the reference's scripts train on), per row format
  api_mode_lidar  LidarInFront(venv, dtype='packed', copy=False).step() host loop (PCIe-inclusive; the wrapper's opt-in fast path)
  resets_in_timed_region   the timed launches always contain auto-reset work: when K < H the episodes are started so
                  that every env reaches the horizon in the middle of the timed region
  gather          (N > 1) the one collective of the path: per-rank pack launch + torch.distributed.gather of the packed
                  observation (RCCL over xGMI) + unpack launch on rank 0, timed separately from the steps
  fused_rollout   the T-steps-per-launch mode (issue-bound, not an HBM figure), api_mode (PCIe-inclusive host loop),
                  c1_single_env (BASELINE configs[0]: the gym.Env adapter), staggered_resets
  cpu_baseline    the CPU oracle (oracle/ngw_oracle.c, a port - the Python reference cannot travel to the GPU box)
                  timed on this box's host cores on a bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import socket
import subprocess
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
    # tuning cases (not BASELINE.json configs): the wrapper predicates of the step - FireWall, FenceRestriction, Crate
    'X1': (POGO, 10, ('firewall', 'hard', '', ''), 65536, "Pogostick-v1 + inject_novelty('firewall','hard'), 65536 envs/GPU, 10x10"),
    'X2': (POGO, 10, ('fencerestriction', 'hard', 'oak', ''), 65536, "Pogostick-v1 + inject_novelty('fencerestriction','hard','oak'), 65536 envs/GPU, 10x10"),
    'X3': (POGO, 10, ('crate', 'hard', '', ''), 65536, "Pogostick-v1 + inject_novelty('crate','hard'), 65536 envs/GPU, 10x10"),
}
HORIZON = 100                  # per-episode step cap of the reference's evaluation scripts (tests/test.py:30, enjoy.py:107)
ACTION_SEED = 1234


def algorithmic_bytes(S, K):
    """what a read-pack-write design moves per env-step."""
    return 2 * S * S + 12 * K + 45


def design_bytes(S, K, map_in_place):
    """What THIS design has to move per env-step: the observation buffers are the state, updated in place.
    Staged kernel: the map row, the inventory row and 29 B of pose / action / counters are read, ~30 B are written through
    (outputs, pose, the changed cell / slots).  In-place kernel: of the map only the lines a step looks at come in - the block in
    front, its four neighbours, two cells ahead: three map rows = three 32-byte sectors - plus the same scalars, inventory row
    and write-through.  Which kernel ran is asked of the handle (ngw_step_kernel_info), not re-derived here."""
    return (96 if map_in_place else S * S) + 4 * K + 29 + 30


def cpu_share():
    """Host threads this process may really use: affinity, capped by the cgroup CPU quota (more only oversubscribes)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            cores = max(1, min(cores, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(spec, budget_s=12.0):
    """Oracle timed on the host cores: same workload shape (uniform actions, autoreset, H = 100), bounded sample."""
    from oracle import ngw_oracle as orc
    cores = cpu_share()
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


def c1_single_env(budget_s=2.0):
    """BASELINE configs[0]: the gym.Env adapter (N = 1) in the shape of the reference's tests/random_action.py:51-64 loop
    (reset every 10 steps with map_size in [10, 20)), without render / print, plus a steady-state variant (reset every 100)."""
    import numpy as np
    import gym_novel_gridworlds_amd as G
    out = {}
    for label, every, resize in (('random_action_loop', 10, True), ('steady_state', 100, False)):
        env = G.make(POGO)
        env.reset()
        rs = np.random.RandomState(0)
        A = env.action_space.n
        for _ in range(30):
            env.step(int(rs.randint(A)))
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget_s / 2:
            for i in range(50):
                env.step(int(rs.randint(A)))
                n += 1
                if (i + 1) % every == 0:
                    if resize:
                        env.map_size = int(rs.randint(10, 20))
                    env.reset()
        dt = time.perf_counter() - t0
        out[label] = {'value': round(n / dt, 1), 'unit': 'env-steps/s', 'us_per_step': round(dt / n * 1e6, 2), 'steps': n}
        env.close()
    return out


def self_launch(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as CHILD processes (this process has not touched the
    GPU and never will), relay their output, exit with their code."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL / device-tensor sharing across processes
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1000)
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--workload', default='C2', choices=sorted(WORKLOADS))
    ap.add_argument('--mode', default='step', choices=['step', 'rollout'])
    ap.add_argument('--envs', type=int, default=0, help='envs per GPU (default: the workload\'s)')
    ap.add_argument('--launch', default='graph', choices=['graph', 'eager'],
                    help='step mode: replay the K step launches from one hipGraph (default from 100 steps on) or launch them one by one')
    ap.add_argument('--adapt-steps', type=int, default=-1,
                    help='untimed eager steps before everything else, so that the handle has adapted its prepared-episode depth / refill '
                         'cadence to the workload before the graph is captured (default: 3000 for the high-churn tuning cases X*, else 0)')
    ap.add_argument('--clock-warm-ms', type=float, default=250.0,
                    help='keep the GPU busy this long on a SCRATCH handle before anything is measured (a fresh process finds the device in a '
                         'low power state, and 5 warm-up steps are 25 us); 0 = off')
    ap.add_argument('--lidar', default='', choices=['', 'int32', 'int16', 'packed'],
                    help='run the MAIN timed region with the fused LidarInFront observation in this row format (profiling runs; the default line reports it in `lidar`)')
    ap.add_argument('--repeats', type=int, default=5, help='run the K-step region this many more times after the contract region (median / spread in `repeats`)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-stagger', action='store_true', help='skip the staggered-episode-ends side measurement')
    ap.add_argument('--no-side', action='store_true', help='skip every side measurement (fused rollout, API mode, C1, stagger): tuning runs')
    ap.add_argument('--reset-prefetch', default='auto', help="prepared next episodes: 'auto' (the library default), 0 = off, N = refill cadence")
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                    help='gloo + --single-device: rehearse the N > 1 code path on a one-GPU box (every rank on cuda:0)')
    ap.add_argument('--single-device', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus > 1 and 'RANK' not in os.environ:
        self_launch(args)                                   # does not return
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        # RCCL (torch's `nccl` backend on ROCm) with a probe collective; a start-up failure is re-raised with the rank / device map
        from gym_novel_gridworlds_amd.dist import init_process_group
        init_process_group(args.dist_backend, local_rank)

    from gym_novel_gridworlds_amd import apply_novelty, make_spec
    from gym_novel_gridworlds_amd.dist import ShardedVecNovelGridworld
    env_id, S, nov, n_default, desc = WORKLOADS[args.workload]
    n = args.envs or n_default
    spec = make_spec(env_id, S)
    if nov:
        np.random.seed(0)                                 # (Crate draws its contents from the global stream at injection, like the reference)
        apply_novelty(spec, *nov)
    K, A = len(spec.items_id), len(spec.actions_id)
    prefetch = args.reset_prefetch if args.reset_prefetch == 'auto' else int(args.reset_prefetch)
    sv = ShardedVecNovelGridworld(spec=spec, global_num_envs=n * world, seed=0, autoreset=True, horizon=HORIZON,
                                  device=local_rank, reset_prefetch=prefetch)
    v = sv.local                                          # this rank's envs [rank * n, (rank + 1) * n)
    if args.lidar:
        v.lidar_configure(num_beams=8, fused=True, dtype={'int32': np.int32, 'int16': np.int16}.get(args.lidar, 'packed'))
    steps, warmup = args.steps, args.warmup

    def start_episodes():
        """reset(), then start every episode so that the timed region holds auto-reset work whatever K is: with K < H the
        horizon falls in the middle of the timed launches (synchronized over the batch: BASELINE's loop resets all envs together)."""
        v.reset()
        hit = warmup + max(1, steps // 2)                  # 1-based batched step at which step_count reaches H
        s0 = (HORIZON - hit) % HORIZON if steps < HORIZON else 0
        if s0:
            v.set_state(0, step_count=np.full(n, s0, np.int32))

    def episode0():
        return int(v.get_state(0, 1)['episode'][0])

    cold = None
    if args.clock_warm_ms > 0:
        # A scratch handle of the same shape, never the measured one: (i) `cold_region` - the contract's shape (W warm-up + K timed eager
        # steps) on the device as this process found it (nothing has run on it yet); (ii) the device-clock warm-up - fused rollouts until
        # the time is up.  The measured handle `v` is not touched by either: its W warm-up steps and K timed steps follow as the contract says.
        from gym_novel_gridworlds_amd import VecNovelGridworld
        scratch = VecNovelGridworld(spec=spec, num_envs=n, seed=99, autoreset=True, horizon=HORIZON, device=local_rank)
        if args.mode == 'step' and not args.no_side:
            g0 = torch.Generator(device='cuda')
            g0.manual_seed(ACTION_SEED + 31 + rank)
            kc = min(steps, 200)
            acts0 = torch.randint(0, A, (warmup + kc, n), dtype=torch.int32, device='cuda', generator=g0)
            torch.cuda.synchronize()
            scratch.reset()
            if warmup:
                scratch.step_device_many(acts0.data_ptr(), n, warmup)
            scratch.sync(); torch.cuda.synchronize()
            t0c = time.perf_counter()
            scratch.step_device_many(acts0[warmup].data_ptr(), n, kc)
            torch.cuda.synchronize()
            dtc = time.perf_counter() - t0c
            cold = {'ms_per_step': round(dtc / kc * 1e3, 6), 'value': round(n * kc / dtc, 1), 'steps': kc, 'warmup': warmup,
                    'what': 'the first W + K eager steps this process ran (on a scratch handle of the same shape), before the %g ms device-clock warm-up '
                            'that precedes the contract region' % args.clock_warm_ms}
            del acts0
        scratch.reset()
        t_end = time.perf_counter() + args.clock_warm_ms * 1e-3
        while time.perf_counter() < t_end:
            scratch.rollout(200, ACTION_SEED, 0)
            scratch.sync()
        scratch.close()
    adapt_steps = args.adapt_steps if args.adapt_steps >= 0 else (3000 if args.workload.startswith('X') else 0)
    if adapt_steps and args.mode == 'step':
        # a live loop adapts by itself (ngw_abi.cpp adapt_cadence: deeper prepared episodes, then more frequent refills, when envs end
        # episodes faster than refills come round); the timed region below is a captured graph, so the handle settles first
        g0 = torch.Generator(device='cuda')
        g0.manual_seed(ACTION_SEED + 7919 + rank)
        acts0 = torch.randint(0, A, (100, n), dtype=torch.int32, device='cuda', generator=g0)
        torch.cuda.synchronize()
        v.reset()
        for _ in range((adapt_steps + 99) // 100):
            v.step_device_many(acts0.data_ptr(), n, 100)
            v.sync()                                       # (the host reads the refills' reports between calls)
        del acts0
    start_episodes()
    GRAPH_MAX = 2048                      # kernel nodes per graph; longer runs replay it (its action rows repeat)
    # ONE graph launch enqueues the whole region (15 us of host time for 20 launches; an eager launch costs ~3.3 us of host time),
    # but the FIRST launch of an instantiated graph costs ~70 us more than a later one (20 steps fence to fence: 187 us against
    # 110 us eager), so regions shorter than 100 launches are issued eagerly, from one C-ABI call (tools/short_run.py).
    # (Tried: event-record nodes inside the graph to time a replay without its launch latency - two such nodes cost ~80 us per
    #  replay on this runtime, more than they explain.)
    # Round 5: short graphs are captured without a closing refill (ngw_graph_build's "open" graphs: the library keeps the refill cadence between replays)
    # and the threshold came down to 16 steps.  A 20-step replay costs ONE host call (13 us) and 85-92 us fence to fence (the first replay of an
    # instantiated graph 92-121 us: that is the contract region, `value_contract`); the eager loop is bound by the HOST's launch rate, which differs
    # from box to box of the pool - 2.7 / 3.7 / 4.3 us per empty-kernel launch seen in this round's runs - and gave 10.7 / 13.5 / 14.2 G in three
    # driver-form lines where the graph form gave 13.1 - 14.7 G.  Even a host that launches in 2.5 us on average jitters: twelve eager regions in a row
    # ran at 3.5 - 5.5 us per step on the device's own clock (the device waiting for launches), twelve replays at 3.86 (+ the regions that hold the
    # all-env reset: 5.4): `repeats.regions_in_order`.  `value` is the median of the contract region and the repeats either way.
    use_graph = args.mode == 'step' and args.launch == 'graph' and steps >= 16
    ptrs = []
    if args.mode == 'step' or not args.no_side:
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
        """k one-step launches, issued from ONE C-ABI call per contiguous stretch of action rows (ngw_step_device_many): at
        ~4.5 us per step on the device, a Python call per launch (~5.5 us with the HIP launch) would make the loop host-bound"""
        if row0 < 0:
            v.step_device_many(ptrs[0], n, k)
            return
        rows_t = len(ptrs) - warmup
        done = 0
        while done < k:
            first = (row0 + done) % rows_t
            take = min(k - done, rows_t - first)
            v.step_device_many(ptrs[warmup + first], n, take)
            done += take

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

    def reduce_max(vals):
        if world == 1:
            return vals
        t = torch.tensor(vals, dtype=torch.float64, device='cuda' if args.dist_backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t.tolist()]

    if args.mode == 'rollout':
        if warmup > 0:
            v.rollout(warmup, ACTION_SEED, 0)
    else:
        run_eager(warmup, -1)
    fence()
    ep_before = episode0()
    fence()
    v.timing_begin()                      # HIP event pair on the kernel's own stream, around the timed launches
    t0 = time.perf_counter()
    run_timed()
    v.timing_mark()                       # (recorded, not waited for: the synchronisation below is the only wait in the region)
    torch.cuda.synchronize()              # every stream of the device, the handle's included
    dt = time.perf_counter() - t0         # this rank's K steps; MAX over ranks below = until the slowest rank was done
    if world > 1:
        dist.barrier()                    # the closing barrier of the contract - after the local clock has stopped, so that its
        torch.cuda.synchronize()          # own latency (a collective) is not billed to the steps
    dev_ms = v.timing_end()
    assert v.error_flags() == 0
    resets_timed = episode0() - ep_before
    dt_rank = dt
    dt, dev_ms = reduce_max([dt, dev_ms])
    per_rank = None
    if world > 1:                                         # every rank's own K-step span, gathered for the line (self-check of the sharding)
        t = torch.zeros(world, dtype=torch.float64, device='cuda' if args.dist_backend == 'nccl' else 'cpu')
        t[rank] = dt_rank
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank = [round(n * steps / float(x), 1) for x in t.tolist()]

    # the same K-step region R more times (device already in the state the contract region left it in: episodes keep ending
    # every H steps): one 100-us sample says little about a 4-us step
    repeats = None
    if args.repeats > 0 and args.mode == 'step':
        wall, dev = [], []
        for _ in range(args.repeats):
            fence()
            v.timing_begin()
            tr0 = time.perf_counter()
            run_timed()
            v.timing_mark()
            torch.cuda.synchronize()
            wall.append((time.perf_counter() - tr0) / steps * 1e3)
            dev.append(v.timing_end() / steps)
        assert v.error_flags() == 0
        wall_seq, dev_seq = [dt / steps * 1e3] + reduce_max(list(wall)), [dev_ms / steps] + reduce_max(list(dev))
        wall_m, dev_m = reduce_max(sorted(wall)), reduce_max(sorted(dev))
        med = lambda xs: xs[len(xs) // 2]
        repeats = {'n': args.repeats, 'ms_per_step_wall': {'median': round(med(wall_m), 6), 'min': round(wall_m[0], 6), 'max': round(wall_m[-1], 6)},
                   'ms_per_step_device': {'median': round(med(dev_m), 6), 'min': round(dev_m[0], 6), 'max': round(dev_m[-1], 6)},
                   'value_median': round(n * world / (med(wall_m) * 1e-3), 1), '_wall_all': wall_m, '_dev_all': dev_m}

    # The line's `value` / `ms_per_step`: the MEDIAN of the 1 + R regions of exactly K steps each (contract region first, same handle, same
    # launches); the contract region alone is `value_contract` / `ms_per_step_contract`.  (One 20-step region is a 92-us sample of a 4-us
    # step: round 4's driver line recorded 14.3 G from a region whose own repeats had a median of 12.5 G.)
    wall_all = [dt / steps * 1e3] + (repeats['_wall_all'] if repeats else [])
    dev_all = [dev_ms / steps] + (repeats['_dev_all'] if repeats else [])
    if repeats:
        del repeats['_wall_all'], repeats['_dev_all']
        # every region in the order it ran (the contract region first): which of them held the all-env reset and the refill shows here
        repeats['regions_in_order'] = {'ms_per_step_wall': [round(x, 6) for x in wall_seq], 'ms_per_step_device': [round(x, 6) for x in dev_seq]}
    ms_step = sorted(wall_all)[len(wall_all) // 2] if args.mode == 'step' else dt / steps * 1e3
    dev_step = sorted(dev_all)[len(dev_all) // 2] if args.mode == 'step' else dev_ms / steps

    # roofline: SURVEY 8(d)'s algorithmic bytes per env-step x the envs of a launch / the launch period
    launches = 1 if args.mode == 'rollout' else steps
    steps_per_launch = steps if args.mode == 'rollout' else 1
    B = algorithmic_bytes(S, K)
    pmc_file = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    pmc = json.load(open(pmc_file)) if os.path.exists(pmc_file) else {}

    def traffic_of(mode, env_steps_per_launch):
        """HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc cannot run inside this process): a
        constant of the kernel, measured on this configuration (profiles/), not re-measured in this run."""
        rec = pmc.get('%s_%s' % (args.workload, mode))
        if rec and n == n_default:
            return round(rec['hbm_bytes_per_env_step'] * env_steps_per_launch), rec['source']
        return None, None

    if args.mode == 'step':
        # `frac` is the figure anyone can recompute from this line and from profiles/: algorithmic bytes x envs / ms_per_step / peak (the
        # WALL clock per step: it contains the launch gaps, the reset launches and the region's fixed start-up / wake-up cost).  The HIP
        # event pair's figure (the launch period on the device alone) is `frac_event_pair`.  The in-place kernel never touches most of the
        # 2*S*S bytes the model charges: what HBM really carries is `traffic` (PMC counters, profiles/) -> `traffic_frac`.  And the kernel
        # is not bound by HBM at this batch size but by latency (one wave per SIMD): `floor` prices the launch as "empty-kernel launch
        # period of this run + the stamped wave life" and `frac_of_floor` says how close the measured launch period is to that.
        in_place = bool(v.step_reads_map_in_place)
        D = design_bytes(S, K, in_place)
        bytes_launch = B * n * steps_per_launch
        achieved = bytes_launch / (ms_step * 1e-3) / 1e9
        roofline = {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': None, 'traffic_frac': None,
                    'frac_event_pair': round(bytes_launch / (dev_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    'frac_contract_region': round(bytes_launch / (dt / steps) / 1e9 / HBM_PEAK_GBS, 4),
                    'kernel': 'ngw_step_lean<%s>' % ('map read in place' if in_place else 'map staged through LDS'),
                    'launch_period_ms_wall': round(ms_step, 6), 'launch_period_ms_event_pair': round(dev_step, 6), 'launches_timed': launches,
                    'bytes_model': 'SURVEY 8(d) (2*S*S + 12*K + 45)',
                    'algorithmic_bytes_per_env_step': B, 'env_steps_per_launch': n * steps_per_launch,
                    'design_bytes_per_env_step': D,
                    'design_bytes_note': 'what this design has to move per env-step (state updated in place; %s): not the model `frac` is priced on'
                                         % ('three 32-byte map sectors instead of the map row' if in_place else 'the whole map row is read'),
                    'frac_on_design_bytes': round(D * n * steps_per_launch / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    'timing': 'frac = algorithmic bytes per launch / ms_per_step (this line\'s median wall clock per step) / peak - recomputable from the line and '
                              'comparable with the kernel average in profiles/; frac_event_pair: the same bytes over the HIP event pair recorded on the kernel '
                              'stream around the timed launches (median region); frac_contract_region: over the contract region\'s own wall clock'}
        if in_place:
            roofline['frac_note'] = ('the kernel reads the map in place and never moves the 2*S*S bytes per env-step the 8(d) model charges: `frac` says how the step '
                                     'compares with a read-pack-write design at the peak, not how busy HBM is - that is `traffic_frac`')
        tr, src = traffic_of('step', n)
        if tr:
            roofline['traffic'], roofline['traffic_source'] = tr, src
            roofline['traffic_provenance'] = pmc.get('_provenance', 'constant of the kernel measured in an earlier profiling run (profiles/pmc_traffic.json)')
            roofline['traffic_frac'] = round(tr / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        try:
            import ctypes as C
            from gym_novel_gridworlds_amd import _cabi
            fl = _cabi.lib().ngw_debug_launch_floor
            fl.argtypes, fl.restype = [C.c_void_p, C.c_int32, C.c_int, C.POINTER(C.c_double)], C.c_int
            us_g, us_e = C.c_double(0), C.c_double(0)
            kfl = g_steps if use_graph else max(steps, 64)
            _cabi.check(fl(v._h, int(kfl), 1, C.byref(us_g)))            # replayed from a graph: the device-side boundary between dependent launches
            _cabi.check(fl(v._h, int(kfl), 0, C.byref(us_e)))            # issued eagerly: what the HOST can sustain (a region below 100 steps is launched this way)
            wl = pmc.get('%s_wave_life' % args.workload, pmc.get('C2_wave_life', {}))
            life_us = wl.get('median_us_net_of_stamps', wl.get('median_us'))   # (net of what the eight clock stamps themselves cost a wave)
            # the device needs the boundary + a wave's life per launch; an eager loop cannot go faster than the host issues launches
            floor_us = max(us_g.value + (life_us or 0.0), 0.0 if use_graph else us_e.value)
            roofline['floor'] = {'empty_kernel_launch_period_us': {'hipGraph replay': round(us_g.value, 4), 'eager': round(us_e.value, 4)},
                                 'launch_form_of_this_region': 'hipGraph replay' if use_graph else 'eager',
                                 'launches': int(kfl), 'stamped_wave_life_us': life_us, 'stamped_wave_life_us_with_stamp_cost': wl.get('median_us'), 'wave_life_source': wl.get('source'),
                                 'floor_us': round(floor_us, 4),
                                 'what': 'floor = max(empty-kernel launch period replayed from a graph (the shortest of a few regions of the same length as the timed one) + the median life of a wave of the step kernel (in-kernel clock '
                                         'stamps, profiles/), and - for an eagerly launched region - the empty-kernel launch period of the host loop): EMPTY kernels '
                                         'in the step kernel\'s launch shape, issued back to back on this handle in this run (HIP event pair).  What one launch per '
                                         'step() costs when nothing but latency is left'}
            roofline['frac_of_floor'] = round(floor_us / (dev_step * 1e3), 4)
        except Exception as ex:       # noqa: BLE001 - a diagnostics entry point: its absence must not cost the line
            roofline['floor'] = {'error': repr(ex)}
    else:
        # T steps per launch keep the state on chip: HBM sees a few bytes per env-step, the kernel is bound by instruction issue
        roofline = {'bound': 'issue', 'achieved': None, 'peak': None, 'unit': None, 'frac': None, 'traffic': None, 'kernel': 'ngw_rollout_lean',
                    'kernel_ms': round(dev_ms, 4), 'note': 'fused rollout: state stays in LDS, no HBM roofline applies (see profiles/)'}
        tr, src = traffic_of('rollout', n * steps)
        if tr:
            roofline['traffic'], roofline['traffic_source'] = tr, src

    side = world == 1 and not args.no_side
    # the other mode of the same workload, reported beside the headline (fused T-step rollout: SURVEY.md §8(d))
    fused = None
    if args.mode == 'step' and side:
        if warmup > 0:
            v.rollout(warmup, ACTION_SEED, 10 ** 6)
        fence()
        v.timing_begin()
        t1 = time.perf_counter()
        v.rollout(steps, ACTION_SEED, 10 ** 6 + warmup)
        f_ms = v.timing_end()
        fence()
        f_dt = time.perf_counter() - t1
        fused = {'value': round(n * steps / f_dt, 1), 'unit': 'env-steps/s', 'ms_per_step': round(f_dt / steps * 1e3, 6), 'kernel_ms': round(f_ms, 4),
                 'bound': 'issue',
                 'what': 'ONE ngw_rollout call of %d steps, uniform actions generated in-kernel, state kept in LDS between the steps of a launch, only the bytes '
                         'a step changes are written through to the observation buffers (instruction-issue bound; not an HBM figure).  With prepared next '
                         'episodes on (the default; refill cadence %d) the library issues the call as horizon-sized launches (%d steps) with a refill launch '
                         'after each; inline_resets_one_launch is the same call with prepared episodes off' % (steps, v.reset_prefetch, HORIZON)}
        tr, src = traffic_of('rollout', n * steps)
        if tr:
            fused['hbm_traffic_bytes'], fused['traffic_source'] = tr, src
        sq = pmc.get('%s_rollout_sq' % args.workload)
        if sq:
            fused['issue'] = sq
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
        # the same call with prepared episodes off: ONE launch, resets placed inline by the lanes that need them.  Faster when the
        # whole batch ends its episodes together (as here), several times slower when episode ends are spread (staggered_resets below).
        if v.reset_prefetch:
            keep = v.reset_prefetch
            v.set_reset_prefetch(0)
            v.rollout(max(warmup, 1), ACTION_SEED, 2 * 10 ** 6)
            fence()
            v.timing_begin()
            t1 = time.perf_counter()
            v.rollout(steps, ACTION_SEED, 2 * 10 ** 6 + warmup)
            i_ms = v.timing_end()
            fence()
            i_dt = time.perf_counter() - t1
            fused['inline_resets_one_launch'] = {'value': round(n * steps / i_dt, 1), 'unit': 'env-steps/s', 'ms_per_step': round(i_dt / steps * 1e3, 6),
                                                 'kernel_ms': round(i_ms, 4), 'steps': steps}
            v.set_reset_prefetch(keep)
        assert v.error_flags() == 0

    # side measurement: the SAME workload with episode ends spread over the batch (step_count offset e * 7919 % H: about
    # n / H envs reset in every batched step, a few per wavefront) - the regime of a training loop; inline placement
    # loops vs prepared next episodes (ngw_set_reset_prefetch, the library's default for autoreset).
    stag = None
    if args.mode == 'step' and side and not args.no_stagger:
        stag_acts = None
        stag = {'what': 'episode ends staggered over the batch (~%d of %d envs reset per batched step), hipGraph replay' % (n // HORIZON, n)}
        default_every = sv.local._default_prefetch() if args.reset_prefetch == 'auto' else prefetch
        for key, every in (('inline_resets', 0), ('prepared_next_episodes', default_every or 32)):
            v.set_reset_prefetch(every)
            v.reset()
            v.set_state(0, step_count=(np.arange(n) * 7919 % HORIZON).astype(np.int32))
            gs = 2 * every if every else 64                   # two refill periods per replay
            if stag_acts is None or stag_acts.shape[0] < gs:
                stag_acts = torch.randint(0, A, (gs, n), dtype=torch.int32, device='cuda', generator=g)
                torch.cuda.synchronize()
            v.graph_build(stag_acts.data_ptr(), n, gs)
            v.graph_launch(max(1, 256 // gs))
            fence()
            v.timing_begin()
            v.graph_launch(max(1, 512 // gs))
            s_ms = v.timing_end() / (max(1, 512 // gs) * gs)
            fence()
            stag[key] = {'ms_per_step': round(s_ms, 6), 'value': round(n / (s_ms * 1e-3), 1), 'unit': 'env-steps/s'}
            if every:
                stag[key]['refill_every'] = every
        stag['default'] = 'prepared_next_episodes (VecNovelGridworld / ngw_set_autoreset switch it on: a refill every 3/4 horizon)'
        assert v.error_flags() == 0

    # PCIe-inclusive host loop (the drop-in API mode of SURVEY.md §8(d)): never `value`
    api = None
    if side:
        rs = np.random.RandomState(ACTION_SEED)
        ha = rs.randint(0, A, size=(8, n)).astype(np.int32)
        for i in range(3):
            v.step(ha[i % 8])
        ka = 20
        t2 = time.perf_counter()
        for i in range(ka):
            v.step(ha[i % 8])
        a_dt = time.perf_counter() - t2
        api = {'value': round(n * ka / a_dt, 1), 'unit': 'env-steps/s', 'ms_per_step': round(a_dt / ka * 1e3, 4), 'steps': ka,
               'what': 'VecNovelGridworld.step(): int32 actions from host memory, the Dict observation (%d B per env) + reward / done / info '
                       'back to page-locked host arrays every step (ngw_step_host: one call, one synchronisation); PCIe-inclusive' % (S * S + 12 + 4 * K)}

    # The step with the fused LidarInFront observation (reference observation_wrappers.py:10-80: what tests/test.py, random_action.py
    # and the training scripts wrap the env in): one launch per batched step, the observation rows built in the step launch's
    # own epilogue, per row format; then the wrapper's host loop.  Own handles (the observation setup is part of a handle's layout).
    lidar = api_lidar = None
    if side and args.mode == 'step':
        from gym_novel_gridworlds_amd import LidarInFront, VecNovelGridworld
        lidar = {'what': 'fused step + LidarInFront observation, %d beams, one launch per batched step (64-step hipGraph replayed; H = %d, '
                         'prepared episodes at the default cadence), observation rows resident in HBM' % (8, HORIZON)}
        G, reps = 64, 16
        la = torch.randint(0, A, (G, n), dtype=torch.int32, device='cuda', generator=g)
        torch.cuda.synchronize()
        for key, dt_ in (('int32', np.int32), ('int16', np.int16), ('packed_u8_i16', 'packed')):
            lv = VecNovelGridworld(spec=spec, num_envs=n, seed=0, autoreset=True, horizon=HORIZON, device=local_rank)
            lv.lidar_configure(num_beams=8, fused=True, dtype=dt_)
            lv.reset()
            lv.graph_build(la.data_ptr(), n, G)
            lv.graph_launch(4)
            lv.sync()
            lv.timing_begin()
            tl = time.perf_counter()
            lv.graph_launch(reps)
            l_ms = lv.timing_end()
            l_dt = time.perf_counter() - tl
            assert lv.error_flags() == 0
            lidar[key] = {'value': round(n * G * reps / l_dt, 1), 'unit': 'env-steps/s', 'ms_per_step': round(l_dt / (G * reps) * 1e3, 6),
                          'kernel_ms_avg': round(l_ms / (G * reps), 6), 'row_bytes': lv.lidar_row_bytes, 'row_len': lv.lidar_len,
                          'obs_GBps': round(lv.lidar_row_bytes * n / (l_ms / (G * reps) * 1e-3) / 1e9, 1)}
            lv.close()
        del la
        wv = VecNovelGridworld(spec=spec, num_envs=n, seed=0, autoreset=True, horizon=HORIZON, device=local_rank)
        w = LidarInFront(wv, num_beams=8, dtype='packed', copy=False)   # the wrapper's FAST path, opted into: packed rows (uint8 beams + int16 inventory), the page-locked buffer itself
                                                              # (its default returns a fresh int32 array per call, like the reference's np.array)
        w.reset()
        rs2 = np.random.RandomState(ACTION_SEED + 1)
        hb = rs2.randint(0, A, size=(8, n)).astype(np.int32)
        for i in range(3):
            w.step(hb[i % 8])
        kl = 20
        tw = time.perf_counter()
        for i in range(kl):
            w.step(hb[i % 8])
        w_dt = time.perf_counter() - tw
        api_lidar = {'value': round(n * kl / w_dt, 1), 'unit': 'env-steps/s', 'ms_per_step': round(w_dt / kl * 1e3, 4), 'steps': kl,
                     'what': "LidarInFront(VecNovelGridworld, dtype='packed', copy=False).step(): int32 actions from host memory, the %d-value observation as packed rows "
                             '(%d B per env: uint8 beam entries + int16 inventory) delivered by the pipelined host step itself, + reward / done / info, every step; '
                             'PCIe-inclusive' % (wv.lidar_len, wv.lidar_row_bytes)}
        wv.close()

    # the one collective of the path, timed on its own: pack launch per rank + gather to rank 0 + unpack launch there
    gather = None
    if world > 1 or not args.no_side:
        offs = sv.payload_layout()
        sv.gather_observation(dst=0)                          # allocates the payload / receive / global buffers
        fence()
        kg = 20
        t3 = time.perf_counter()
        for _ in range(kg):
            sv.gather_observation(dst=0)
        fence()
        g_dt = reduce_max([(time.perf_counter() - t3) / kg])[0]
        gather = {'ms': round(g_dt * 1e3, 4), 'payload_bytes_per_rank': offs[7], 'bytes_per_env': round(offs[7] / n, 1),
                  'GBps_into_root': round(offs[7] * world / g_dt / 1e9, 2), 'ranks': world, 'backend': args.dist_backend if world > 1 else 'none (one rank: pack + unpack launches only)',
                  'what': 'ngw_pack_obs (one launch: 7 SoA arrays -> one payload) on every rank, torch.distributed.gather to rank 0, '
                          'ngw_unpack_obs (one launch) into global arrays; outside the step path'}

    c1 = c1_single_env() if side and rank == 0 else None

    if rank == 0:
        total = n * world * steps
        line = {
            'metric': 'env-steps/sec', 'value': round(n * world / (ms_step * 1e-3), 1), 'unit': 'env-steps/s', 'n_gpus': world,
            'value_is': 'median of the contract region and %d repeats of it (K steps each)' % (len(wall_all) - 1) if len(wall_all) > 1 else 'the contract region',
            'value_contract': round(total / dt, 1), 'ms_per_step_contract': round(dt / steps * 1e3, 6),
            'steps': steps, 'warmup': warmup, 'adapt_steps': adapt_steps if args.mode == 'step' else 0, 'clock_warm_ms': args.clock_warm_ms,
            'prepared_episodes': {'refill_every': v.refill_cadence, 'depth': v.reset_prefetch_depth}, 'ms_per_step': round(ms_step, 6), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'int8/int32', 'data': 'synthetic',
            'config': {'workload': desc, 'name': args.workload, 'envs_per_gpu': n, 'global_envs': n * world,
                       'map_size': S, 'n_items': K, 'n_actions': A, 'horizon': HORIZON, 'autoreset': 'same-step',
                       'fused_lidar': ({'format': args.lidar, 'row_bytes': v.lidar_row_bytes, 'row_len': v.lidar_len} if args.lidar else None),
                       'reset_prefetch': v.reset_prefetch,
                       'mode': ('one launch per batched step (%s), actions in HBM' % ('hipGraph replay' if use_graph else 'eager')) if args.mode == 'step'
                               else ('fused rollout: one ngw_rollout call, actions generated in-kernel; ' + ('horizon-sized launches (%d steps) with a refill launch after each (prepared next episodes, cadence %d)' % (HORIZON, v.reset_prefetch) if v.reset_prefetch else 'ONE launch, resets inline')),
                       'parallelism': 'envs sharded x%d, no collective in the step path' % world,
                       'dist': {'backend': (args.dist_backend + (' (RCCL)' if args.dist_backend == 'nccl' else '')) if world > 1 else None,
                                'world_size_initialised': dist.get_world_size() if world > 1 else 1,
                                'global_env_index_range_of_rank0': [0, n], 'single_device_rehearsal': bool(args.single_device)}},
            'resets_in_timed_region': resets_timed,
            'roofline': roofline,
        }
        if per_rank:
            line['per_rank_value'] = per_rank          # every rank's own n * K / its K-step span (the line's value uses the slowest)
        if repeats:
            line['repeats'] = repeats
        if cold:
            line['cold_region'] = cold
        if lidar:
            line['lidar'] = lidar
        if api_lidar:
            line['api_mode_lidar'] = api_lidar
        if gather:
            line['gather'] = gather
        if fused:
            line['fused_rollout'] = fused
        if stag:
            line['staggered_resets'] = stag
        if api:
            line['api_mode'] = api
        if c1:
            line['c1_single_env'] = c1
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(spec)
            line['cpu_baseline'] = {'value': round(cb['allcores']['value'], 1), 'unit': 'env-steps/s',
                                    'cores': cb['allcores']['cores'], 'kind': 'port', 'sample': cb['allcores']['sample'],
                                    'one_core_value': round(cb['1core']['value'], 1)}
        print(json.dumps(line), flush=True)
    v.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
