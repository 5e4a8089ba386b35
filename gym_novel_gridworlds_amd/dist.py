"""Multi-GPU sharding: one process per GPU, envs split in contiguous blocks, no data-path collective.

Environments never interact, so step()/reset() need no exchange: rank r owns global envs
[r*n_local, (r+1)*n_local) on its own GPU, keyed by GLOBAL env index so results do not depend on the GPU count.
The only collective is optional and sits outside the step path: `gather_observation()` stacks the packed
observation batch of all ranks on one rank, `all_gather_observation()` on every rank (RCCL over xGMI for device
tensors; gloo for the CPU tests and the one-GPU rehearsal).  On MI355X the 7 xGMI links of the root are all inbound
peers, so a direct gather moves each shard over its own link (shard bytes / ~153 GB/s) instead of a ring's per-link
serialisation.

Stream order instead of device-wide waits: the pack / unpack launches run on the handle's stream, the collective on
torch's current stream; `ngw_stream_order` (one event record + one stream wait, no host synchronisation) puts them
behind each other, and the tensors handed out are safe to use on torch's current stream.
"""

from .vec_env import VecNovelGridworld


def shard_range(global_num_envs, world, rank):
    """Contiguous block of rank `rank`; requires an even split so that gathers need no padding."""
    if global_num_envs % world:
        raise ValueError("global_num_envs (%d) must be divisible by the number of ranks (%d)" % (global_num_envs, world))
    n = global_num_envs // world
    return rank * n, n


def init_process_group(backend='nccl', local_rank=0):
    """torch.distributed.init_process_group for one process per GPU, failing LOUDLY: which rank, which device, which
    rendezvous - an RCCL start-up problem on a node must not look like a hang or a bare stack trace from c10d."""
    import os
    import torch
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    where = "rank %s/%s, local_rank %d, backend %s, MASTER_ADDR=%s MASTER_PORT=%s, HSA_ENABLE_IPC_MODE_LEGACY=%s" % (
        os.environ.get('RANK', '?'), os.environ.get('WORLD_SIZE', '?'), local_rank, backend, os.environ.get('MASTER_ADDR'),
        os.environ.get('MASTER_PORT'), os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'))
    try:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))   # RCCL on ROCm
            probe = torch.ones(1, device='cuda:%d' % local_rank)
            dist.all_reduce(probe)                          # the first collective builds the communicator: fail here, with context
            if int(probe.item()) != dist.get_world_size():
                raise RuntimeError("all_reduce of ones returned %r on a world of %d" % (probe.item(), dist.get_world_size()))
        else:
            dist.init_process_group(backend)
    except Exception as ex:                                 # noqa: BLE001 - re-raised with the rank / device map
        visible = os.environ.get('HIP_VISIBLE_DEVICES', os.environ.get('ROCR_VISIBLE_DEVICES', '(all)'))
        raise RuntimeError("torch.distributed start-up failed (%s; visible devices %s, %d seen by torch): %s: %s" % (
            where, visible, torch.cuda.device_count(), type(ex).__name__, ex)) from ex
    return dist


class ShardedVecNovelGridworld:
    """Rank-local view of `global_num_envs` environments sharded over the ranks of a torch.distributed group."""

    def __init__(self, env_id='NovelGridworld-Pogostick-v1', global_num_envs=65536, map_size=None, novelty=None, seed=0,
                 autoreset=False, horizon=0, spec=None, device=None, group=None, reset_prefetch='auto', reset_prefetch_depth=0,
                 exchange_always=False):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.global_num_envs = int(global_num_envs)
        self.first, self.num_envs = shard_range(self.global_num_envs, self.world, self.rank)
        self.local = self._make_local(env_id=env_id, map_size=map_size, novelty=novelty, spec=spec, device=device, num_envs=self.num_envs,
                                      seed=seed, autoreset=autoreset, horizon=horizon, env_index_base=self.first,
                                      reset_prefetch=reset_prefetch, reset_prefetch_depth=reset_prefetch_depth)
        self.spec = self.local.spec
        self._payload = self._recv = self._global = None        # gather buffers, allocated on first use
        # a one-rank group normally skips the collective (its stack is its own payload); True runs it anyway - a one-GPU box can
        # then exercise the RCCL calls themselves (tests/test_multi_gpu_rehearsal.py)
        self.exchange_always = bool(exchange_always)

    def _make_local(self, device=None, **kw):
        """This rank's envs on its GPU."""
        import torch
        return VecNovelGridworld(device=torch.cuda.current_device() if device is None else device, **kw)

    def rebuild(self, spec):
        """inject_novelty() on a shard: the local env is rebuilt in place on the edited spec and keeps its global env indices."""
        self.local.rebuild(spec)
        self.spec = self.local.spec
        self._payload = self._recv = self._global = None        # the payload layout follows the spec
        return self

    # step / reset are purely local
    def reset(self, mask=None):
        return self.local.reset(mask)

    def step(self, actions):
        return self.local.step(actions)

    def step_device(self, actions_ptr):
        return self.local.step_device(actions_ptr)

    def rollout(self, n_steps, action_seed=1234, t0=0):
        return self.local.rollout(n_steps, action_seed, t0)

    def sync(self):
        return self.local.sync()

    def close(self):
        return self.local.close()

    # ------------------------------------------------------------------ the one collective
    FIELDS = ('map', 'agent_location', 'agent_facing_id', 'inventory_items_quantity', 'reward', 'done', 'info')

    def payload_layout(self):
        """Byte offsets of the seven sections of one rank's payload + its size (ngw_pack_layout, include/ngw.h): the SoA
        arrays back to back, each padded to 16 bytes - every section is a straight coalesced copy."""
        return self.local.pack_layout()

    def _field_shapes(self, n):
        import torch
        S, K = self.spec.map_size, len(self.spec.items_id)
        return [((n, S, S), torch.int8), ((n, 2), torch.int32), ((n,), torch.int32), ((n, K), torch.int32),
                ((n,), torch.int32), ((n,), torch.uint8), ((n,), torch.int32)]

    def _torch_stream(self):
        import torch
        return torch.cuda.current_stream(self.local.device).cuda_stream

    def packed_observation(self):
        """This rank's payload: uint8 [payload_bytes] on the env's device, filled by ONE kernel launch (ngw_pack_obs) into a
        buffer allocated once; torch's current stream is ordered behind the launch (no host wait)."""
        import torch
        offs = self.payload_layout()
        if self._payload is None:
            self._payload = torch.empty(offs[7], dtype=torch.uint8, device='cuda:%d' % self.local.device)
        self.local.stream_order(self._torch_stream(), handle_waits=True)    # (the buffer's last reader ran on torch's stream)
        self.local.pack_obs(self._payload.data_ptr())
        self.local.stream_order(self._torch_stream(), handle_waits=False)   # the collective runs on torch's stream, the pack on the handle's
        return self._payload

    def unpack(self, payloads, world=None):
        """`world` payloads back to back (uint8 [world * payload_bytes]) -> dict of global arrays, rank r's envs at
        [r * n, (r + 1) * n).  Kernel launches (ngw_unpack_obs) into tensors allocated once, ordered behind torch's current
        stream (which produced `payloads`) and in front of it again (which will read the result); done comes back as bool."""
        import torch
        world = self.world if world is None else world
        shapes = self._field_shapes(self.num_envs * world)
        if self._global is None or self._global[0].shape[0] != self.num_envs * world:
            self._global = [torch.empty(sh, dtype=dt, device=payloads.device) for sh, dt in shapes]
        self.local.stream_order(self._torch_stream(), handle_waits=True)
        self.local.unpack_obs(payloads.data_ptr(), world, [t.data_ptr() for t in self._global])
        self.local.stream_order(self._torch_stream(), handle_waits=False)
        out = dict(zip(self.FIELDS, self._global))
        out['done'] = out['done'].bool()
        return out

    def _exchange(self, send, dst):
        """The collective itself: gather to `dst`, or all_gather when dst is None.  Returns [world, payload] on the receivers."""
        import torch
        receiver = dst is None or self.rank == dst
        if receiver and (self._recv is None or self._recv.device != send.device or self._recv.shape[1] != send.numel()):
            self._recv = torch.empty((self.world, send.numel()), dtype=torch.uint8, device=send.device)
        try:
            if dst is None:
                self.dist.all_gather_into_tensor(self._recv.reshape(-1), send, group=self.group)
            else:
                self.dist.gather(send, [self._recv[r] for r in range(self.world)] if receiver else None, dst=dst, group=self.group)
        except Exception as ex:                             # noqa: BLE001 - re-raised with the rank / device map
            raise RuntimeError("observation %s failed on rank %d of %d (device %s, backend %s, %d payload bytes): %s: %s" % (
                'all_gather' if dst is None else 'gather to rank %d' % dst, self.rank, self.world, send.device,
                self.dist.get_backend(self.group), send.numel(), type(ex).__name__, ex)) from ex
        return self._recv if receiver else None

    def _stack(self, dst):
        mine = self.packed_observation()
        if self.world == 1 and not (self.exchange_always and self.dist.is_initialized()):
            return self.unpack(mine, 1)
        host_side = self.dist.get_backend(self.group) == 'gloo' and mine.is_cuda          # gloo moves host tensors
        send = mine.cpu() if host_side else mine            # (.cpu() waits for torch's stream, which waits for the pack)
        recv = self._exchange(send, dst)
        if recv is None:
            return None
        stacked = recv.to(mine.device) if host_side else recv
        return self.unpack(stacked.reshape(-1))

    def gather_observation(self, dst=0):
        """Stack every rank's payload on rank `dst` (global env order): one pack launch per rank, ONE collective
        (torch.distributed.gather: RCCL over xGMI for device tensors, gloo on host copies), the unpack launches on `dst`.
        Returns the dict of [global_num_envs, ...] tensors on `dst`, None elsewhere."""
        return self._stack(dst)

    def all_gather_observation(self):
        """The same stack on EVERY rank (torch.distributed.all_gather_into_tensor): what a learner replicated on all ranks reads."""
        return self._stack(None)
