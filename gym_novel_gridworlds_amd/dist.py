"""Multi-GPU sharding: one process per GPU, envs split in contiguous blocks, no data-path collective.

Environments never interact, so step()/reset() need no exchange: rank r owns global envs
[r*n_local, (r+1)*n_local) on its own GPU, keyed by GLOBAL env index so results do not depend on the GPU count.
The only collective is optional and sits outside the step path: `gather_observation()` stacks the packed
observation batch of all ranks on one rank (RCCL gather over xGMI for device tensors; gloo for the CPU tests).
On MI355X the 7 xGMI links of the root are all inbound peers, so a direct gather moves each shard over its own
link (shard bytes / ~153 GB/s) instead of a ring's per-link serialisation.
"""

from .vec_env import VecNovelGridworld


def shard_range(global_num_envs, world, rank):
    """Contiguous block of rank `rank`; requires an even split so that gathers need no padding."""
    if global_num_envs % world:
        raise ValueError("global_num_envs (%d) must be divisible by the number of ranks (%d)" % (global_num_envs, world))
    n = global_num_envs // world
    return rank * n, n


class ShardedVecNovelGridworld:
    """Rank-local view of `global_num_envs` environments sharded over the ranks of a torch.distributed group."""

    def __init__(self, env_id='NovelGridworld-Pogostick-v1', global_num_envs=65536, map_size=None, novelty=None, seed=0,
                 autoreset=False, horizon=0, spec=None, device=None, group=None, local_factory=None, reset_prefetch='auto'):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.global_num_envs = int(global_num_envs)
        self.first, self.num_envs = shard_range(self.global_num_envs, self.world, self.rank)
        kw = dict(num_envs=self.num_envs, seed=seed, autoreset=autoreset, horizon=horizon, env_index_base=self.first)
        if local_factory is not None:                      # CPU tests: oracle-backed stand-in
            self.local = local_factory(spec=spec, env_id=env_id, map_size=map_size, novelty=novelty, **kw)
        else:
            import torch
            dev = torch.cuda.current_device() if device is None else device
            self.local = VecNovelGridworld(env_id=env_id, map_size=map_size, novelty=novelty, spec=spec, device=dev,
                                           reset_prefetch=reset_prefetch, **kw)
        self.spec = self.local.spec
        self._payload = self._recv = self._global = None        # gather buffers, allocated on first use

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
        """Byte offsets of the seven sections of one rank's payload + its size (mirror of ngw_pack_layout, include/ngw.h):
        the SoA arrays back to back, each padded to 16 bytes - every section is a straight coalesced copy."""
        if hasattr(self.local, 'pack_layout'):
            return self.local.pack_layout()
        n, S, K = self.num_envs, self.spec.map_size, len(self.spec.items_id)
        offs = [0]
        for w in (S * S, 8, 4, 4 * K, 4, 1, 4):
            offs.append(offs[-1] + ((n * w + 15) & ~15))
        return offs

    def _field_shapes(self, n):
        import torch
        S, K = self.spec.map_size, len(self.spec.items_id)
        return [((n, S, S), torch.int8), ((n, 2), torch.int32), ((n,), torch.int32), ((n, K), torch.int32),
                ((n,), torch.int32), ((n,), torch.uint8), ((n,), torch.int32)]

    def packed_observation(self):
        """This rank's payload: uint8 [payload_bytes] on the env's device, filled by ONE kernel launch (ngw_pack_obs) into a
        buffer allocated once.  (The CPU stand-in of the tests builds the same bytes on the host.)"""
        import torch
        offs = self.payload_layout()
        if hasattr(self.local, 'pack_obs'):
            if self._payload is None:
                self._payload = torch.empty(offs[7], dtype=torch.uint8, device='cuda:%d' % self.local.device)
            self.local.pack_obs(self._payload.data_ptr())
            self.local.sync()                               # the collective runs on torch's stream, the pack on the handle's
            return self._payload
        o, out = self.local.device_observation(), self.local.device_outputs()
        buf = torch.zeros(offs[7], dtype=torch.uint8)
        parts = [o['map'], o['agent_location'], o['agent_facing_id'], o['inventory_items_quantity'], out['reward'], out['done'], out['info']]
        for off, t in zip(offs, parts):
            b = t.contiguous().reshape(-1).view(torch.uint8)
            buf[off:off + b.numel()] = b
        return buf

    def unpack(self, payloads, world=None):
        """`world` payloads back to back (uint8 [world * payload_bytes]) -> dict of global arrays, rank r's envs at
        [r * n, (r + 1) * n).  One kernel launch (ngw_unpack_obs) into tensors allocated once; the done flags come back as bool."""
        import torch
        world = self.world if world is None else world
        n, offs = self.num_envs, self.payload_layout()
        shapes = self._field_shapes(n * world)
        if hasattr(self.local, 'unpack_obs'):
            if self._global is None:
                self._global = [torch.empty(sh, dtype=dt, device=payloads.device) for sh, dt in shapes]
            self.local.unpack_obs(payloads.data_ptr(), world, [t.data_ptr() for t in self._global])
            self.local.sync()
            out = dict(zip(self.FIELDS, self._global))
        else:
            pl = payloads.reshape(world, offs[7])
            out = {}
            for name, off, (sh, dt), (sh1, _) in zip(self.FIELDS, offs, shapes, self._field_shapes(n)):
                nbytes = int(torch.tensor([], dtype=dt).element_size())
                for d in sh1:
                    nbytes *= d
                out[name] = pl[:, off:off + nbytes].contiguous().view(dt).reshape(sh)
        out = dict(out)
        out['done'] = out['done'].bool()
        return out

    def gather_observation(self, dst=0):
        """Stack every rank's payload on rank `dst` (global env order): one pack launch per rank, ONE collective
        (torch.distributed.gather: RCCL over xGMI for device tensors, gloo on host copies), one unpack launch on `dst`.
        Returns the dict of [global_num_envs, ...] tensors on `dst`, None elsewhere."""
        import torch
        mine = self.packed_observation()
        if self.world == 1:
            return self.unpack(mine, 1)
        host_side = self.dist.get_backend(self.group) == 'gloo' and mine.is_cuda          # gloo gathers host tensors
        send = mine.cpu() if host_side else mine
        recv = None
        if self.rank == dst:
            if self._recv is None or self._recv.device != send.device:
                self._recv = torch.empty((self.world, send.numel()), dtype=torch.uint8, device=send.device)
            recv = [self._recv[r] for r in range(self.world)]
        self.dist.gather(send, recv, dst=dst, group=self.group)
        if send.is_cuda:
            # RCCL runs the collective on torch's stream and only orders THAT stream behind it; the unpack launch goes to the
            # handle's own stream, so wait for the device here (host tensors / gloo: the call above is already synchronous)
            torch.cuda.synchronize(send.device)
        if self.rank != dst:
            return None
        stacked = self._recv.to(mine.device) if host_side else self._recv
        return self.unpack(stacked.reshape(-1))
