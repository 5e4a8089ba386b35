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
    def packed_observation(self):
        """[n_local, S*S + 12 + 4K + 9] uint8: map | agent_location | agent_facing_id | inventory | reward | done | info."""
        import torch
        o, out = self.local.device_observation(), self.local.device_outputs()
        n = self.num_envs
        parts = [o['map'].reshape(n, -1).view(torch.uint8), o['agent_location'].reshape(n, 2).view(torch.uint8),
                 o['agent_facing_id'].reshape(n, 1).view(torch.uint8), o['inventory_items_quantity'].view(torch.uint8),
                 out['reward'].reshape(n, 1).view(torch.uint8), out['done'].reshape(n, 1).view(torch.uint8),
                 out['info'].reshape(n, 1).view(torch.uint8)]
        return torch.cat([p.reshape(n, -1) for p in parts], dim=1).contiguous()

    def unpack(self, packed):
        import torch
        S, K = self.spec.map_size, len(self.spec.items_id)
        n = packed.shape[0]
        ofs = [0]
        for w in (S * S, 8, 4, 4 * K, 4, 1, 4):
            ofs.append(ofs[-1] + w)
        cut = [packed[:, ofs[i]:ofs[i + 1]].contiguous() for i in range(7)]
        return {'map': cut[0].view(torch.int8).reshape(n, S, S), 'agent_location': cut[1].view(torch.int32).reshape(n, 2),
                'agent_facing_id': cut[2].view(torch.int32).reshape(n),
                'inventory_items_quantity': cut[3].view(torch.int32).reshape(n, K),
                'reward': cut[4].view(torch.int32).reshape(n), 'done': cut[5].reshape(n).bool(),
                'info': cut[6].view(torch.int32).reshape(n)}

    def gather_observation(self, dst=0):
        """Stack every rank's packed observation on rank `dst` (global env order).  Returns the unpacked dict of
        [global_num_envs, ...] tensors on `dst`, None elsewhere."""
        import torch
        self.local.sync()
        mine = self.packed_observation()
        if self.world == 1:
            return self.unpack(mine)
        bufs = [torch.empty_like(mine) for _ in range(self.world)] if self.rank == dst else None
        self.dist.gather(mine, bufs, dst=dst, group=self.group)
        if self.rank != dst:
            return None
        return self.unpack(torch.cat(bufs, dim=0))
