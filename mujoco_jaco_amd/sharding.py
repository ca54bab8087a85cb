"""Multi-GPU layout: independent environment shards, one observation gather per rollout step.

Environments never interact (no cross-env term anywhere in env_script/env_mujoco*.py), so rank r of W owns the
contiguous env range [r*B/W, (r+1)*B/W) and steps it with no data-path collective.  The only exchange is the
concatenation of the per-rank observation rows ([B_local, 26] f32 = 6.8 MB at 65 536 envs) once per step:
torch.distributed.all_gather_into_tensor, i.e. RCCL over xGMI with backend "nccl" (gloo on CPU in the tests).
"""
import torch
import torch.distributed as dist


def shard_range(rank, world, total):
    """Contiguous env range [lo, hi) of `rank`; the remainder goes to the first ranks."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_seed(base_seed, rank):
    """Distinct counter-based RNG streams per rank (seeds rank * 2^32 + i)."""
    return (int(base_seed) + (int(rank) << 32)) & 0xFFFFFFFFFFFFFFFF


class ObsGather:
    """Pre-allocated all_gather of equally sized observation shards."""

    def __init__(self, local_rows, width, device, dtype=torch.float32, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.out = torch.empty(self.world * local_rows, width, device=device, dtype=dtype)

    def __call__(self, local_obs):
        if self.world == 1:
            self.out.copy_(local_obs)
        else:
            dist.all_gather_into_tensor(self.out, local_obs.contiguous(), group=self.group)
        return self.out
