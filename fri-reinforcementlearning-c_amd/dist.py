"""Multi-GPU plumbing: one process per GPU, environments sharded by env id, statistics-only collective.

The path shards naturally: every environment owns its rule base, scratch and episode state, so
rank r simply runs environments [start, start+count) of the global batch and NO data-path collective exists.
What is exchanged is the reference's per-episode report (frirl_sequential_run.c:77-80: steps, reward, rules)
summed over the ranks: one all-reduce(SUM) of 5 doubles plus one all-reduce(MIN)/(MAX) of the reward --
latency-bound on xGMI (backend "nccl" is RCCL on ROCm; "gloo" for the CPU tests).
"""
import os
from dataclasses import dataclass


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def shard(total_envs, world, rank):
    """Contiguous balanced partition of env ids: (start, count) for `rank`; counts differ by at most one."""
    assert 0 <= rank < world and total_envs >= 0
    base, extra = divmod(total_envs, world)
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def init(device=None, backend=None):
    """Initialises torch.distributed from the torchrun environment (no-op for a single process)."""
    import torch.distributed as dist
    world, rank, _ = env_world()
    if world == 1 or dist.is_initialized():
        return world, rank
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if (device is not None and getattr(device, "type", "cpu") == "cuda") else "gloo"
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(backend)
    return world, rank


@dataclass
class EpisodeStats:
    reward_sum: float
    steps_sum: float
    success: float
    rules_sum: float
    envs: float
    reward_min: float
    reward_max: float

    @property
    def mean_reward(self):
        return self.reward_sum / max(self.envs, 1.0)

    @property
    def mean_rules(self):
        return self.rules_sum / max(self.envs, 1.0)


def local_stats(ep_reward, ep_steps, success, nrules):
    """[sum reward, sum steps, #success, sum rules, #envs], [min reward], [max reward] as float64 tensors
    on the inputs' device."""
    import torch
    s = torch.stack([ep_reward.sum(dtype=torch.float64), ep_steps.sum(dtype=torch.float64), success.sum(dtype=torch.float64),
                     nrules.sum(dtype=torch.float64), torch.tensor(float(ep_reward.numel()), dtype=torch.float64, device=ep_reward.device)])
    if ep_reward.numel():
        mn, mx = ep_reward.min().reshape(1).double(), ep_reward.max().reshape(1).double()
    else:
        mn = torch.full((1,), float("inf"), dtype=torch.float64, device=ep_reward.device)
        mx = torch.full((1,), float("-inf"), dtype=torch.float64, device=ep_reward.device)
    return s, mn, mx


def allreduce_stats(ep_reward, ep_steps, success, nrules):
    """Global EpisodeStats over all ranks (all-reduce of reward statistics only)."""
    import torch.distributed as dist
    s, mn, mx = local_stats(ep_reward, ep_steps, success, nrules)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    v = s.tolist()
    return EpisodeStats(v[0], v[1], v[2], v[3], v[4], float(mn.item()), float(mx.item()))


def max_over_ranks(value, device):
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
