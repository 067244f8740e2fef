"""Multi-GPU layout: boards shard trivially.

One process per GPU (torchrun); rank r owns global envs [r*E, (r+1)*E) and passes
env_offset = r*E, so the counter RNG (keyed by the global env id) makes results
independent of the number of GPUs.  The step path has no collective; the only
exchange is one all_gather of 4 doubles per rank for logging (RCCL over xGMI on
GPUs, gloo in the CPU tests).
"""
import os

import torch
import torch.distributed as dist


def rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(total_envs, rank=None, world=None):
    """contiguous env range [lo, hi) of `rank`; sizes differ by at most one"""
    if rank is None or world is None:
        rank, world = rank_world()
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(int(total_envs), world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_episode_stats(local_stats):
    """local_stats: float64 [4] (sum_return, sum_len, n_episodes, n_agent_wins) on this rank's
    device -> (per-rank [world, 4], totals [4]).  One all_gather; a no-op without a process group."""
    local_stats = local_stats.to(torch.float64).reshape(4)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        per = local_stats.unsqueeze(0)
        return per, per.sum(0)
    bufs = [torch.empty_like(local_stats) for _ in range(dist.get_world_size())]
    dist.all_gather(bufs, local_stats)
    per = torch.stack(bufs)
    return per, per.sum(0)


def max_over_ranks(value, device=None):
    """max of a python float over all ranks (the bench's timing rule); identity without a group"""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_gradients(params, average=True):
    """Data-parallel PPO (optional): every rank collects rollouts from its own env shard
    and back-propagates its own minibatch; before optimizer.step() the gradients are summed over the
    ranks.  The SB3 default policy of the reference's scripts (obs -> 64 -> 64 -> N*N logits + value,
    ~41 k parameters = 164 KB in fp32 at 11x11) is far below one RCCL bucket, so ALL gradients travel as
    ONE flattened all-reduce (xGMI is point-to-point: one latency-bound ring pass instead of one per
    tensor), then are copied back in place.  A no-op without a process group.  Returns the bytes reduced."""
    params = list(params)
    if not params or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    # EVERY parameter takes part, in the caller's order, zeros where this rank has no gradient (a head that a
    # minibatch did not use): the flattened buffer has the same length and layout on every rank whatever each
    # rank's minibatch touched -- a per-rank "grad is not None" filter would reduce misaligned buffers (or hang).
    ref = next((p.grad for p in params if p.grad is not None), params[0])
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(ref.dtype)
                      for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    off = 0
    for p in params:
        n = p.numel()
        if p.grad is None:      # some other rank may have had one: adopt the reduced value
            p.grad = flat[off:off + n].view_as(p).clone()
        else:
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n
    return flat.numel() * flat.element_size()


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def summarize(totals):
    t = [float(x) for x in totals]
    n = max(t[2], 1.0)
    return {"episodes": t[2], "ep_rew_mean": t[0] / n, "ep_len_mean": t[1] / n, "agent_win_rate": t[3] / n}
