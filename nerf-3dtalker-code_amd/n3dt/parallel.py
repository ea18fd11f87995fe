"""Frame-level data parallelism: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The render path has no cross-frame term, so inference shards frames with no collective at all.
Training replicates the weights and averages gradients with ONE flat-buffer all-reduce per step:
HeadNeRFNet's ~3 M parameters (11-14 MB fp32) plus whatever extra modules the caller passes (the reference
trains an Audio2style LSTM next to the renderer, talker_trainer.py:665).  A single bucket keeps the ring
per-link bound at a few hundred microseconds on xGMI, well under one backward pass, so no overlap logic.
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous frame range [lo, hi) of `rank`; the first total % world ranks take one extra frame."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allreduce_gradients(params, world=None, group=None):
    """Average .grad over the ranks through one flat buffer (in place).  Parameters without a gradient
    contribute zeros so that every rank reduces the same layout."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world)
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank `src`'s weights (and buffers)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
