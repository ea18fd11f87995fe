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
    """Make every rank start from rank `src`'s weights (and buffers): ONE broadcast of a flat buffer, copied back
    into the tensors under no_grad.  The copy moves the version counters (a collective writing a tensor in place does
    not), so packed copies of the weights are rebuilt; modules that keep such copies (HeadNeRFNet.invalidate_packed)
    are told explicitly as well."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    tensors = list(module.parameters()) + list(module.buffers())
    by_kind = {}
    for t in tensors:
        by_kind.setdefault((t.dtype, t.device), []).append(t)
    with torch.no_grad():
        for ts in by_kind.values():
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
    for m in module.modules():
        if hasattr(m, "invalidate_packed"):
            m.invalidate_packed()


class FlatBucket(torch.nn.Module):
    """A flat fp32 parameter standing in for a co-trained module's gradients in the step's all-reduce.  The reference
    trains an Audio2style LSTM next to the renderer (talker_trainer.py:428-473, second Adam at :665): 21.5 M parameters,
    86 MB of fp32 gradients riding in the same bucket as HeadNeRFNet's 11-14 MB."""

    AUDIO2STYLE_PARAMS = 21_546_624  # nn.LSTM(1280, 640, 2 layers, bidirectional) + RNNModel.fc1 (1280->640, unused in forward but a parameter) + Linear 1280-640-320-64 (talker_trainer.py:408-461)

    def __init__(self, numel=AUDIO2STYLE_PARAMS):
        super().__init__()
        self.flat = torch.nn.Parameter(torch.zeros(numel))

    def fill_grad(self, value=0.0):
        if self.flat.grad is None:
            self.flat.grad = torch.full_like(self.flat, value)
        else:
            self.flat.grad.fill_(value)
