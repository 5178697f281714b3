"""spp-split data parallelism: every rank renders its own slice of the global sample indices of
every pixel with a replicated scene; the f64 accumulation framebuffers are summed onto rank 0 with one
``torch.distributed.reduce`` (RCCL over xGMI on the GPU box, gloo in the CPU tests)."""


def spp_range(rank, world_size, spp_total):
    """Global sample indices [begin, begin+count) of `rank`; remainders go to the lowest ranks."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, rem = divmod(int(spp_total), int(world_size))
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)


def reduce_framebuffer(fb, dst=0, force=False):
    """Sum the per-rank framebuffers (a torch tensor, any device) onto rank `dst`; no-op for a single process unless `force`
    (a one-rank rehearsal of the collective: the sum over one rank is the identity)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force):
        dist.reduce(fb, dst=dst, op=dist.ReduceOp.SUM)
    return fb
