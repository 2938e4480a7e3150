"""Host-side sharding helpers for the multi-GPU path (one process per GPU, independent right-hand sides).

Plumbing only: which RHS a rank owns, and the single small all-reduce that lets every rank see every
per-RHS reduction result.  `dist` is torch.distributed (backend "nccl" == RCCL on the GPU box, "gloo" in the
CPU tests) or None for a single process.
"""


def shard_rhs(total_rhs, rank, world):
    """Contiguous block of right-hand-side indices owned by `rank`; sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world) or total_rhs < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(total_rhs, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def slot_range(total_rhs, rank, world):
    idx = shard_rhs(total_rhs, rank, world)
    return (idx[0], idx[-1] + 1) if idx else (0, 0)


def allgather_by_allreduce(buf, total_rhs, rank, world, dist, own=None):
    """ONE sum all-reduce per global reduction step.  `buf` is a float64 tensor of total_rhs (x width) entries in
    which this rank has filled its own slots; afterwards every rank holds every entry, so all ranks take the same
    convergence / restart decision in lock-step.  The slots a rank does not own must be ZERO when the sum is taken:
    pass `own` = (first, one-past-last) index of this rank's slots to have every other slot cleared first -- a buffer
    that is reused from step to step still holds the other ranks' values of the previous step, which would otherwise
    be summed in again (world - 1 stale copies per step)."""
    if dist is not None and world > 1:
        if own is not None:
            buf[:own[0]].zero_()
            buf[own[1]:].zero_()
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf


def all_converged(norms_sq, bnorms_sq, tol):
    """Lock-step stopping rule over ALL right-hand sides (same on every rank after the all-reduce)."""
    return bool(((norms_sq <= (tol * tol) * bnorms_sq)).all())


def max_over_ranks(value, dist, device):
    """Max of a python float over ranks (bench timing contract)."""
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
