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


# ---------------- y-slab decomposition of ONE lattice (mirror of csrc/qmg_comm.hip: qmg_halo_exchange) ----------------
def slab_rows(Ly, rank, world):
    """Rows [y0, y0 + n) of rank `rank`: equal slabs of an EVEN number of rows (the slab's colouring is the global one)."""
    if world < 1 or not (0 <= rank < world) or Ly % world or (Ly // world) % 2 or Ly // world < 2:
        raise ValueError("%d rows do not split into %d slabs of an even number of rows" % (Ly, world))
    n = Ly // world
    return rank * n, n


def slab_halo_exchange(vec, Lx, Ly_local, nc, rank, world, dist):
    """The message pattern of qmg_halo_exchange on a host tensor: `vec` is a slab in the even-odd layout
    [parity][Ly_local][Lx/2][nc] (complex128, flat).  Returns (halo_lo, halo_hi), each [parity][Lx/2][nc]: the LAST row of
    rank-1 and the FIRST row of rank+1 (periodic over the ranks).  Order of the point-to-point calls as in the C code:
    per parity, send last -> up, send first -> down, recv lo <- down, recv hi <- up; with two ranks up == down and the
    messages to one peer are matched in issue order (hence the tags here)."""
    import torch
    row = (Lx // 2) * nc
    v = vec.view(2, Ly_local, row)
    lo, hi = torch.empty(2, row, dtype=vec.dtype), torch.empty(2, row, dtype=vec.dtype)
    if dist is None or world == 1:
        lo.copy_(v[:, Ly_local - 1])
        hi.copy_(v[:, 0])
        return lo.reshape(-1), hi.reshape(-1)
    up, down = (rank + 1) % world, (rank + world - 1) % world
    reqs = []
    for q in range(2):
        last, first = v[q, Ly_local - 1].contiguous(), v[q, 0].contiguous()
        reqs.append(dist.isend(last, up, tag=2 * q))         # my last row is up's row "-1"
        reqs.append(dist.isend(first, down, tag=2 * q + 1))  # my first row is down's row "Ly"
        reqs.append(dist.irecv(lo[q], down, tag=2 * q))
        reqs.append(dist.irecv(hi[q], up, tag=2 * q + 1))
    for r in reqs:
        r.wait()
    return lo.reshape(-1), hi.reshape(-1)


def dist_sum(value, dist):
    """A reduction result summed over the slabs (qmg_comm_set_distributed_reductions)."""
    if dist is None:
        return value
    import torch
    t = torch.tensor([value.real, value.imag] if isinstance(value, complex) else [float(value), 0.0], dtype=torch.float64)
    dist.all_reduce(t)
    return complex(t[0].item(), t[1].item()) if isinstance(value, complex) else t[0].item()
