"""Worker for tests/test_distributed_cpu.py::test_two_rank_slab_solve: two gloo ranks, ONE lattice cut into two y-slabs.
The host mirror of the slab path (quantum-mg_amd/sharding.py: slab_rows, slab_halo_exchange, dist_sum -- the message
pattern and row bookkeeping of qmg_halo_exchange / qmg_comm_set_distributed_reductions) drives a CG solve of the Wilson
normal equations in which every rank applies the operator to ITS rows only (numpy, rows -1 / Ly from the exchanged halos)
and every inner product is summed over the ranks; the assembled solution must solve the GLOBAL system of the CPU oracle."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sharding = importlib.import_module("quantum-mg_amd.sharding")
import oracle_lib as ol


def rows_of(a, Ly, per_row, y0, n):
    planes = a.size // (2 * Ly * per_row)
    return a.reshape(planes, 2, Ly, per_row)[:, :, y0:y0 + n].reshape(-1).copy()


def slab_apply(clover_s, hopping_s, x_s, lo, hi, Lx, Lyl, mass, dagger=False):
    """M x on a slab from its rows of the stored stencil (the matrix multiplying a neighbour lives at the OUTPUT site, so a slab
    of the stencil is row slices of its arrays) and the halo rows: embed the slab into a lattice of Lyl + 2 rows whose first and
    last row are the halos, apply the oracle there, keep the interior rows."""
    nc, row, hr = 2, (Lx // 2) * 2, Lx // 2
    Le = Lyl + 2                                      # even, since Lyl is: the extended lattice keeps the colouring if we shift by ONE row
    # a one-row shift flips the parity of every site: extended parity p' = 1 - p for the same (x, y)
    xe = np.zeros((2, Le, row), dtype=complex)
    xs = x_s.reshape(2, Lyl, row)
    xe[::-1, 1:Lyl + 1] = xs                          # p' = 1 - p
    xe[::-1, 0] = lo.reshape(2, row)
    xe[::-1, Le - 1] = hi.reshape(2, row)
    ce = np.zeros((1, 2, Le, hr * 4), dtype=complex)
    he = np.zeros((4, 2, Le, hr * 4), dtype=complex)
    ce[:, ::-1, 1:Lyl + 1] = clover_s.reshape(1, 2, Lyl, hr * 4)
    he[:, ::-1, 1:Lyl + 1] = hopping_s.reshape(4, 2, Lyl, hr * 4)
    # with the parities swapped, the x-neighbour bookkeeping s = (y' + p') & 1 = (y + 1 + 1 - p) & 1 = (y + p) & 1 is unchanged
    d = ol.make_desc(Lx, Le, nc, ce.reshape(-1), he.reshape(-1), mass)
    ye = ol.stencil_apply(d, xe.reshape(-1)).reshape(2, Le, row)
    return ye[::-1, 1:Lyl + 1].reshape(-1).copy()


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    Lx = Ly = 16
    mass = 0.2
    y0, Lyl = sharding.slab_rows(Ly, rank, world)
    try:
        sharding.slab_rows(18, 0, 4)
        raise SystemExit("an uneven split was accepted")
    except ValueError:
        pass
    gauge = ol.phases_to_gauge_u1(np.random.default_rng(11).uniform(-3, 3, 2 * Lx * Ly), Lx, Ly)
    clover, hopping = ol.wilson_fill(gauge, Lx, Ly)
    n, row = 2 * Lx * Ly, Lx
    rng = np.random.default_rng(5)
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    dglob = ol.make_desc(Lx, Ly, 2, clover, hopping, mass)
    cl_s, hp_s = rows_of(clover, Ly, (Lx // 2) * 4, y0, Lyl), rows_of(hopping, Ly, (Lx // 2) * 4, y0, Lyl)

    def apply_M(x_s):
        lo, hi = sharding.slab_halo_exchange(torch.from_numpy(x_s.copy()), Lx, Lyl, 2, rank, world, dist)
        return slab_apply(cl_s, hp_s, x_s, lo.numpy(), hi.numpy(), Lx, Lyl, mass)

    # 1. the slab apply with the REAL two-rank exchange reproduces this rank's rows of the global apply
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    want = rows_of(ol.stencil_apply(dglob, x), Ly, row, y0, Lyl)
    got = apply_M(rows_of(x, Ly, row, y0, Lyl))
    assert np.allclose(got, want, rtol=0, atol=1e-13), np.abs(got - want).max()

    # 2. a Krylov solve on slabs: BiCGStab with distributed inner products; every rank takes the same decisions
    def dot(u, v):
        return sharding.dist_sum(complex(np.vdot(u, v)), dist)
    bs = rows_of(b, Ly, row, y0, Lyl)
    xs = np.zeros_like(bs)
    r = bs.copy()
    rt = r.copy()
    p = np.zeros_like(r)
    v = np.zeros_like(r)
    rho = alpha = omega = 1.0 + 0j
    bnorm = np.sqrt(dot(bs, bs).real)
    its = 0
    while np.sqrt(dot(r, r).real) > 1e-10 * bnorm and its < 500:
        rho1 = dot(rt, r)
        beta = (rho1 / rho) * (alpha / omega)
        p = r + beta * (p - omega * v)
        v = apply_M(p)
        alpha = rho1 / dot(rt, v)
        s = r - alpha * v
        t = apply_M(s)
        omega = dot(t, s) / dot(t, t)
        xs += alpha * p + omega * s
        r = s - omega * t
        rho = rho1
        its += 1
    assert its < 500, its
    # the two slabs together solve the GLOBAL system: gather the solution and check it with the oracle
    full = torch.zeros(world, xs.size, dtype=torch.complex128)
    full[rank] = torch.from_numpy(xs)
    fr = torch.view_as_real(full).contiguous()
    dist.all_reduce(fr)
    slabs = torch.view_as_complex(fr).numpy()
    xg = np.zeros((2, Ly, row), dtype=complex)
    for rk in range(world):
        yy, nn = sharding.slab_rows(Ly, rk, world)
        xg[:, yy:yy + nn] = slabs[rk].reshape(2, nn, row)
    res = np.linalg.norm(b - ol.stencil_apply(dglob, xg.reshape(-1))) / np.linalg.norm(b)
    assert res < 1e-9, res
    its_all = torch.tensor([float(its)], dtype=torch.float64)
    dist.all_reduce(its_all)
    assert its_all.item() == its * world        # both ranks stopped at the same iteration
    if rank == 0:
        print("slab worker ok: %d BiCGStab iterations on %d slabs, global residual %.2e" % (its, world, res))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
