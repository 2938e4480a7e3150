"""Worker for tests/test_distributed_cpu.py: run under torch.distributed.run with backend gloo (world_size 2).
Exercises the N>1 host path of bench.py on CPU tensors: RHS sharding, the single all-reduce that shows every rank
every per-RHS norm, the lock-step convergence decision, and the max-over-ranks timing reduction."""
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


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    total_rhs, n = 5, 8 * 8 * 1                     # 5 right-hand sides over 2 ranks: 3 + 2 (ragged)
    mine = sharding.shard_rhs(total_rhs, rank, world)
    # every RHS owned exactly once
    owners = torch.zeros(total_rhs, dtype=torch.float64)
    owners[mine] = 1.0
    dist.all_reduce(owners)
    assert torch.equal(owners, torch.ones(total_rhs, dtype=torch.float64)), owners

    # each rank applies the (oracle) staggered operator to ITS right-hand sides and reduces their norms locally ...
    Lx = Ly = 8
    gauge = ol.phases_to_gauge_u1(np.random.default_rng(3).uniform(-3, 3, 2 * Lx * Ly), Lx, Ly)
    hop = ol.staggered_fill(gauge, Lx, Ly)
    d = ol.make_desc(Lx, Ly, 1, None, hop, 0.04)
    def rhs_vec(k):
        rng = np.random.default_rng(100 + k)          # RHS k is the same whoever owns it
        return rng.standard_normal(n) + 1j * rng.standard_normal(n)
    buf = torch.zeros(total_rhs, dtype=torch.float64)
    for k in mine:
        buf[k] = ol.norm2sq(ol.stencil_apply(d, rhs_vec(k)))
    # ... and ONE all-reduce shows every rank every norm
    sharding.allgather_by_allreduce(buf, total_rhs, rank, world, dist)
    want = torch.tensor([ol.norm2sq(ol.stencil_apply(d, rhs_vec(k))) for k in range(total_rhs)], dtype=torch.float64)
    assert torch.allclose(buf, want, rtol=1e-14), (buf, want)

    # a REUSED buffer (bench.py StaggeredMultiRHS.step): after the first step the foreign slots hold last step's values;
    # with `own` they are cleared before the sum, so every step returns exactly the fresh values (three steps here)
    slots = torch.zeros(world * 2, dtype=torch.float64)
    for step in range(3):
        slots[2 * rank] = 10.0 * step + rank
        slots[2 * rank + 1] = 100.0 * step + rank
        sharding.allgather_by_allreduce(slots, world * 2, rank, world, dist, own=(2 * rank, 2 * rank + 2))
        fresh = torch.tensor([v for r in range(world) for v in (10.0 * step + r, 100.0 * step + r)], dtype=torch.float64)
        assert torch.equal(slots, fresh), (step, slots, fresh)

    # the C-ABI's own rendezvous (qmg_comm_rendezvous: the TCP leg of qmg_comm_init_env, no GPU involved): rank 0's
    # 128 bytes reach every rank
    import ctypes as C
    qmg = importlib.import_module("quantum-mg_amd")
    os.environ["QMG_COMM_PORT"] = str(int(os.environ["MASTER_PORT"]) + 7)
    blob = (C.c_ubyte * 128)(*([(7 * i + 3) % 251 for i in range(128)] if rank == 0 else [0] * 128))
    assert qmg.lib().qmg_comm_rendezvous(blob, world, rank) == 0
    assert list(blob) == [(7 * i + 3) % 251 for i in range(128)]

    # lock-step decision: identical on all ranks, true only when ALL rhs are below tolerance
    bn = torch.tensor([ol.norm2sq(rhs_vec(k)) for k in range(total_rhs)], dtype=torch.float64)
    res = bn * 1e-22
    assert sharding.all_converged(res, bn, 1e-10)
    res[total_rhs - 1] = bn[total_rhs - 1] * 1e-18       # one straggler owned by the LAST rank keeps everyone iterating
    flag = torch.tensor([0.0 if sharding.all_converged(res, bn, 1e-10) else 1.0], dtype=torch.float64)
    dist.all_reduce(flag)
    assert flag.item() == world, flag

    # timing contract: max over ranks
    assert sharding.max_over_ranks(1.0 + rank, dist, "cpu") == float(world)
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
