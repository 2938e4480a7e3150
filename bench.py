#!/usr/bin/env python3
"""bench.py -- fine Wilson stencil apply on MI355X (BASELINE.json metric), one process per GPU.

A "step" is one pass of the hot path over one synthetic right-hand side:
    apply_stencil_2D_M  (stencil/stencil_2d.h:2571-2576)  ==  lhs = (clover + hopping + shift) rhs
on the L x L even-odd lattice, Wilson nc = 2 (2 spin components over a U(1) gauge field), fp64.
Default workload: L = 4096 (the north-star target); the 2048^2 configuration is timed as well and
reported under "also".  Inputs are resident in HBM before the timed region.

Multi-GPU: independent right-hand sides, one per rank, every rank holding a replica of the stencil
(SURVEY 8e).  The apply has no exchange step, so there is no data-path collective: weak scaling,
`value` = (sites processed by all ranks) * 176 flop / max-over-ranks time.

The JSON line also carries
  roofline     -- ALGORITHMIC bytes per launch (384 B/site: five 2x2 matrices + rhs + lhs)
                  / average launch duration measured live with HIP events on the launch stream,
                  against the 8 TB/s HBM3E peak;
  cpu_baseline -- the reference-structured CPU oracle (kind "port", 1 thread) timed on this box's host
                  cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FLOP_PER_SITE = 176          # 8 nc^2 * 5 + 8 nc, nc = 2
BYTES_PER_SITE = 384         # 5 nc^2 c + 2 nc c, c = 16 B
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MASS = -0.07                 # beta = 6.0, m_crit ~ -0.0706


def site_index_grid(Lx, Ly):
    x = np.arange(Lx, dtype=np.int64)[:, None]
    y = np.arange(Ly, dtype=np.int64)[None, :]
    p = (x + y) & 1
    return (y + p * Ly) * (Lx // 2) + x // 2


def tiled_gauge(L, fixture_path):
    """Periodic tiling of the committed 64^2 U(1) fixture to L x L, in the reference's (mu,eo,y,x) layout."""
    ph = np.loadtxt(fixture_path).reshape(64, 64, 2)
    reps = L // 64
    idx = site_index_grid(L, L)
    g = np.empty(2 * L * L, dtype=np.complex128)
    for mu in range(2):
        u = np.exp(1j * np.tile(ph[:, :, mu], (reps, reps)))
        g[mu * L * L + idx.reshape(-1)] = u.reshape(-1)
    return g


def tile_vector(v64, L, nc):
    """Periodic tiling of a 64^2 (eo,y,x,c) vector to L^2 (for the size-independent parity gate)."""
    grid = v64.reshape(64 * 64, nc)[site_index_grid(64, 64)]            # [x,y,c]
    reps = L // 64
    big = np.tile(grid, (reps, reps, 1))
    out = np.empty((L * L, nc), dtype=np.complex128)
    out[site_index_grid(L, L)] = big
    return out.reshape(-1)


class Workload:
    def __init__(self, qmg, L, fixture, seed):
        self.qmg, self.L = qmg, L
        vol = L * L
        g = qmg.DeviceArray.from_host(tiled_gauge(L, fixture))
        self.clover = qmg.DeviceArray(4 * vol)
        self.hopping = qmg.DeviceArray(16 * vol)
        qmg.wilson_fill(self.clover, self.hopping, g, L, L, 1.0)
        qmg.sync()
        g.free()
        self.desc = qmg.make_desc(L, L, 2, self.clover, self.hopping, MASS)
        self.rhs = qmg.DeviceArray(2 * vol)
        self.lhs = qmg.DeviceArray(2 * vol)
        qmg.gaussian(self.rhs, 2 * vol, seed)
        qmg.sync()

    def step(self):
        self.qmg.stencil_apply(self.desc, self.lhs, self.rhs, self.qmg.P_ALL | self.qmg.P_ZERO)

    def parity_gate(self, fixture):
        """GPU at full size vs the CPU oracle through periodicity: a 64-periodic rhs on the 64-periodic gauge
        field gives the 64-periodic image of the oracle's 64^2 result."""
        import oracle_lib as ol
        qmg, L = self.qmg, self.L
        ph = np.loadtxt(fixture)
        clover, hopping = ol.wilson_fill(ol.phases_to_gauge_u1(ph, 64, 64), 64, 64)
        rng = np.random.default_rng(1337)
        v = rng.standard_normal(64 * 64 * 2) + 1j * rng.standard_normal(64 * 64 * 2)
        want = tile_vector(ol.stencil_apply(ol.make_desc(64, 64, 2, clover, hopping, MASS), v), L, 2)
        keep = self.rhs.to_host()
        self.rhs.upload(tile_vector(v, L, 2))
        self.step()
        got = self.lhs.to_host()
        self.rhs.upload(keep)
        err = float(np.linalg.norm(got - want) / np.linalg.norm(want))
        if not err < 1e-13:
            raise SystemExit("parity gate failed at L=%d: rel L2 error %.3e" % (L, err))
        return err

    def free(self):
        for a in (self.clover, self.hopping, self.rhs, self.lhs):
            a.free()


class StaggeredMultiRHS:
    """BASELINE configs[3]: staggered Dslash, L x L, nrhs independent right-hand sides per GPU sharing ONE read of the
    hopping matrices per site, followed by the per-RHS residual-norm reduction of a Krylov step and the single small
    all-reduce that shows every rank every norm (64 RHS over 8 GPUs = 8 per rank)."""
    MASS = 0.04          # tests/n20...:43
    FLOP_PER_SITE_RHS = 40
    def __init__(self, qmg, L, fixture, seed, nrhs, rank, world, dist, torch):
        self.qmg, self.L, self.nrhs, self.rank, self.world, self.dist = qmg, L, nrhs, rank, world, dist
        vol = L * L
        self.vol = vol
        g = qmg.DeviceArray.from_host(tiled_gauge(L, fixture))
        self.hopping = qmg.DeviceArray(4 * vol)
        qmg.staggered_fill(self.hopping, g, L, L)
        qmg.sync()
        g.free()
        self.desc = qmg.make_desc(L, L, 1, None, self.hopping, self.MASS)
        self.rhs = qmg.DeviceArray(nrhs * vol)
        self.lhs = qmg.DeviceArray(nrhs * vol)
        qmg.gaussian(self.rhs, nrhs * vol, seed)
        # all per-RHS norms of the whole job in one small buffer; this rank fills slots [rank*nrhs, (rank+1)*nrhs)
        self.norms = torch.zeros(world * nrhs, dtype=torch.float64, device="cuda")
        qmg.sync()
        self.bytes_per_site_rhs = 4 * 16.0 / nrhs + 32.0

    def step(self):
        import ctypes as C
        from importlib import import_module
        sharding = import_module("quantum-mg_amd.sharding")
        q = self.qmg
        # the apply leaves |lhs_k|^2 from the same pass (kernel A2 with NORM): the vectors are not read again for the norms
        base = self.norms.data_ptr() + 8 * self.rank * self.nrhs
        q.stencil_apply_norm2(self.desc, self.lhs, self.rhs, q.P_ALL | q.P_ZERO, nrhs=self.nrhs, vec_stride=self.vol, norms_dev=base)
        # the buffer is reused from step to step: the slots of the other ranks are cleared before the sum (sharding.py)
        sharding.allgather_by_allreduce(self.norms, self.world * self.nrhs, self.rank, self.world, self.dist, own=(self.rank * self.nrhs, (self.rank + 1) * self.nrhs))

    def parity_gate(self, fixture):
        import oracle_lib as ol
        qmg, L, vol = self.qmg, self.L, self.vol
        ph = np.loadtxt(fixture)
        hop = ol.staggered_fill(ol.phases_to_gauge_u1(ph, 64, 64), 64, 64)
        rng = np.random.default_rng(1337)
        v = rng.standard_normal(64 * 64) + 1j * rng.standard_normal(64 * 64)
        want = tile_vector(ol.stencil_apply(ol.make_desc(64, 64, 1, None, hop, self.MASS), v), L, 1)
        k = self.nrhs - 1                               # check the LAST right-hand side of the batch
        keep = self.rhs.to_host()
        probe = keep.copy()
        probe[k * vol:(k + 1) * vol] = tile_vector(v, L, 1)
        self.rhs.upload(probe)
        self.step()
        got = self.lhs.to_host()[k * vol:(k + 1) * vol]
        norm_dev = float(self.norms[self.rank * self.nrhs + k].item())
        self.rhs.upload(keep)
        err = float(np.linalg.norm(got - want) / np.linalg.norm(want))
        nerr = abs(norm_dev - float(np.vdot(want, want).real)) / float(np.vdot(want, want).real)
        if not (err < 1e-13 and nerr < 1e-12):
            raise SystemExit("staggered parity gate failed: rel L2 %.3e, norm %.3e" % (err, nerr))
        return err

    def free(self):
        for a in (self.hopping, self.rhs, self.lhs):
            a.free()


def timed(qmg, wl, steps, warmup, barrier):
    for _ in range(warmup):
        wl.step()
    qmg.sync()
    timer = qmg.Timer()
    barrier()
    qmg.sync()
    t0 = time.perf_counter()
    timer.start()
    for _ in range(steps):
        wl.step()
    ev_ms = timer.stop_ms()          # HIP events on the launch stream: synchronises on the stop event
    qmg.sync()
    barrier()
    wall = time.perf_counter() - t0
    return wall, ev_ms / steps


def cpu_baseline(fixture, budget_s=12.0):
    """Reference-structured CPU path (oracle, 1 thread) on a bounded sample of the same workload."""
    import oracle_lib as ol
    L = 1024
    gauge = tiled_gauge(L, fixture)
    clover, hopping = ol.wilson_fill(gauge, L, L)
    d = ol.make_desc(L, L, 2, clover, hopping, MASS)
    rng = np.random.default_rng(7)
    rhs = rng.standard_normal(2 * L * L) + 1j * rng.standard_normal(2 * L * L)
    t1 = ol.time_apply(d, rhs, ol.P_ALL | ol.P_ZERO, 1)
    reps = max(2, min(200, int(budget_s / max(t1, 1e-3))))
    t = ol.time_apply(d, rhs, ol.P_ALL | ol.P_ZERO, reps) / reps
    return {"value": FLOP_PER_SITE * L * L / t / 1e9, "unit": "GFLOP/s", "cores": 1, "kind": "port",
            "host_cores_available": os.cpu_count(),
            "sample": "Wilson apply_M, %dx%d tiled l64t64b60, %d applies, reference pass structure (8 cshift + 9 cMATxpy + 2 caxpy), g++ -O2" % (L, L, reps),
            "ms_per_apply": t * 1e3, "gb_per_s_algorithmic": BYTES_PER_SITE * L * L / t / 1e9}


def cpu_worker(L, budget_s):
    """`bench.py --cpu-worker L budget`: one oracle apply loop in its own process (no GPU, no torch); prints seconds per apply."""
    import oracle_lib as ol
    fixture = os.path.join(ROOT, "tests", "golden", "l64t64b60_heatbath.dat")
    gauge = tiled_gauge(L, fixture)
    clover, hopping = ol.wilson_fill(gauge, L, L)
    d = ol.make_desc(L, L, 2, clover, hopping, MASS)
    rng = np.random.default_rng(7 + os.getpid())
    rhs = rng.standard_normal(2 * L * L) + 1j * rng.standard_normal(2 * L * L)
    t1 = ol.time_apply(d, rhs, ol.P_ALL | ol.P_ZERO, 1)
    reps = max(2, min(200, int(budget_s / max(t1, 1e-3))))
    print(json.dumps({"s_per_apply": ol.time_apply(d, rhs, ol.P_ALL | ol.P_ZERO, reps) / reps, "reps": reps}), flush=True)


def cpu_baseline_all_cores(budget_s=10.0, L=1024):
    """SURVEY 8d "all host cores": one independent right-hand side per core (the reference has no threading, so the
    faithful many-core number is N copies of the 1-core run), each worker its own process applying the oracle to its own
    vector for ~budget_s.  Aggregate = sum of the workers' rates while all run together (shared memory bandwidth)."""
    import subprocess
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    n = max(1, min(ncores, 64))
    try:
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(L), str(budget_s)], stdout=subprocess.PIPE, text=True,
                                  env=dict(os.environ, OMP_NUM_THREADS="1")) for _ in range(n)]
        per = [json.loads(pr.communicate(timeout=300)[0].strip().splitlines()[-1])["s_per_apply"] for pr in procs]
        agg = sum(FLOP_PER_SITE * L * L / t for t in per) / 1e9
        return {"value": agg, "unit": "GFLOP/s", "cores": n, "kind": "port", "host_cores_visible": ncores,
                "sample": "%d concurrent 1-thread oracle processes, one right-hand side each, Wilson apply_M %dx%d, ~%.0f s each" % (n, L, L, budget_s),
                "gb_per_s_algorithmic": sum(BYTES_PER_SITE * L * L / t for t in per) / 1e9, "slowest_worker_ms_per_apply": max(per) * 1e3,
                "fastest_worker_ms_per_apply": min(per) * 1e3}
    except Exception as e:
        return {"error": repr(e)}


def kcycle_cpu_reference():
    """CPU timing beside the K-cycle rate: the GPU driver solves n13 at 256^2 (3 levels, nc = 8, l64 tiled) and
    dumps its null vectors and right-hand side; the 1-thread oracle K-cycle then solves the SAME system from the SAME null
    vectors.  Both rates are outer VPGCR iterations per second of solve time (setup excluded on both sides)."""
    import re
    import subprocess
    import tempfile
    import oracle_lib as ol
    drivers = os.path.join(ROOT, "quantum-mg_amd", "drivers")
    fixture = os.path.join(ROOT, "tests", "golden", "l64t64b60_heatbath.dat")
    L, n_refine, dof = 256, 2, 8
    try:
        with tempfile.TemporaryDirectory() as tmp:
            p = subprocess.run([os.path.join(drivers, "n13_wilson_kcycle"), str(L), str(MASS), "6.0", str(n_refine), str(dof), fixture, "64"], cwd=drivers,
                               env=dict(os.environ, QMG_QUIET="1", QMG_DUMP_DIR=tmp), capture_output=True, text=True, timeout=180)
            m = re.search(r"setup ([\d.e+-]+) s ; solve ([\d.e+-]+) s ; outer iterations/s ([\d.e+-]+)", p.stdout)
            git = int(re.search(r"Multigrid converged in (\d+) iterations", p.stdout).group(1))
            nullvecs = [np.fromfile(os.path.join(tmp, "nullvecs_level%d.bin" % l), dtype=np.complex128) for l in range(n_refine)]
            b = np.fromfile(os.path.join(tmp, "b.bin"), dtype=np.complex128)
        gauge = tiled_gauge(L, fixture)
        t0 = time.perf_counter()
        it, x, res, ops, its = ol.wilson_kcycle(L, MASS, n_refine, dof, gauge, nullvecs, b)
        t_all = time.perf_counter() - t0
        # the oracle call includes its (CPU) setup: time the setup alone with a zero-iteration solve and subtract
        t0 = time.perf_counter()
        ol.wilson_kcycle(L, MASS, n_refine, dof, gauge, nullvecs, b, max_iter=0)
        t_setup = time.perf_counter() - t0
        solve = max(t_all - t_setup, 1e-9)
        return {"workload": "n13 K-cycle 256x256 (l64 tiled), 3 levels, nc=8, same null vectors and rhs on both sides",
                "gpu_iterations": git, "gpu_solve_s": float(m.group(2)), "gpu_iterations_per_s": float(m.group(3)),
                "cpu_iterations": abs(it), "cpu_true_residual": res, "cpu_solve_s": solve, "cpu_setup_s": t_setup, "cpu_iterations_per_s": abs(it) / solve,
                "cpu": {"cores": 1, "kind": "port"}, "note": "256^2 is launch-latency-bound on the GPU (coarsest level 16^2); the BASELINE-size rate is also_kcycle.value"}
    except Exception as e:
        return {"error": repr(e)}


def kcycle_c5_schur_and_f32():
    """BASELINE configs[4] on one GPU: the adaptive n22 K-cycle at 4096^2 (4 levels, nc = 8, one adaptive pass) in the
    RED-BLACK (right-block-Jacobi Schur) form of n19, (i) all fp64 and (ii) with the K-cycle preconditioner in fp32
    (complex<float> hierarchy inside the fp64 outer VPGCR).  Residuals are true residuals of the ORIGINAL system."""
    import re
    import subprocess
    drivers = os.path.join(ROOT, "quantum-mg_amd", "drivers")
    exe = os.path.join(drivers, "n22_wilson_kcycle_adaptive")
    fixture = os.path.join(ROOT, "tests", "golden", "l64t64b60_heatbath.dat")
    try:
        p = subprocess.run([exe, "4096", str(MASS), "6.0", "3", "1", fixture, "64", "schur", "nrhs=1", "f32"], cwd=drivers, env=dict(os.environ, QMG_QUIET="1"),
                           capture_output=True, text=True, timeout=900)
        m = re.search(r"setup ([\d.e+-]+) s ; solve ([\d.e+-]+) s ; outer iterations/s ([\d.e+-]+)", p.stdout)
        it = re.search(r"Multigrid (converged|failed to converge) in (\d+) iterations", p.stdout)
        res = re.search(r"Check tolerance ([\d.e+-]+)", p.stdout)
        f = re.search(r"rhs 0 (converged|failed to converge) in (\d+) iterations ; alleged tolerance [\d.e+-]+ ; check tolerance ([\d.e+-]+)", p.stdout)
        ft = re.search(r"batched solve of 1 systems ([\d.e+-]+) s ; aggregate outer iterations/s ([\d.e+-]+)", p.stdout)
        al = re.findall(r"device allocator inside the (?:last )?solve ([\d.e+-]+) s in (\d+) calls", p.stdout)
        out = {"workload": "adaptive Wilson K-cycle (n22 parameters, 1 adaptive pass), 4096x4096, 4 levels, coarse nc=8, red-black (Schur) on every level, 1 GPU",
               "metric": "outer VPGCR iterations per second", "returncode": p.returncode,
               "fp64": {"value": float(m.group(3)), "outer_iterations": int(it.group(2)), "converged": it.group(1) == "converged",
                        "true_residual_original_system": float(res.group(1)), "solve_s": float(m.group(2)), "setup_s": float(m.group(1))}}
        if f and ft:
            out["fp32_kcycle"] = {"value": float(ft.group(2)), "outer_iterations": int(f.group(2)), "converged": f.group(1) == "converged",
                                  "true_residual_original_system": float(f.group(3)), "solve_s": float(ft.group(1)),
                                  "note": "K-cycle preconditioner entirely in complex<float> (vectors, matrices, null vectors); outer VPGCR, tolerance and residual check fp64"}
            out["fp32_over_fp64"] = out["fp32_kcycle"]["value"] / out["fp64"]["value"]
        if al:
            out["device_allocator_inside_solves_s"] = [float(a[0]) for a in al]
        out["fp64"]["note"] = ("vectors and arithmetic fp64; the Galerkin matrices, right-block-Jacobi hops and transfer null vectors of the preconditioner levels are STORED as complex<float> "
                               "(the facade's default for a hierarchy that only preconditions)")
        # the same fp64 solve with the reference's storage precision on every level, and with complex<half> storage (opt-in)
        for key, bits in (("fp64_strict_storage", "64"), ("fp64_16bit_storage", "16")):
            q = subprocess.run([exe, "4096", str(MASS), "6.0", "3", "1", fixture, "64", "schur"], cwd=drivers, env=dict(os.environ, QMG_QUIET="1", QMG_COARSE_BITS=bits),
                               capture_output=True, text=True, timeout=900)
            m2 = re.search(r"setup ([\d.e+-]+) s ; solve ([\d.e+-]+) s ; outer iterations/s ([\d.e+-]+)", q.stdout)
            it2 = re.search(r"Multigrid (converged|failed to converge) in (\d+) iterations", q.stdout)
            res2 = re.search(r"Check tolerance ([\d.e+-]+)", q.stdout)
            if m2 and it2 and res2:
                out[key] = {"value": float(m2.group(3)), "outer_iterations": int(it2.group(2)), "converged": it2.group(1) == "converged",
                            "true_residual_original_system": float(res2.group(1)), "solve_s": float(m2.group(2)), "QMG_COARSE_BITS": int(bits)}
        q = subprocess.run([exe, "4096", str(MASS), "6.0", "3", "1", fixture, "64", "schur", "nrhs=1", "f32"], cwd=drivers, env=dict(os.environ, QMG_QUIET="1", QMG_F16_COARSE="1"),
                           capture_output=True, text=True, timeout=900)
        f2 = re.search(r"rhs 0 (converged|failed to converge) in (\d+) iterations ; alleged tolerance [\d.e+-]+ ; check tolerance ([\d.e+-]+)", q.stdout)
        ft2 = re.search(r"batched solve of 1 systems ([\d.e+-]+) s ; aggregate outer iterations/s ([\d.e+-]+)", q.stdout)
        if f2 and ft2:
            out["fp32_kcycle_16bit_storage"] = {"value": float(ft2.group(2)), "outer_iterations": int(f2.group(2)), "converged": f2.group(1) == "converged",
                                                "true_residual_original_system": float(f2.group(3)), "solve_s": float(ft2.group(1)),
                                                "note": "the complex<float> K-cycle with the Galerkin levels' matrices and right-block-Jacobi hops stored as complex<half> (opt-in, QMG_F16_COARSE=1)"}
        if "fp32_kcycle" in out and "fp64_strict_storage" in out:
            out["fp32_over_fp64_strict_storage"] = out["fp32_kcycle"]["value"] / out["fp64_strict_storage"]["value"]
        return out
    except Exception as e:
        return {"error": repr(e)}


def f32_fine_apply(qmg, L, fixture, steps, warmup, barrier):
    """The fp32 instantiation of the headline kernel (BASELINE configs[4] "fp32"): same operator and lattice, complex<float>
    matrices and vectors, 192 B/site.  Parity gate: 5e-6 against the fp64 oracle through periodicity."""
    import oracle_lib as ol
    vol = L * L
    wl = Workload(qmg, L, fixture, 1337)
    c32, h32 = qmg.DeviceArray(4 * vol, np.complex64), qmg.DeviceArray(16 * vol, np.complex64)
    qmg.convert(c32, qmg.C32, wl.clover, qmg.C64, 4 * vol)
    qmg.convert(h32, qmg.C32, wl.hopping, qmg.C64, 16 * vol)
    qmg.sync()
    wl.free()
    ph = np.loadtxt(fixture)
    clover, hopping = ol.wilson_fill(ol.phases_to_gauge_u1(ph, 64, 64), 64, 64)
    clover, hopping = clover.astype(np.complex64).astype(np.complex128), hopping.astype(np.complex64).astype(np.complex128)
    rng = np.random.default_rng(1337)
    v = (rng.standard_normal(64 * 64 * 2) + 1j * rng.standard_normal(64 * 64 * 2)).astype(np.complex64).astype(np.complex128)
    want = tile_vector(ol.stencil_apply(ol.make_desc(64, 64, 2, clover, hopping, MASS), v), L, 2)
    rhs = qmg.DeviceArray.from_host(tile_vector(v, L, 2).astype(np.complex64))
    lhs = qmg.DeviceArray(2 * vol, np.complex64)
    d32 = qmg.make_desc(L, L, 2, c32, h32, MASS)

    class W:
        def step(self_inner):
            qmg.stencil_apply_t(qmg.C32, d32, lhs, rhs, qmg.P_ALL | qmg.P_ZERO)
    w = W()
    w.step()
    got = lhs.to_host().astype(np.complex128)
    err = float(np.linalg.norm(got - want) / np.linalg.norm(want))
    if not err < 5e-6:
        raise SystemExit("fp32 parity gate failed at L=%d: rel L2 error %.3e" % (L, err))
    wall, kern_ms = timed(qmg, w, steps, warmup, barrier)
    out = {"workload": "Wilson apply_stencil_2D_M, %dx%d, nc=2, complex<float> matrices and vectors (qmg_stencil_apply_t QMG_C32)" % (L, L), "dtype": "f32 (complex64)",
           "gflops": vol * FLOP_PER_SITE * steps / wall / 1e9, "ms_per_step": wall / steps * 1e3, "parity_gate_rel_l2_vs_fp64_oracle": err,
           "roofline": {"bound": "hbm", "achieved": 192 * vol / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": 192 * vol / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel": "k_stencil_site<1,1,true,false> (csrc/qmg_site.hip)",
                        "algorithmic_bytes_per_launch": 192 * vol, "avg_launch_ms": kern_ms, "note": "192 B/site (BASELINE.md); reported beside, never instead of, the fp64 line"}}
    # the same apply with the MATRICES stored in 16 bits (complex<half>; vectors complex<float>, fp32 arithmetic): 112 B/site.
    # Not a reference configuration (SURVEY 8f-4 "16-bit-storage smoother"): it serves the smoother inside the fp32 K-cycle.
    # Gate: against the fp64 oracle applied to the fp16-ROUNDED matrices, at fp32 accuracy.
    c16, h16 = qmg.DeviceArray(4 * vol, np.float32), qmg.DeviceArray(16 * vol, np.float32)
    qmg.convert_to_c16(c16, c32, qmg.C32, 4 * vol)
    qmg.convert_to_c16(h16, h32, qmg.C32, 16 * vol)
    d16 = qmg.make_desc(L, L, 2, c16, h16, MASS)

    def rounded(a):
        a = a.astype(np.complex64)
        return (a.real.astype(np.float16).astype(np.float64) + 1j * a.imag.astype(np.float16).astype(np.float64))
    want16 = tile_vector(ol.stencil_apply(ol.make_desc(64, 64, 2, rounded(clover), rounded(hopping), MASS), v), L, 2)

    class W16:
        def step(self_inner):
            qmg.stencil_apply_h16(d16, lhs, rhs, qmg.P_ALL | qmg.P_ZERO)
    w16 = W16()
    w16.step()
    err16 = float(np.linalg.norm(lhs.to_host().astype(np.complex128) - want16) / np.linalg.norm(want16))
    if not err16 < 5e-6:
        raise SystemExit("16-bit-matrix parity gate failed at L=%d: rel L2 error %.3e" % (L, err16))
    wall16, kern16 = timed(qmg, w16, steps, warmup, barrier)
    out["matrices_in_16_bit"] = {"ms_per_step": wall16 / steps * 1e3, "parity_gate_rel_l2_vs_fp64_oracle_on_rounded_matrices": err16,
                                 "roofline": {"bound": "hbm", "achieved": 112 * vol / (kern16 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": 112 * vol / (kern16 * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                              "kernel": "k_stencil_site<0,1,true,false>", "algorithmic_bytes_per_launch": 112 * vol, "avg_launch_ms": kern16,
                                              "note": "80 B/site of complex<half> matrices + 16 + 16 B/site of complex<float> vectors"}}
    for a in (c32, h32, c16, h16, rhs, lhs):
        a.free()
    return out


def wilson_from_links(qmg, L, fixture, steps, warmup, barrier):
    """The same operator and lattice as the headline, applied straight from the gauge links (qmg_wilson_apply_direct, csrc/qmg_wilson.hip):
    no stored matrices, 96 B/site in fp64 (links 32 + rhs 32 + lhs 32) and 48 B/site in fp32 -- the route Wilson2D's applies take inside
    the solvers.  Reported BESIDE the headline, which stays the reference's algorithm (a general stored stencil, 384 B/site).  Gates: the
    fp64 oracle on the stored stencil, through periodicity, 1e-13 (fp32: 5e-6)."""
    import oracle_lib as ol
    vol = L * L
    g = qmg.DeviceArray.from_host(tiled_gauge(L, fixture))
    g32 = qmg.DeviceArray(2 * vol, np.complex64)
    qmg.convert(g32, qmg.C32, g, qmg.C64, 2 * vol)
    ph = np.loadtxt(fixture)
    clover, hopping = ol.wilson_fill(ol.phases_to_gauge_u1(ph, 64, 64), 64, 64)
    rng = np.random.default_rng(1337)
    v = rng.standard_normal(64 * 64 * 2) + 1j * rng.standard_normal(64 * 64 * 2)
    want = tile_vector(ol.stencil_apply(ol.make_desc(64, 64, 2, clover, hopping, MASS), v), L, 2)
    d = qmg.make_desc(L, L, 2, None, None, MASS)
    out = {"workload": "Wilson apply_M straight from the U(1) links, %dx%d, nc=2 (no stored stencil)" % (L, L)}
    for name, dtype, gauge, np_t, bytes_site, tol in (("fp64", qmg.C64, g, np.complex128, 96, 1e-13), ("fp32", qmg.C32, g32, np.complex64, 48, 5e-6)):
        rhs = qmg.DeviceArray.from_host(tile_vector(v, L, 2).astype(np_t))
        lhs = qmg.DeviceArray(2 * vol, np_t)

        class W:
            def step(self_inner):
                qmg.wilson_apply_direct(dtype, d, gauge, lhs, rhs, qmg.P_ALL | qmg.P_ZERO)
        w = W()
        w.step()
        err = float(np.linalg.norm(lhs.to_host().astype(np.complex128) - want) / np.linalg.norm(want))
        if not err < tol:
            raise SystemExit("from-the-links parity gate failed (%s, L=%d): rel L2 error %.3e" % (name, L, err))
        wall, kern_ms = timed(qmg, w, steps, warmup, barrier)
        # (a 0.3 ms kernel: the rate is taken from the HIP-event time of the K launches; the wall clock around K = 20-200 of them
        # also holds a fixed 25-60 ms of host-side synchronisation that 1000 steps amortise -- measured, tools note in DESIGN.md 5)
        out[name] = {"gflops": vol * FLOP_PER_SITE / (kern_ms * 1e-3) / 1e9, "ms_per_step": kern_ms, "wall_ms_per_step": wall / steps * 1e3,
                     "parity_gate_rel_l2_vs_fp64_oracle": err,
                     "speedup_over_the_stored_stencil_bytes": (384 if name == "fp64" else 192) / bytes_site,
                     "roofline": {"bound": "hbm", "achieved": bytes_site * vol / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": bytes_site * vol / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                  "kernel": "k_wilson_pair2<%s,true> (kernel W2, both parities x two rows per lane group)" % ("double" if name == "fp64" else "float"),
                                  "algorithmic_bytes_per_launch": bytes_site * vol, "avg_launch_ms": kern_ms,
                                  "note": "%d B/site: links %d (each link serves the two sites it joins) + rhs + lhs" % (bytes_site, bytes_site // 3)}}
        rhs.free()
        lhs.free()
    g.free()
    g32.free()
    return out


def staggered_8rhs(qmg, L, fixture, steps, warmup, barrier, torch):
    """BASELINE configs[3] per-GPU workload on this one GPU: staggered 4096^2, 8 right-hand sides sharing one matrix read,
    per-RHS |lhs|^2 from the same pass (qmg_stencil_apply_norm2), and the (here one-rank) all-reduce slot buffer."""
    wl = StaggeredMultiRHS(qmg, L, fixture, 1337, 8, 0, 1, None, torch)
    gate = wl.parity_gate(fixture)
    wall, kern_ms = timed(qmg, wl, steps, warmup, barrier)
    sites_rhs = L * L * 8
    b = wl.bytes_per_site_rhs * sites_rhs
    out = {"workload": "staggered apply + per-RHS norm2sq, %dx%d, nc=1, 8 rhs sharing one read of the hopping matrices (BASELINE configs[3] per-GPU share)" % (L, L),
           "gflops": sites_rhs * wl.FLOP_PER_SITE_RHS * steps / wall / 1e9, "ms_per_step": wall / steps * 1e3, "achieved_gb_per_s": b / (kern_ms * 1e-3) / 1e9,
           "frac_of_hbm_peak": b / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "parity_gate_rel_l2": gate,
           "note": "whole step in one pass: apply with fused per-RHS norms (64/nrhs + 32 B/site/rhs; a separate norm2sq would re-read 16); the N-GPU form is `--workload staggered`"}
    wl.free()
    return out


def slab_solve(L, world, rank):
    """ONE Wilson system on ONE L x L lattice, strong-scaled over the ranks by y-slabs (drivers/slab_wilson_solve.cpp:
    halo rows over RCCL send/recv overlapped with the interior apply, reductions summed over ranks, BiCGStab-6 of krylov.hpp
    unchanged).  EVERY rank starts its child (the children build their own communicator: qmg_comm_init_env over
    MASTER_ADDR : MASTER_PORT + 1); rank 0's child reports.  The same lattice, source and tolerance at every N, so
    `solve_s` across the driver's N = 1, 2, 4, 8 runs is the strong-scaling curve and `x_norm2` must agree between them."""
    import re
    import subprocess
    drivers = os.path.join(ROOT, "quantum-mg_amd", "drivers")
    exe = os.path.join(drivers, "slab_wilson_solve")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", drivers, "-j4"], stdout=subprocess.DEVNULL)
    try:
        p = subprocess.run([exe, str(L), "0.05", "6.0", "200", "1337", "1e-10", "1", "1"], cwd=drivers, env=dict(os.environ, QMG_QUIET="1"),
                           capture_output=True, text=True, timeout=180)
    except subprocess.TimeoutExpired:
        return {"error": "slab_wilson_solve timed out after 180 s on rank %d" % rank}
    m = re.search(r"BiCGStab-6 (converged|FAILED) in (\d+) iterations, ([\d.e+-]+) s, (\d+) applies, true relative residual ([\d.e+-]+), \|b\| [\d.e+-]+, \|x\|\^2 ([\d.e+-]+), world (\d+)", p.stdout)
    a = re.search(r"apply_M on a slab: ([\d.e+-]+) ms with the exchange overlapped, ([\d.e+-]+) ms serialised", p.stdout)
    v = re.findall(r"slab apply vs single-domain apply, rel diff ([\d.e+-]+) \((ok|MISMATCH)\)", p.stdout)
    if rank != 0:
        return {"rank": rank, "rc": p.returncode} if p.returncode == 0 else {"error": "rc %d on rank %d" % (p.returncode, rank)}
    if not m or p.returncode != 0:
        return {"error": "rc %d" % p.returncode, "tail": (p.stdout + p.stderr)[-600:]}
    return {"workload": "one Wilson solve (BiCGStab-6, tol 1e-10, mass 0.05, beta 6.0 device heatbath) on ONE %dx%d lattice cut into %d y-slab(s)" % (L, L, world),
            "scaling": "strong", "world": int(m.group(7)), "converged": m.group(1) == "converged", "iterations": int(m.group(2)), "solve_s": float(m.group(3)),
            "applies": int(m.group(4)), "true_rel_residual": float(m.group(5)), "x_norm2": float(m.group(6)),
            "slab_apply_ms_exchange_overlapped": float(a.group(1)) if a else None, "slab_apply_ms_exchange_serialised": float(a.group(2)) if a else None,
            "slab_apply_equals_single_domain_apply_on_rank0": bool(v) and v[0][1] == "ok",
            "note": "each rank also builds the single-domain operator once and checks its slab apply (real exchange) against its rows of it"}


def slab_kcycle(L, world, rank):
    """SURVEY 8f-4, the multigrid half: the n13 K-cycle (3 levels, 4x4 blocks, coarse nc = 8, the reference's constants) with ONE L x L lattice
    cut into `world` y-slabs on EVERY level (drivers/n13_wilson_kcycle_slab.cpp).  Every rank starts its child; rank 0's child reports.  The
    decomposed run draws the single-domain run's random vectors, so `outer_iterations` and `x_norm2` must agree across the driver's
    N = 1, 2, 4, 8 runs (to the rounding of the fp32 setup) and `solve_s` / `setup_s` are the strong-scaling curves."""
    import re
    import subprocess
    drivers = os.path.join(ROOT, "quantum-mg_amd", "drivers")
    exe = os.path.join(drivers, "n13_wilson_kcycle_slab")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", drivers, "-j4"], stdout=subprocess.DEVNULL)
    fixture = os.path.join(ROOT, "tests", "golden", "l64t64b60_heatbath.dat")
    try:
        env = dict(os.environ, QMG_QUIET="1")
        if "MASTER_PORT" in os.environ:   # its own rendezvous port (the slab_solve children used MASTER_PORT + 1 a moment ago)
            env["QMG_COMM_PORT"] = str(int(os.environ["MASTER_PORT"]) + 2)
        p = subprocess.run([exe, str(L), str(MASS), "6.0", "2", "8", fixture, "64"], cwd=drivers, env=env, capture_output=True, text=True, timeout=180)
    except subprocess.TimeoutExpired:
        return {"error": "n13_wilson_kcycle_slab timed out after 180 s on rank %d" % rank}
    if rank != 0:
        return {"rank": rank, "rc": p.returncode} if p.returncode == 0 else {"error": "rc %d on rank %d" % (p.returncode, rank)}
    m = re.search(r"Multigrid (converged|failed to converge) in (\d+) iterations", p.stdout)
    c = re.search(r"Check tolerance ([\d.e+-]+)", p.stdout)
    sl = re.search(r"\[QMG-SLAB\]: world (\d+) ; \|b\| ([\d.e+-]+) ; \|x\|\^2 ([\d.e+-]+)", p.stdout)
    t = re.search(r"\[QMG-TIMING\]: setup ([\d.e+-]+) s ; solve ([\d.e+-]+) s ; outer iterations/s ([\d.e+-]+)", p.stdout)
    if not (m and c and sl and t) or p.returncode != 0:
        return {"error": "rc %d" % p.returncode, "tail": (p.stdout + p.stderr)[-600:]}
    return {"workload": "Wilson K-cycle (n13 parameters), %dx%d, 3 levels, coarse nc=8, fp64 (Galerkin matrices stored as complex<float>, as on one domain), ONE lattice cut into %d y-slab(s) on every level" % (L, L, world),
            "scaling": "strong", "world": int(sl.group(1)), "converged": m.group(1) == "converged", "outer_iterations": int(m.group(2)),
            "true_residual": float(c.group(1)), "x_norm2": float(sl.group(3)), "setup_s": float(t.group(1)), "solve_s": float(t.group(2)),
            "outer_iterations_per_s": float(t.group(3))}


def slab_kcycle_c5(world, rank):
    """BASELINE configs[4] (adaptive Wilson K-cycle, 4096^2, 4 levels, coarse nc = 8, red-black on every level) with the ONE lattice cut into
    `world` y-slabs on every level (drivers/n22_wilson_kcycle_adaptive.cpp in slab mode).  As `also_slab_kcycle`: every rank starts its child,
    rank 0's reports; iterations and `x_norm2` must agree across N, `setup_s` / `solve_s` are the strong-scaling curves."""
    import re
    import subprocess
    drivers = os.path.join(ROOT, "quantum-mg_amd", "drivers")
    exe = os.path.join(drivers, "n22_wilson_kcycle_adaptive")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", drivers, "-j4"], stdout=subprocess.DEVNULL)
    fixture = os.path.join(ROOT, "tests", "golden", "l64t64b60_heatbath.dat")
    env = dict(os.environ, QMG_QUIET="1", QMG_SLAB="1")
    if "MASTER_PORT" in os.environ:
        env["QMG_COMM_PORT"] = str(int(os.environ["MASTER_PORT"]) + 3)
    try:
        # one child: the fp64 solve, then (`nrhs=1 f32`) the same system again with the K-cycle preconditioner in complex<float>
        p = subprocess.run([exe, "4096", str(MASS), "6.0", "3", "1", fixture, "64", "schur", "nrhs=1", "f32"], cwd=drivers, env=env, capture_output=True, text=True, timeout=180)
        stdout, stderr, rc, timed_out = p.stdout, p.stderr, p.returncode, False
    except subprocess.TimeoutExpired as e:   # keep what the fp64 part printed before the fp32 part ran out of time
        dec = lambda b: b.decode(errors="replace") if isinstance(b, bytes) else (b or "")
        stdout, stderr, rc, timed_out = dec(e.stdout), dec(e.stderr), 0, True
    if rank != 0:
        if timed_out:
            return {"error": "n22_wilson_kcycle_adaptive (slab mode) timed out after 180 s on rank %d" % rank}
        return {"rank": rank, "rc": rc} if rc == 0 else {"error": "rc %d on rank %d" % (rc, rank)}
    m = re.search(r"Multigrid (converged|failed to converge) in (\d+) iterations", stdout)
    c = re.search(r"Check tolerance ([\d.e+-]+)", stdout)
    sl = re.search(r"\[QMG-SLAB\]: world (\d+) ; \|b\| ([\d.e+-]+) ; \|x\|\^2 ([\d.e+-]+)", stdout)
    t = re.search(r"\[QMG-TIMING\]: setup ([\d.e+-]+) s ; solve ([\d.e+-]+) s ; outer iterations/s ([\d.e+-]+)", stdout)
    if not (m and c and sl and t) or rc != 0:
        return {"error": "timed out after 180 s" if timed_out else "rc %d" % rc, "tail": (stdout + stderr)[-600:]}
    out = {"workload": "adaptive Wilson K-cycle (n22 parameters, 1 adaptive pass), 4096x4096, 4 levels, coarse nc=8, red-black on every level, fp64, ONE lattice cut into %d y-slab(s)" % world,
           "scaling": "strong", "world": int(sl.group(1)), "converged": m.group(1) == "converged", "outer_iterations": int(m.group(2)),
           "true_residual_original_system": float(c.group(1)), "x_norm2": float(sl.group(3)), "setup_s": float(t.group(1)), "solve_s": float(t.group(2)),
           "outer_iterations_per_s": float(t.group(3))}
    f = re.search(r"\[QMG-MRHS\]: rhs 0 converged in (\d+) iterations ; alleged tolerance ([\d.e+-]+) ; check tolerance ([\d.e+-]+)", stdout)
    ft = re.search(r"batched solve of 1 systems ([\d.e+-]+) s ; aggregate outer iterations/s ([\d.e+-]+)", stdout)
    if f and ft:   # BASELINE configs[4] as written (fp32): the K-cycle in complex<float> inside the fp64 outer solve, same slabs
        out["fp32_kcycle"] = {"outer_iterations": int(f.group(1)), "true_residual_original_system": float(f.group(3)), "solve_s": float(ft.group(1)),
                              "outer_iterations_per_s": float(ft.group(2))}
    elif timed_out:
        out["fp32_kcycle"] = {"error": "the fp32 part of the child did not finish within the child's 180 s"}
        out["error_fp32"] = True
    return out


def pmc_traffic(L):
    """HBM bytes per launch of the headline kernel from the committed rocprofv3 PMC passes, ONLY if they were taken on the
    kernel source that is being run now (sha256 of csrc/qmg_stencil.hip stamped by tools/summarize_profiles.py)."""
    import hashlib
    src = os.path.join(ROOT, "quantum-mg_amd", "csrc", "qmg_stencil.hip")
    sha = hashlib.sha256(open(src, "rb").read()).hexdigest()
    for tag in ("r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % tag)
        if L == 4096 and os.path.exists(path):
            pmc = json.load(open(path))
            if pmc.get("kernel_source_sha256") == sha:
                return pmc["hbm_traffic_bytes_per_launch"], "profiles/%s_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 gfx950 correction; taken at git %s on this kernel source)" % (tag, pmc.get("git_head", "?"))
            return None, "profiles/%s_pmc_traffic.json is STALE: csrc/qmg_stencil.hip has changed since the PMC passes (sha256 mismatch); re-run tools/summarize_profiles.py" % tag
    return None, "no PMC passes committed for this configuration"


def kcycle_c3_16bit():
    """`also_kcycle` with QMG_COARSE_BITS=16: the Galerkin matrices of the preconditioner levels stored as complex<half> (opt-in; a quarter of
    the fp64 matrix stream; vectors, arithmetic and the outer solve fp64)."""
    out = kcycle_c3(extra_env={"QMG_COARSE_BITS": "16"})
    if "workload" in out:
        out["workload"] = out["workload"].replace("complex<float> (the facade's default)", "complex<half> (opt-in, QMG_COARSE_BITS=16)")
    return out


def kcycle_c3_strict_fp64():
    """`also_kcycle` with QMG_COARSE_F32=0: the Galerkin coarse matrices stay complex<double> (the reference's storage precision on
    every level).  Reported beside the default, in which the preconditioner levels STORE their matrices as complex<float>
    (vectors, shifts, arithmetic and the outer solve fp64; same outer iterations and true residual)."""
    out = kcycle_c3(extra_env={"QMG_COARSE_F32": "0"})
    if "workload" in out:
        out["workload"] += "; Galerkin matrices stored as complex<double> on every level (QMG_COARSE_F32=0)"
    return out


def kcycle_c3(extra_env=None):
    """Second half of the BASELINE metric: K-cycle outer iterations per second on BASELINE configs[2]
    (Wilson 2048^2, 3 levels 2048^2 -> 512^2 -> 128^2, coarse.h nc = 24, n13 parameters) through the C++ facade
    driver (product path; every step a HIP kernel).  Setup (null vectors, block-ortho, Galerkin builds) is timed
    separately by the driver and is not part of the rate."""
    import re
    import subprocess
    drivers = os.path.join(ROOT, "quantum-mg_amd", "drivers")
    exe = os.path.join(drivers, "n13_wilson_kcycle")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", drivers, "-j4"], stdout=subprocess.DEVNULL)
    fixture = os.path.join(ROOT, "tests", "golden", "l64t64b60_heatbath.dat")
    try:
        p = subprocess.run([exe, "2048", str(MASS), "6.0", "2", "24", fixture, "64"], cwd=drivers, env=dict(os.environ, QMG_QUIET="1", **(extra_env or {})),
                           capture_output=True, text=True, timeout=600)
        m = re.search(r"setup ([\d.e+-]+) s ; solve ([\d.e+-]+) s ; outer iterations/s ([\d.e+-]+)", p.stdout)
        it = re.search(r"Multigrid (converged|failed to converge) in (\d+) iterations", p.stdout)
        res = re.search(r"Check tolerance ([\d.e+-]+)", p.stdout)
        ops = re.findall(r"Level (\d) .* Total (\d+)", p.stdout)
        al = re.search(r"device allocator inside the solve ([\d.e+-]+) s in (\d+) calls", p.stdout)
        f32c = "complex<float>" in p.stdout or "complex<half>" in p.stdout
        return {"workload": "Wilson K-cycle (n13 parameters), 2048x2048, 3 levels, coarse nc=24, fp64, 1 GPU" + ("; Galerkin matrices of the preconditioner levels and the null vectors of the K-cycle's transfers stored as complex<float> (the facade's default), arithmetic and vectors fp64" if f32c else ""),
                "metric": "outer VPGCR iterations per second",
                "value": float(m.group(3)), "outer_iterations": int(it.group(2)), "converged": it.group(1) == "converged",
                "true_residual": float(res.group(1)), "solve_s": float(m.group(2)), "setup_s": float(m.group(1)),
                "operator_applies_per_level": {l: int(t) for l, t in ops}, "returncode": p.returncode,
                "device_allocator_inside_solve_s": float(al.group(1)) if al else None}
    except Exception as e:   # the headline number must not be lost to a problem in the extra measurement
        return {"error": repr(e)}


def kcycle_c3_batched(nrhs=8, f32=False, extra_env=None):
    """The same solve for a lock-step batch of `nrhs` independent right-hand sides on the one GPU (include/qmg/batch.hpp):
    coarse operators / null vectors streamed once per step for the batch, coarse applies on the f64 matrix cores.
    `value` is the aggregate over the batch (sum of the systems' outer iterations / wall).
    f32: the K-cycle preconditioner entirely in complex<float> (vectors, matrices, null vectors) inside the fp64 outer VPGCR, which
    still converges to 1e-10 in fp64 (mg_preconditioner_batch_mixed)."""
    import re
    import subprocess
    drivers = os.path.join(ROOT, "quantum-mg_amd", "drivers")
    exe = os.path.join(drivers, "n13_wilson_kcycle_mrhs")
    fixture = os.path.join(ROOT, "tests", "golden", "l64t64b60_heatbath.dat")
    try:
        if not os.path.exists(exe):
            subprocess.check_call(["make", "-C", drivers, "-j4"], stdout=subprocess.DEVNULL)
        p = subprocess.run([exe, "2048", str(MASS), "6.0", "2", "24", fixture, "64", str(nrhs)] + (["f32"] if f32 else []), cwd=drivers,
                           env=dict(os.environ, QMG_QUIET="1", **(extra_env or {})), capture_output=True, text=True, timeout=600)
        m = re.search(r"setup ([\d.e+-]+) s ; batched solve of (\d+) systems ([\d.e+-]+) s ; aggregate outer iterations/s ([\d.e+-]+) ; systems/s ([\d.e+-]+)", p.stdout)
        rows = re.findall(r"rhs (\d+) (converged|failed to converge) in (\d+) iterations ; alleged tolerance [\d.e+-]+ ; check tolerance ([\d.e+-]+)", p.stdout)
        return {"workload": "Wilson K-cycle (n13 parameters), 2048x2048, 3 levels, coarse nc=24, %s, %d independent right-hand side%s%s on 1 GPU"
                            % ("K-cycle in complex<float> inside the fp64 outer solve" if f32 else "fp64", nrhs, "s" if nrhs > 1 else "", " in lock step" if nrhs > 1 else ""),
                "metric": "aggregate outer VPGCR iterations per second", "value": float(m.group(4)), "systems_per_s": float(m.group(5)),
                "solve_s": float(m.group(3)), "setup_s": float(m.group(1)), "nrhs": nrhs,
                "outer_iterations": [int(r[2]) for r in rows], "all_converged": all(r[1] == "converged" for r in rows) and len(rows) == nrhs,
                "worst_true_residual": max(float(r[3]) for r in rows), "returncode": p.returncode}
    except Exception as e:
        return {"error": repr(e)}


def kcycle_c5_shape():
    """BASELINE configs[4] shape on one GPU in fp64: the adaptive n22 K-cycle, 4096^2 -> 1024^2 -> 256^2 -> 64^2, nc = 8,
    one adaptive pass, the reference's ORIGINAL-operator mode (its fp32 / red-black / 8-GPU qualifiers are not built)."""
    import re
    import subprocess
    drivers = os.path.join(ROOT, "quantum-mg_amd", "drivers")
    exe = os.path.join(drivers, "n22_wilson_kcycle_adaptive")
    fixture = os.path.join(ROOT, "tests", "golden", "l64t64b60_heatbath.dat")
    try:
        if not os.path.exists(exe):
            subprocess.check_call(["make", "-C", drivers, "-j4"], stdout=subprocess.DEVNULL)
        p = subprocess.run([exe, "4096", str(MASS), "6.0", "3", "1", fixture, "64"], cwd=drivers, env=dict(os.environ, QMG_QUIET="1"),
                           capture_output=True, text=True, timeout=600)
        m = re.search(r"setup ([\d.e+-]+) s ; solve ([\d.e+-]+) s ; outer iterations/s ([\d.e+-]+)", p.stdout)
        it = re.search(r"Multigrid (converged|failed to converge) in (\d+) iterations", p.stdout)
        res = re.search(r"Check tolerance ([\d.e+-]+)", p.stdout)
        return {"workload": "adaptive Wilson K-cycle (n22 parameters, 1 adaptive pass), 4096x4096, 4 levels, coarse nc=8, fp64, 1 GPU",
                "metric": "outer VPGCR iterations per second", "value": float(m.group(3)), "outer_iterations": int(it.group(2)),
                "converged": it.group(1) == "converged", "true_residual": float(res.group(1)), "solve_s": float(m.group(2)), "setup_s": float(m.group(1)),
                "returncode": p.returncode}
    except Exception as e:
        return {"error": repr(e)}


def self_launch(n, argv):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): THIS process becomes the launcher.  It starts
    N fresh children -- one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run would
    set them -- before anything here has touched the GPU (no torch import, no HIP call: a process that has initialised HIP is
    never re-executed), forwards rank 0's JSON line and returns the worst exit code.  A child that dies takes the others with
    it (exact PIDs), so a failed rank cannot leave the rest waiting in a collective."""
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0", LOCAL_WORLD_SIZE=str(n))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0 = ""
    rc = 0
    try:
        out0, _ = procs[0].communicate()
        for pr in procs:
            pr.wait()
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
                pr.wait()
    for pr in procs:
        if pr.returncode != 0:
            rc = pr.returncode if pr.returncode > 0 else 1
    sys.stdout.write(out0)
    sys.stdout.flush()
    return rc


def launch_check(rank, world):
    """Smallest end-to-end exercise of the launch path: every rank joins the process group the launcher's environment describes
    (RCCL when a GPU is visible, gloo otherwise), ONE sum all-reduce of a one per rank, rank 0 prints what the collective saw.
    `collective_ranks` == n_gpus proves that `world` ranks took part in a real collective.  CPU test: tests/test_distributed_cpu.py."""
    import torch
    import torch.distributed as dist
    on_gpu = torch.cuda.is_available()
    if on_gpu:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
    else:
        dist.init_process_group("gloo")
    t = torch.ones(1, dtype=torch.float64, device="cuda" if on_gpu else "cpu")
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "collective_ranks": int(t.item()), "backend": dist.get_backend(),
                          "group_world_size": dist.get_world_size()}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--L", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true")
    ap.add_argument("--no-kcycle", action="store_true")
    ap.add_argument("--workload", choices=["wilson", "staggered", "kcycle"], default="wilson",
                    help="wilson: the headline fine Wilson apply (default); staggered: BASELINE configs[3], 8 rhs per GPU + one all-reduce per step; "
                         "kcycle: BASELINE configs[2] K-cycle, --nrhs independent systems per GPU in lock step, right-hand sides sharded over ranks (no collective)")
    ap.add_argument("--nrhs", type=int, default=8)
    ap.add_argument("--cpu-worker", nargs=2, metavar=("L", "BUDGET_S"), help="internal: one CPU-oracle apply loop (cpu_baseline_all_cores)")
    ap.add_argument("--launch-check", action="store_true", help="only the launch path: N ranks join one process group (RCCL on GPUs, gloo on a CPU box), one all-reduce, one JSON line")
    args = ap.parse_args()
    if args.cpu_worker:
        cpu_worker(int(args.cpu_worker[0]), float(args.cpu_worker[1]))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:   # no launcher around us: be the launcher (before any GPU use)
        import sys
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.launch_check:
        launch_check(rank, world)
        return

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    qmg = importlib.import_module("quantum-mg_amd")
    if not os.path.exists(qmg.SO_PATH):
        qmg.build()
    qmg.init(local_rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fixture = os.path.join(ROOT, "tests", "golden", "l64t64b60_heatbath.dat")
    L = args.L
    sharding = importlib.import_module("quantum-mg_amd.sharding")
    if args.workload == "kcycle":
        # every rank runs the batched n13 driver on its own GPU (the child takes LOCAL_RANK for the device and RANK for its
        # share of the right-hand sides); one step = one complete solve of the rank's batch; setup is outside the rate
        import re
        import subprocess
        drivers = os.path.join(ROOT, "quantum-mg_amd", "drivers")
        exe = os.path.join(drivers, "n13_wilson_kcycle_mrhs")
        if not os.path.exists(exe):
            subprocess.check_call(["make", "-C", drivers, "-j4"], stdout=subprocess.DEVNULL)
        barrier()
        p = subprocess.run([exe, "2048", str(MASS), "6.0", "2", "24", fixture, "64", str(args.nrhs)], cwd=drivers, env=dict(os.environ, QMG_QUIET="1"),
                           capture_output=True, text=True, timeout=900)
        m = re.search(r"setup ([\d.e+-]+) s ; batched solve of (\d+) systems ([\d.e+-]+) s ; aggregate outer iterations/s ([\d.e+-]+)", p.stdout)
        rows = re.findall(r"rhs (\d+) (converged|failed to converge) in (\d+) iterations ; alleged tolerance [\d.e+-]+ ; check tolerance ([\d.e+-]+)", p.stdout)
        ok_run = bool(m) and len(rows) == args.nrhs and all(r[1] == "converged" for r in rows) and p.returncode == 0
        solve_s = float(m.group(3)) if m else float("inf")
        iters = sum(int(r[2]) for r in rows)
        wall = sharding.max_over_ranks(solve_s, dist, "cuda")
        total_iters = iters
        if dist is not None:
            t = torch.tensor([float(iters), 1.0 if ok_run else 0.0], device="cuda", dtype=torch.float64)
            dist.all_reduce(t)
            total_iters, ok_all = t[0].item(), t[1].item() == world
        else:
            ok_all = ok_run
        out = {"metric": "K-cycle outer iterations per second (aggregate over systems and GPUs)", "value": total_iters / wall, "unit": "iterations/s",
               "n_gpus": world, "steps": 1, "warmup": 0, "ms_per_step": wall * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f64 (complex128)", "data": "synthetic",
               "config": {"workload": "Wilson K-cycle (n13 parameters), 2048x2048, 3 levels, coarse nc=24, %d systems per GPU in lock step (%d total)" % (args.nrhs, world * args.nrhs),
                          "lattice": [2048, 2048], "nc": 2, "coarse_nc": 24, "mass": MASS, "rhs_per_gpu": args.nrhs,
                          "parallelism": "right-hand sides sharded over ranks, batched within a rank, no collective in the solve"},
               "all_converged": ok_all, "worst_true_residual_rank0": max([float(r[3]) for r in rows]) if rows else None,
               "setup_s_rank0": float(m.group(1)) if m else None}
        if rank == 0:
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.workload == "staggered":
        wl = StaggeredMultiRHS(qmg, L, fixture, 1337 + rank, args.nrhs, rank, world, dist, torch)
        gate_err = wl.parity_gate(fixture)
        wall, kern_ms = timed(qmg, wl, args.steps, args.warmup, barrier)
        wall = sharding.max_over_ranks(wall, dist, "cuda")
        sites_rhs = L * L * args.nrhs
        out = {"metric": "staggered multi-RHS Dslash throughput", "value": world * sites_rhs * wl.FLOP_PER_SITE_RHS * args.steps / wall / 1e9,
               "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 (complex128)", "data": "synthetic",
               "config": {"workload": "staggered apply + per-RHS norm2sq + one all-reduce, %dx%d U(1) tiled, nc=1, %d rhs per GPU (%d total)" % (L, L, args.nrhs, world * args.nrhs),
                          "lattice": [L, L], "nc": 1, "mass": wl.MASS, "rhs_per_gpu": args.nrhs, "parallelism": "rhs sharded over ranks, 1 all-reduce of %d doubles per step" % (world * args.nrhs)},
               "roofline": {"bound": "hbm", "achieved": wl.bytes_per_site_rhs * sites_rhs / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": wl.bytes_per_site_rhs * sites_rhs / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                            "kernel": "k_stencil_pair<double,1,2,NORM,PF> (nrhs loop, fused norms) + k_apply_norm_final",
                            "note": "whole step in one pass: apply with fused per-RHS norms, 64/nrhs + 32 B/site/rhs (a separate norm2sq would re-read 16)"},
               "parity_gate_rel_l2": gate_err}
        if rank == 0:
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    wl = Workload(qmg, L, fixture, seed=1337 + rank)
    gate_err = wl.parity_gate(fixture)
    wall, kern_ms = timed(qmg, wl, args.steps, args.warmup, barrier)

    wall = sharding.max_over_ranks(wall, dist, "cuda")
    sites = L * L
    value = world * sites * FLOP_PER_SITE * args.steps / wall / 1e9
    achieved = BYTES_PER_SITE * sites / (kern_ms * 1e-3) / 1e9

    out = {
        "metric": "fine Wilson stencil apply throughput", "value": value, "unit": "GFLOP/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 (complex128)", "data": "synthetic",
        "config": {"workload": "Wilson apply_stencil_2D_M, %dx%d U(1) (l64t64b60 tiled), nc=2, fp64, 1 rhs per GPU" % (L, L),
                   "lattice": [L, L], "nc": 2, "mass": MASS, "rhs_per_gpu": 1, "parallelism": "independent rhs per GPU, no collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "kernel": "k_stencil_pair<double,2,2>", "algorithmic_bytes_per_launch": BYTES_PER_SITE * sites,
                     "avg_launch_ms": kern_ms},
        "hbm_gb_per_s_aggregate": world * BYTES_PER_SITE * sites * args.steps / wall / 1e9,
        "parity_gate_rel_l2": gate_err,
    }
    # HBM traffic per launch from the committed rocprofv3 PMC passes of this same command (profiles/), collected separately
    # because counters cannot be read from inside the timed run -- and only if they belong to the kernel source in use
    out["roofline"]["traffic"], out["roofline"]["traffic_source"] = pmc_traffic(L)
    wl.free()

    if world > 1 and dist is not None:
        # (i) proof that `world` ranks met in a real RCCL collective; (ii) BASELINE configs[3]'s step on every rank: 8 right-hand sides
        # per GPU, per-RHS norms from the apply pass, ONE all-reduce of world x 8 doubles per step (north_star: "a single RCCL all-reduce
        # per global reduction over xGMI") -- the default workload above has no collective by construction
        ones = torch.ones(1, dtype=torch.float64, device="cuda")
        dist.all_reduce(ones)
        out["collective"] = {"backend": dist.get_backend(), "is_rccl": True, "group_world_size": dist.get_world_size(), "ranks_seen_by_allreduce": int(ones.item())}
        if not args.no_also:
            swl = StaggeredMultiRHS(qmg, 4096, fixture, 1337 + rank, 8, rank, world, dist, torch)
            sgate = swl.parity_gate(fixture)
            ssteps = max(10, args.steps // 4)
            swall, skern = timed(qmg, swl, ssteps, args.warmup, barrier)
            swall = sharding.max_over_ranks(swall, dist, "cuda")
            srhs = 4096 * 4096 * 8
            norms_host = swl.norms.cpu().numpy()
            out["also_staggered_allreduce"] = {
                "workload": "BASELINE configs[3]: staggered 4096x4096, %d right-hand sides = 8 per GPU x %d GPUs, apply + per-RHS norms + ONE all-reduce of %d doubles per step" % (8 * world, world, 8 * world),
                "gflops_aggregate": world * srhs * swl.FLOP_PER_SITE_RHS * ssteps / swall / 1e9, "ms_per_step": swall / ssteps * 1e3, "steps": ssteps,
                "kernel_ms_rank0": skern, "frac_of_hbm_peak_rank0": swl.bytes_per_site_rhs * srhs / (skern * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "parity_gate_rel_l2": sgate, "scaling": "weak",
                "every_rank_sees_every_norm": bool((norms_host > 0).all()) and len(norms_host) == 8 * world}
            swl.free()

    if rank == 0 and world == 1 and not args.no_also and L != 2048:
        wl2 = Workload(qmg, 2048, fixture, seed=1337)
        e2 = wl2.parity_gate(fixture)
        # one discarded pass first (the first pass after freeing the 4096^2 workload's 6.5 GB occasionally carries a one-off
        # 30-45 ms stall of the runtime inside the wall-clock window), then ONE timed pass, measured as the headline is
        timed(qmg, wl2, args.steps, args.warmup, barrier)
        w2, k2 = timed(qmg, wl2, args.steps, args.warmup, barrier)
        s2 = 2048 * 2048
        out["also"] = {"workload": "Wilson apply, 2048x2048 (BASELINE configs[1])", "gflops": s2 * FLOP_PER_SITE * args.steps / w2 / 1e9,
                       "ms_per_step": w2 / args.steps * 1e3, "achieved_gb_per_s": BYTES_PER_SITE * s2 / (k2 * 1e-3) / 1e9,
                       "frac_of_hbm_peak": BYTES_PER_SITE * s2 / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS, "parity_gate_rel_l2": e2,
                       "note": "working set 1.6 GB; the 128 MiB vectors partly live in the 256 MiB Infinity Cache"}
        wl2.free()

    if rank == 0 and world == 1 and not args.no_also:
        out["also_f32"] = f32_fine_apply(qmg, L, fixture, args.steps, args.warmup, barrier)
        out["also_wilson_from_links"] = wilson_from_links(qmg, L, fixture, args.steps, args.warmup, barrier)
        out["also_staggered_8rhs"] = staggered_8rhs(qmg, 4096, fixture, max(10, args.steps // 4), args.warmup, barrier, torch)

    if rank == 0 and world == 1 and not args.no_also and not args.no_kcycle:
        out["also_kcycle"] = kcycle_c3()
        out["also_kcycle"]["cpu_reference_same_system"] = kcycle_cpu_reference()
        out["also_kcycle_c5_schur"] = kcycle_c5_schur_and_f32()
        out["also_kcycle_strict_fp64"] = kcycle_c3_strict_fp64()
        out["also_kcycle_16bit_storage"] = kcycle_c3_16bit()
        out["also_kcycle_batched"] = kcycle_c3_batched()
        # the same configuration with the K-cycle in complex<float> (the outer solve, its tolerance and the residual check stay fp64): one system, then 8
        f1, f8 = kcycle_c3_batched(1, f32=True), kcycle_c3_batched(8, f32=True)
        out["also_kcycle"]["fp32_kcycle"] = f1
        if "value" in f1 and "value" in out["also_kcycle"]:
            out["also_kcycle"]["fp32_over_fp64"] = f1["value"] / out["also_kcycle"]["value"]
        out["also_kcycle_batched"]["fp32_kcycle"] = f8
        # ... and with the Galerkin levels' matrices of that complex<float> K-cycle stored as complex<half> (opt-in, QMG_F16_COARSE=1)
        h1 = kcycle_c3_batched(1, f32=True, extra_env={"QMG_F16_COARSE": "1"})
        if "workload" in h1:
            h1["workload"] += "; coarse-level matrices of the K-cycle stored as complex<half> (QMG_F16_COARSE=1)"
        out["also_kcycle"]["fp32_kcycle_16bit_storage"] = h1
        out["also_kcycle_c5_shape"] = kcycle_c5_shape()

    if not args.no_also:   # every rank takes part: the slabs of one lattice
        def everyone_ok(res):   # a leg that failed on ANY rank ends the slab legs for all ranks (the next leg's children would only wait for each other)
            good = 1.0 if (res is not None and "error" not in res) else 0.0
            if dist is not None:
                t = torch.tensor([good], device="cuda", dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                good = t.item()
            return good > 0.5
        barrier()
        slab = slab_solve(L, world, rank)
        go_on = everyone_ok(slab)
        slab_k = slab_c5 = None
        if go_on and not args.no_kcycle:
            barrier()
            slab_k = slab_kcycle(2048, world, rank)
            go_on = everyone_ok(slab_k)
        if go_on and not args.no_kcycle:
            barrier()
            slab_c5 = slab_kcycle_c5(world, rank)
            go_on = everyone_ok(slab_c5)
        if rank == 0:
            out["also_slab_solve"] = slab
            if slab_k is not None:
                out["also_slab_kcycle"] = slab_k
            if slab_c5 is not None:
                out["also_slab_kcycle_c5"] = slab_c5
            if not go_on:
                out["also_slab_note"] = "a slab leg failed on some rank; the remaining slab legs were skipped"

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(fixture)
        out["cpu_baseline_all_cores"] = cpu_baseline_all_cores()

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
