"""GPU Krylov drivers (include/qmg/krylov.hpp: CG, BiCGStab-L, Richardson, MR, GCR -- the device restatement of the absent
quantum-linalg inverters) against the CPU oracle's twins on the SAME dumped right-hand sides
(drivers/krylov_parity.cpp).  The twins are pinned to scipy in tests/test_oracle_krylov.py; against the reference itself
these drivers stay PARITY UNPINNED (quantum-linalg stores no outputs).

Bar: iteration counts equal +-1 (bit-different reductions can move a stopping decision by one step; BiCGStab-L: one sweep of
L), solutions equal to 1e-7 relative (both solve to <= 1e-9), fixed-iteration runs (MR, Richardson) equal to 1e-12.
Also: CG driven from HOST vectors through apply_stencil_2D_host_thunk (the reference's matrix_op_cplx signature)."""
import os
import re
import subprocess

import numpy as np
import pytest

import coordspace as cs
import oracle_lib as ol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVERS = os.path.join(ROOT, "quantum-mg_amd", "drivers")
L, MASS = 32, 0.05


@pytest.fixture(scope="module")
def run(golden_dir, tmp_path_factory):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "quantum-mg_amd"), "-j4", "libqmg_hip.so"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", DRIVERS, "-j4"], stdout=subprocess.DEVNULL)
    tmp = str(tmp_path_factory.mktemp("krylov"))
    gauge_file = os.path.join(golden_dir, "l32t32b60_heatbath.dat")
    out = subprocess.run([os.path.join(DRIVERS, "krylov_parity"), str(L), str(MASS), gauge_file, tmp], cwd=DRIVERS, capture_output=True, text=True, timeout=200)
    assert out.returncode == 0 and "[QMG-ERROR]" not in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    rows = {m[0]: (int(m[1]), int(m[2]), int(m[3]), float(m[4])) for m in re.findall(r"\[KRYLOV\] (\w+) success (\d) iter (\d+) ops (\d+) rel_res ([-\d.e+]+)", out.stdout)}
    load = lambda name: np.fromfile(os.path.join(tmp, name + ".bin"), dtype=np.complex128)
    gauge = ol.phases_to_gauge_u1(np.loadtxt(gauge_file), L, L)
    wc, wh = ol.wilson_fill(gauge, L, L)
    dc, dh = ol.build_dagger(wc, wh, L, L, 2)
    lc, lh = ol.laplace_fill(gauge, L, L)
    ops = {"wilson": ol.make_desc(L, L, 2, wc, wh, MASS), "dagger": ol.make_desc(L, L, 2, dc, dh, MASS), "laplace": ol.make_desc(L, L, 1, lc, lh, 0.01)}
    return rows, load, ops


CASES = [  # name, oracle kind, operator, rhs file, max_iter, tol, param_i, param_d, iteration slack
    ("bicgstab1", ol.KRYLOV_BICGSTAB_L, "wilson", "b_wilson", 500, 1e-9, 1, 0.0, 2),
    ("bicgstab6", ol.KRYLOV_BICGSTAB_L, "wilson", "b_wilson", 500, 5e-5, 6, 0.0, 6),
    ("gcr8", ol.KRYLOV_GCR, "wilson", "b_wilson", 400, 1e-9, 8, 0.0, 1),
    ("gcr", ol.KRYLOV_GCR, "wilson", "b_wilson", 60, 1e-9, -1, 0.0, 1),
    ("cg_laplace", ol.KRYLOV_CG, "laplace", "b_laplace", 2000, 1e-10, 0, 0.0, 1),
]


@pytest.mark.parametrize("name,kind,op,bfile,max_iter,tol,pi,pd,slack", CASES)
def test_converging_drivers_match_the_oracle_twins(run, name, kind, op, bfile, max_iter, tol, pi, pd, slack):
    rows, load, ops = run
    b = load(bfile)
    conv, it, x, rsq, _ = ol.krylov_solve(kind, ops[op], b, max_iter, tol, param_i=pi, param_d=pd)
    g_ok, g_it, g_ops, g_rel = rows[name]
    assert bool(g_ok) == conv
    assert abs(g_it - it) <= slack, (name, g_it, it)
    xg = load("x_" + name)
    true = np.linalg.norm(b - ol.stencil_apply(ops[op], xg)) / np.linalg.norm(b)    # the GPU solution against the ORACLE's operator
    assert true <= 1.5 * max(tol, 1e-12) or not conv
    assert cs.rel_l2(xg, x) < (1e-3 if tol > 1e-6 else 1e-7), (name, cs.rel_l2(xg, x))


def test_fixed_iteration_drivers_match_exactly(run):
    rows, load, ops = run
    b = load("b_wilson")
    _, it, x, rsq, _ = ol.krylov_solve(ol.KRYLOV_MR, ops["wilson"], b, 6, 1e-30, param_d=0.85)
    assert rows["mr"][1] == it == 6 and cs.rel_l2(load("x_mr"), x) < 1e-12
    assert abs(rows["mr"][3] - np.sqrt(rsq) / np.linalg.norm(b)) < 1e-10
    _, it, x, rsq, _ = ol.krylov_solve(ol.KRYLOV_RICHARDSON, ops["wilson"], b, 10, 1e-10, param_i=250, param_d=0.33)
    assert rows["richardson"][1] == it == 10 and rows["richardson"][0] == 0 and cs.rel_l2(load("x_richardson"), x) < 1e-12
    assert abs(rows["richardson"][3] - np.sqrt(rsq) / np.linalg.norm(b)) < 1e-10


def test_cg_on_the_normal_operator_matches(run):
    rows, load, ops = run
    bn = load("b_normal")
    assert cs.rel_l2(bn, ol.stencil_apply(ops["dagger"], load("b_wilson"))) < 1e-13     # M^dag b on the device
    conv, it, x, _, _ = ol.krylov_solve(ol.KRYLOV_CG, ops["wilson"], bn, 3000, 1e-10, dagger_desc=ops["dagger"], normal=True)
    assert conv and rows["cg_normal"][0] == 1 and abs(rows["cg_normal"][1] - it) <= 2
    assert cs.rel_l2(load("x_cg_normal"), x) < 1e-6


def test_host_vectors_through_the_matrix_op_cplx_thunk(run):
    """An unmodified CPU solver (drivers/krylov_parity.cpp host_cg, which sees only the reference's callback type
    matrix_op_cplx = void(*)(complex<double>*, complex<double>*, void*), stencil_2d.h:15-19) drives the GPU operator
    through apply_stencil_2D_host_thunk: host rhs staged to HBM, applied, copied back.  Same iterates as the device CG."""
    rows, load, ops = run
    assert rows["host_thunk_cg"][0] == 1
    assert abs(rows["host_thunk_cg"][1] - rows["cg_laplace"][1]) <= 1
    x = load("x_host_thunk_cg")
    b = load("b_laplace")
    assert cs.rel_l2(ol.stencil_apply(ops["laplace"], x), b) < 1.5e-10
    assert cs.rel_l2(x, load("x_cg_laplace")) < 1e-8
