"""U(1) gauge generation and observables on the device (csrc/qmg_u1.hip, include/qmg/u1.hpp).

Pins, strongest first:
  * plaquette / topological charge / non-compact action: GPU vs an independent numpy statement (np.roll on (x, y) grids)
    on the reference's own stored configurations -- deterministic, 1e-13;
  * heatbath: the parallel four-colour heatbath samples the same Gibbs measure as the reference's sequential sweep, so its
    ENSEMBLE must reproduce (i) the exact free-field results of the non-compact action, <cos theta_p> = exp(-1/(2 beta)) and
    <beta/2 theta_p^2> = 1/2, and (ii) the plaquettes of the reference's stored beta = 6.0 / 10.0 configurations;
  * physics the reference stores: the would-be pion mass of tests/n15_wilson_goldstone_u1_heatbath/critical_mass.txt
    (32^2, beta = 6.0, m = +0.01: m_pi = 0.28205(47)) from the n15 counterpart driver -- the one numeric OUTPUT the reference
    holds for this path (a statistical band, not bit parity)."""
import importlib
import os
import re
import subprocess

import numpy as np
import pytest

import coordspace as cs

qmg = importlib.import_module("quantum-mg_amd")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVERS = os.path.join(ROOT, "quantum-mg_amd", "drivers")


@pytest.fixture(scope="module", autouse=True)
def _device():
    qmg.build()
    subprocess.check_call(["make", "-C", DRIVERS, "-j4"], stdout=subprocess.DEVNULL)
    qmg.init(0)
    yield
    qmg.sync()


def np_plaquette(Ux, Uy):
    p = Ux * cs.fwd(Uy, 0) * np.conj(cs.fwd(Ux, 1)) * np.conj(Uy)     # U_x(x) U_y(x+xhat) U_x^*(x+yhat) U_y^*(x), u1_utils.h:424-462
    return p.mean(), np.angle(p).sum() / (2 * np.pi)


def eo_links(phases, L):
    Ux, Uy = cs.phases_to_links(phases, L, L)
    return Ux, Uy, cs.links_to_eo_gauge(Ux, Uy, L, L)


@pytest.mark.parametrize("name,L", [("l32t32b60", 32), ("l64t64b60", 64), ("l128t128b60", 128)])
def test_plaquette_and_topology_of_the_stored_configurations(golden_dir, name, L):
    ph = np.loadtxt(os.path.join(golden_dir, name + "_heatbath.dat"))
    Ux, Uy, g = eo_links(ph, L)
    want_p, want_q = np_plaquette(Ux, Uy)
    got_p, got_q = qmg.u1_plaquette(qmg.DeviceArray.from_host(g), L, L)
    assert abs(got_p - want_p) < 1e-13 and abs(got_q - want_q) < 1e-9
    assert abs(got_q - round(got_q)) < 1e-9                      # an integer on a periodic lattice
    assert 0.90 < got_p.real < 0.94                              # beta = 6.0: exp(-1/12) = 0.9200


def test_noncompact_action_and_polar_round_trip():
    L = 24
    rng = np.random.default_rng(5)
    A = rng.normal(0.0, 0.4, size=(L, L, 2))
    Ax, Ay = A[:, :, 0], A[:, :, 1]
    theta = Ax + cs.fwd(Ay, 0) - cs.fwd(Ax, 1) - Ay
    phase_eo = np.concatenate([cs.grid_to_eo(Ax[:, :, None].astype(complex), L, L, 1).real, cs.grid_to_eo(Ay[:, :, None].astype(complex), L, L, 1).real])
    dph = qmg.DeviceArray.from_host(phase_eo.astype(np.float64))
    assert abs(qmg.u1_noncompact_action(dph, L, L, 6.0) - 3.0 * np.sum(theta ** 2)) < 1e-10
    dg = qmg.DeviceArray(2 * L * L)
    qmg.u1_phase_to_gauge(dg, dph, 2 * L * L)
    assert np.allclose(dg.to_host(), np.exp(1j * phase_eo), atol=1e-15)
    back = qmg.DeviceArray(2 * L * L, np.float64)
    qmg.u1_gauge_to_phase(back, dg, 2 * L * L)
    assert np.allclose(back.to_host(), phase_eo, atol=1e-14)    # |A| < pi here


@pytest.mark.parametrize("beta,stored", [(6.0, ["l32t32b60", "l64t64b60", "l128t128b60"]), (10.0, [])])
def test_heatbath_ensemble_matches_free_field_theory_and_the_stored_configs(golden_dir, beta, stored):
    L, n_therm, n_meas, n_sep = 64, 400, 120, 10
    V = L * L
    ph = qmg.DeviceArray.zeros(2 * V, np.float64)
    g = qmg.DeviceArray(2 * V)
    qmg.u1_heatbath_noncompact(ph, L, L, beta, n_therm, 2024)
    done = n_therm
    plaq, act = [], []
    for _ in range(n_meas):
        qmg.u1_heatbath_noncompact(ph, L, L, beta, n_sep, 2024, first_sweep=done)
        done += n_sep
        qmg.u1_phase_to_gauge(g, ph, 2 * V)
        plaq.append(qmg.u1_plaquette(g, L, L)[0].real)
        act.append(qmg.u1_noncompact_action(ph, L, L, beta) / V)
    plaq, act = np.array(plaq), np.array(act)
    # plaquette angles are independent N(0, 1/beta): <cos> = exp(-1/(2 beta)), <beta/2 theta^2> = 1/2 (minus one zero mode per lattice)
    err_p = plaq.std(ddof=1) / np.sqrt(n_meas) * 2.0             # x2: residual autocorrelation at 10 sweeps separation
    assert abs(plaq.mean() - np.exp(-0.5 / beta)) < 5 * err_p + 2e-4, (plaq.mean(), np.exp(-0.5 / beta), err_p)
    assert abs(act.mean() - 0.5 * (V - 1) / V) < 5 * act.std(ddof=1) / np.sqrt(n_meas) * 2.0 + 2e-4, act.mean()
    # the scatter of single configurations: var(cos theta) / V
    sig1 = np.sqrt((0.5 * (1 + np.exp(-2.0 / beta)) - np.exp(-1.0 / beta)) / V)
    assert 0.6 * sig1 < plaq.std(ddof=1) < 1.6 * sig1
    for name in stored:   # the reference's own configurations are draws of the same ensemble
        Ls = int(name[1:name.index("t")])
        Ux, Uy, _ = eo_links(np.loadtxt(os.path.join(golden_dir, name + "_heatbath.dat")), Ls)
        p_ref = np_plaquette(Ux, Uy)[0].real
        sig_ref = np.sqrt((0.5 * (1 + np.exp(-2.0 / beta)) - np.exp(-1.0 / beta)) / (Ls * Ls))
        assert abs(p_ref - plaq.mean()) < 4.5 * sig_ref, (name, p_ref, plaq.mean(), sig_ref)


def test_heatbath_is_reproducible_and_layout_independent():
    L = 16
    a, b = qmg.DeviceArray.zeros(2 * L * L, np.float64), qmg.DeviceArray.zeros(2 * L * L, np.float64)
    qmg.u1_heatbath_noncompact(a, L, L, 6.0, 7, 11)
    qmg.u1_heatbath_noncompact(b, L, L, 6.0, 3, 11)
    qmg.u1_heatbath_noncompact(b, L, L, 6.0, 4, 11, first_sweep=3)     # continuing the stream == one call
    assert np.array_equal(a.to_host(), b.to_host())
    qmg.u1_heatbath_noncompact(b, L, L, 6.0, 1, 12, first_sweep=7)
    assert not np.array_equal(a.to_host(), b.to_host())


@pytest.mark.parametrize("mass,m_pi_ref", [(0.01, 0.28205), (-0.01, 0.23957)])
def test_n15_pion_mass_matches_the_reference_table(tmp_path, mass, m_pi_ref):
    """tests/n15_wilson_goldstone_u1_heatbath/critical_mass.txt:8-9: 32^2, beta = 6.0, m = +0.01 -> m_pi = 0.28205(47), m = -0.01 -> 0.23957(53).
    The counterpart driver: device heatbath (100 sweeps between measurements, as n15:55), two BiCGStab-6 inversions per
    configuration, correlator through qmg_norm2sq_cv_timeslice.  400 configurations, cosh fit over t = 7..16 of the folded
    correlator; the band also has to absorb the fit-window choice, so +-0.012 -- which still separates the table's neighbouring masses
    (a 0.042 step).  Why not tighter, and why not the rows nearer the critical mass: the Wilson correlator's distribution is heavy-tailed
    there (near-zero modes on single configurations), and 400 configurations leave the fit +-0.007 (m = -0.01) to +-0.1 (m = -0.05) -- all
    five rows measured in profiles/r03_pion_table.txt; the table's 5e-4 needs the reference author's (unrecorded) statistics.  The
    staggered table below has no such tail and is held to +-0.006 on every row."""
    from scipy.optimize import curve_fit
    cfg = str(tmp_path / "last.dat")
    out = subprocess.run([os.path.join(DRIVERS, "n15_wilson_goldstone_u1_heatbath"), "32", str(mass), "6.0", "400", "100", "1000", "1337", cfg], cwd=DRIVERS,
                         env=dict(os.environ, QMG_QUIET="1"), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "400 measurements, 0 unconverged" in out.stdout
    plaq = float(re.search(r"The plaquette is ([-\d.]+)", out.stdout).group(1))
    assert abs(plaq - np.exp(-1.0 / 12.0)) < 2e-3
    body = out.stdout[out.stdout.index("[QMG-BEGIN-PION]"):out.stdout.index("[QMG-END-PION]")]
    rows = re.findall(r"^(\d+) ([-\d.e+]+) \+/- ([-\d.e+]+)$", body, re.M)
    t = np.array([int(r[0]) for r in rows], dtype=float)
    c, dc = np.array([float(r[1]) for r in rows]), np.array([float(r[2]) for r in rows])
    assert len(t) == 32 and np.all(c > 0)
    sel = (t >= 7) & (t <= 16)                                   # the folded correlator: A cosh(m (t - T/2))
    (amp, m_pi), cov = curve_fit(lambda tt, a, m: a * np.cosh(m * (tt - 16.0)), t[sel], c[sel], p0=(c[16], 0.3), sigma=dc[sel], absolute_sigma=True)
    assert abs(m_pi - m_pi_ref) < 0.012, (m_pi, np.sqrt(cov[1, 1]))
    # and the reference's own estimator (n15:211-216), averaged over the plateau
    eff = np.array([np.arccosh((c[j + 1] + c[j - 1]) / (2.0 * c[j])) for j in range(8, 15)])
    assert abs(np.nanmean(eff) - m_pi_ref) < 0.02, eff
    # the written configuration is in the reference's format: 2 L^2 phases, one per line, in (-pi, pi]
    ph = np.loadtxt(cfg)
    assert ph.shape == (2 * 32 * 32,) and np.all(np.abs(ph) <= np.pi)
    Ux, Uy, g = eo_links(ph, 32)
    assert abs(qmg.u1_plaquette(qmg.DeviceArray.from_host(g), 32, 32)[0] - np_plaquette(Ux, Uy)[0]) < 1e-13


@pytest.mark.parametrize("mass,m_pi_ref", [(0.1, 0.355891), (0.08, 0.308843), (0.06, 0.258516), (0.04, 0.202947)])
def test_n20_staggered_pion_mass_matches_the_reference_table(mass, m_pi_ref):
    """tests/n20_staggered_goldstone_u1_heatbath/critical_mass.txt:4-7, EVERY row: 32^2, beta = 6.0, m = 0.1 / 0.08 / 0.06 / 0.04 -> m_pi =
    0.355891(41) / 0.308843(42) / 0.258516(48) / 0.202947(55).  The counterpart driver (device heatbath, one BiCGStab-6 staggered inversion
    per configuration, Goldstone correlator through qmg_norm2sq_cv_timeslice), 400 configurations, cosh fit over t = 7..16 of the folded
    correlator.  Measured (profiles/r03_pion_table.txt): 0.35744(228) / 0.31023(254) / 0.25985(298) / 0.20385(387), i.e. 0.9e-3 - 1.6e-3 from
    the table with a fit-window spread of 1.5e-3; the band is +-0.006 (twice the statistical error of the worst row), an eighth of the
    table's step between neighbouring masses."""
    from scipy.optimize import curve_fit
    out = subprocess.run([os.path.join(DRIVERS, "n20_staggered_goldstone_u1_heatbath"), "32", str(mass), "6.0", "400", "100", "1000", "1337"], cwd=DRIVERS,
                         env=dict(os.environ, QMG_QUIET="1"), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "400 measurements, 0 unconverged" in out.stdout
    body = out.stdout[out.stdout.index("[QMG-BEGIN-PION]"):out.stdout.index("[QMG-END-PION]")]
    rows = re.findall(r"^(\d+) ([-\d.e+]+) \+/- ([-\d.e+]+)$", body, re.M)
    t = np.array([int(r[0]) for r in rows], dtype=float)
    c, dc = np.array([float(r[1]) for r in rows]), np.array([float(r[2]) for r in rows])
    assert len(t) == 32 and np.all(c > 0)
    sel = (t >= 7) & (t <= 16)
    (amp, m_pi), cov = curve_fit(lambda tt, a, m: a * np.cosh(m * (tt - 16.0)), t[sel], c[sel], p0=(c[16], 0.3), sigma=dc[sel], absolute_sigma=True)
    assert abs(m_pi - m_pi_ref) < 0.006, (mass, m_pi, np.sqrt(cov[1, 1]))
