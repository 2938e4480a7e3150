"""Independent coordinate-space statement of the reference operators (numpy, np.roll on (x,y) grids).

Used to PIN the CPU oracle: nothing here shares the even-odd cshift/index algebra of
oracle/qmg_oracle.cpp; only the storage-layout definition of the reference README.md:4-11
(site i = (y + p*Ly)*Lx/2 + x/2, p = (x+y)&1) is used to move data in and out.

Operator definitions are read off the reference's constructors:
  Wilson2D          operators/wilson.h:167-209
  Staggered2D       operators/staggered.h:50-72, 253-259
  GaugedLaplace2D   operators/gaugedlaplace.h:45-68
"""
import numpy as np


def site_index_grid(Lx, Ly):
    """idx[x, y] = even-odd site index (README.md:4-6, lattice/lattice.h:75-81)."""
    x = np.arange(Lx)[:, None]
    y = np.arange(Ly)[None, :]
    p = (x + y) & 1
    return (y + p * Ly) * (Lx // 2) + x // 2


def eo_to_grid(v, Lx, Ly, nc):
    """(eo,y,x,c) flat vector -> psi[x,y,c]."""
    idx = site_index_grid(Lx, Ly)
    return v.reshape(Lx * Ly, nc)[idx]


def grid_to_eo(psi, Lx, Ly, nc):
    idx = site_index_grid(Lx, Ly)
    out = np.zeros((Lx * Ly, nc), dtype=np.complex128)
    out[idx] = psi
    return out.reshape(-1)


def phases_to_links(phases, Lx, Ly):
    """File order: x outer, y, mu inner (u1/u1_utils.h:53-63). Returns U[mu][x,y]."""
    ph = np.asarray(phases, dtype=np.float64).reshape(Lx, Ly, 2)
    return np.exp(1j * ph[:, :, 0]), np.exp(1j * ph[:, :, 1])


def links_to_eo_gauge(Ux, Uy, Lx, Ly):
    """U[mu][x,y] -> reference LatticeGauge (mu,eo,y,x), nc=1."""
    return np.concatenate([grid_to_eo(Ux[:, :, None], Lx, Ly, 1), grid_to_eo(Uy[:, :, None], Lx, Ly, 1)])


def fwd(a, mu):   # a(x + mu)
    return np.roll(a, -1, axis=mu)


def bwd(a, mu):   # a(x - mu)
    return np.roll(a, +1, axis=mu)


def wilson_apply(psi, Ux, Uy, mass, w=1.0):
    """psi[x,y,2] -> (D psi)[x,y,2]."""
    I2 = np.eye(2)
    s1 = np.array([[0, 1], [1, 0]], dtype=np.complex128)
    s2 = np.array([[0, -1j], [1j, 0]], dtype=np.complex128)
    out = (2.0 * w + mass) * psi
    for mu, (U, sig) in enumerate(((Ux, s1), (Uy, s2))):
        Hp = 0.5 * (-w * I2 + sig)    # times U_mu(x)
        Hm = 0.5 * (-w * I2 - sig)    # times conj U_mu(x-mu)
        out = out + U[:, :, None] * np.einsum("rc,xyc->xyr", Hp, fwd(psi, mu))
        out = out + np.conj(bwd(U, mu))[:, :, None] * np.einsum("rc,xyc->xyr", Hm, bwd(psi, mu))
    return out


def staggered_apply(psi, Ux, Uy, mass):
    """psi[x,y,1]; eta_y = (-1)^x (staggered.h:258)."""
    x = np.arange(psi.shape[0])[:, None, None]
    eta = 1.0 - 2.0 * (x % 2)
    out = mass * psi
    out = out - 0.5 * Ux[:, :, None] * fwd(psi, 0) + 0.5 * np.conj(bwd(Ux, 0))[:, :, None] * bwd(psi, 0)
    out = out - 0.5 * eta * Uy[:, :, None] * fwd(psi, 1) + 0.5 * eta * np.conj(bwd(Uy, 1))[:, :, None] * bwd(psi, 1)
    return out


def laplace_apply(psi, Ux, Uy, mass_sq):
    out = (4.0 + mass_sq) * psi
    for mu, U in enumerate((Ux, Uy)):
        out = out - U[:, :, None] * fwd(psi, mu) - np.conj(bwd(U, mu))[:, :, None] * bwd(psi, mu)
    return out


def gaussian_cvec(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex128)


def rel_l2(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
