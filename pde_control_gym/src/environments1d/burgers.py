"""BurgersPDE1D -- EXTENSION, not part of lukebhan/PDEControlGym.

BASELINE's third configuration is worded "Hyperbolic1D Burgers (nonlinear) nx=512", but the reference ships no Burgers
environment; its nonlinear hyperbolic environment is TrafficPDE1D.  This class offers the
obvious nonlinear sibling of TransportPDE1D on the same kernel skeleton:

    u_t = u u_x + beta(x) u(0, t),   first-order upwind   n[j] = p[j] + dt*(p[j]*((p[j+1]-p[j])/dx) + (p[0]*beta)[j])

with the same boundary control, sensing options, normalisation, truncation and rewards as TransportPDE1D
(interface and semantics: environments1d/hyperbolic.py:25-227).  **Parity unpinned**: there is no reference
implementation; the HIP kernel is checked bit for bit against this repository's own NumPy restatement only
(oracle/pde_oracle.py: BurgersOracle).  Stability needs ``dt * max|u| / dx <= 1``.
"""
from pde_control_gym.src.environments1d.hyperbolic import TransportPDE1D


class BurgersPDE1D(TransportPDE1D):
    _flux = "burgers"
