"""Host-side helper methods of the drop-in classes that the reference's own unit tests call
(VERDICT r2 'missing' item 1): the pins of tests/modules/test_psi_thermwind.py:77-86,
test_psi_SO.py:135-142 and the taper / boundary-condition / matrix helpers, restated against
the wrappers.  Pure NumPy: no GPU needed (the device paths are pinned in the *_gpu.py files)."""
import numpy as np
import pytest

from pymoc_amd.modules import Psi_SO, Psi_Thermwind, SO_ML


def test_thermwind_bc_and_ode():
  """test_psi_thermwind.py:77-86: bc picks the stream function at both ends; ode returns
  (Psi', (b2 - b1)/f)."""
  z = np.linspace(-4000., 0., 80)
  psi = Psi_Thermwind(z=z, b1=np.linspace(0.03, -0.01, 80), b2=np.linspace(0.02, 0.0, 80))
  assert np.array_equal(psi.bc([1, 2], [3, 4]), np.array([1, 3]))
  # scalar profiles, scalar depth (the reference's own case: b1 = 0, b2 = 0.005)
  q = Psi_Thermwind(z=z, b1=0.0, b2=0.005, f=1.2e-4)
  ode = q.ode(-1e3, [0, 1])
  assert ode[0] == 1
  assert ode[1][0] == pytest.approx(1.0 / 1.2e-4 * 0.005, rel=1e-15)
  # array profiles, the whole grid at once: rows (y[1], rhs)
  y = np.vstack((np.zeros(80), np.arange(80.)))
  out = psi.ode(z, y)
  assert out.shape == (2, 80) and np.array_equal(out[0], y[1])
  assert np.allclose(out[1], (psi.b2(z) - psi.b1(z)) / psi.f, rtol=1e-15)


def _so(**kw):
  base = dict(z=np.linspace(-4000., 0., 81), y=np.linspace(0., 2.0e6, 51),
              b=np.linspace(0.03, -0.001, 81), bs=np.linspace(0.05, 0.10, 51), tau=0.12)
  base.update(kw)
  return Psi_SO(**base)


def test_psi_so_calc_N2_of_a_linear_profile():
  """test_psi_SO.py:135-142: constant stratification, end levels included."""
  so = _so()
  N2 = so.calc_N2()
  want = (so.b(so.z[1]) - so.b(so.z[0])) / (so.z[1] - so.z[0])
  assert np.allclose(N2(so.z), want, rtol=0, atol=1e-10)


def test_psi_so_tapers():
  """psi_SO.py:164-216: quadratic tapers over the bottom / top H metres; without H the weight
  is 1, except the Ekman form (scalar=False): ones with a zero at the surface level."""
  so = _so()
  z = so.z
  assert so.calc_bottom_taper(None, z) == 1. and so.calc_top_taper(None, z) == 1.
  bt = so.calc_bottom_taper(1000., z)
  assert bt[0] == 0. and np.all(bt[z >= -3000.] == 1.) and np.all(np.diff(bt) >= 0)
  k = 7
  assert bt[k] == pytest.approx(1 - (-3000. - z[k])**2 / 1e6, rel=1e-14)
  tt = so.calc_top_taper(200., z)
  assert tt[-1] == 0. and np.all(tt[z <= -200.] == 1.) and np.all(np.diff(tt) <= 0)
  assert tt[-2] == pytest.approx(1 - (z[-2] + 200.)**2 / 4e4, rel=1e-14)
  ek = so.calc_top_taper(None, z, scalar=False)
  assert ek.shape == z.shape and ek[-1] == 0. and np.all(ek[:-1] == 1.)


def test_psi_so_bc_GM():
  """psi_SO.py:245-275: zero eddy transport at both ends, or cancelling the Ekman transport
  (Sv -> m^3/s) when bvp_with_Ek."""
  so = _so()
  assert np.array_equal(so.bc_GM([1., 2.], [3., 4.]), np.array([1., 3.]))
  so.bvp_with_Ek = True
  so.Psi_Ek = np.linspace(2., 5., 81)
  assert np.array_equal(so.bc_GM([1., 2.], [3., 4.]), np.array([1. + 2e6, 3. + 5e6]))


def _ml(**kw):
  y = np.linspace(0., 2.0e6, 51)
  base = dict(y=y, Ks=100, h=50, L=4e6, surflux=5.9e3, rest_mask=0.0, b_rest=0.0,
              v_pist=2.0 / 86400.0, bs=0.02 * np.cos(y * 2.0 * np.pi / 2.0e6))
  base.update(kw)
  return SO_ML(**base)


def test_so_ml_boundary_conditions():
  """SO_ML.py:77-98."""
  ml = _ml()
  b_basin, Psi_b = np.linspace(-0.001, 0.03, 80), np.linspace(-1., 2., 80)
  ml.Psi_s = np.zeros(51)
  ml.Psi_s[1] = 0.5       # upwelling next to the boundary: densest upwelling basin water
  ml.set_boundary_conditions(b_basin, Psi_b)
  assert ml.bs[0] == b_basin[np.nonzero(Psi_b > 0)[0][0]]
  ml.Psi_s[1] = -0.5      # otherwise no flux: copy of the neighbour
  ml.set_boundary_conditions(b_basin, Psi_b)
  assert ml.bs[0] == ml.bs[1]
  ml.Psi_s[1] = 0.5
  with pytest.raises(IndexError):
    ml.set_boundary_conditions(b_basin, -np.abs(Psi_b))  # nothing upwells (hazard H10's sibling)


def test_so_ml_advective_tendency_is_upwind():
  """SO_ML.py:100-134, point by point."""
  ml = _ml()
  rng = np.random.default_rng(3)
  ml.Psi_s = rng.standard_normal(51)
  ml.Psi_s[[5, 9]] = 0.
  dy = ml.y[1] - ml.y[0]
  got = ml.calc_advective_tendency(dy)
  assert got[0] == 0. and got[-1] == 0. and got[5] == 0. and got[9] == 0.
  for j in range(1, 50):
    p = ml.Psi_s[j]
    if p < 0:
      want = -p * 1e6 * (ml.bs[j + 1] - ml.bs[j]) / ml.h / ml.L / dy
    elif p > 0:
      want = -p * 1e6 * (ml.bs[j] - ml.bs[j - 1]) / ml.h / ml.L / dy
    else:
      want = 0.
    assert got[j] == want


def test_so_ml_crank_nicolson_helpers():
  """SO_ML.py:136-196: the matrices, and the step they define (U x = V bs)."""
  ml = _ml()
  s = 0.3
  U, V = ml.calc_diffusion_matrix(s), ml.calc_diffusion_matrix(-s)
  n = 51
  assert U.shape == (n, n)
  assert np.array_equal(U[0], np.eye(n)[0]) and np.array_equal(U[-1], np.eye(n)[-1])
  assert np.all(np.diag(U)[1:-1] == 1 + s) and np.all(np.diag(V)[1:-1] == 1 - s)
  assert np.all(np.diag(U, 1)[1:] == -s / 2) and np.all(np.diag(U, -1)[:-1] == -s / 2)
  assert np.count_nonzero(U) == 3 * (n - 2) + 2
  dy, dt = ml.y[1] - ml.y[0], 86400. * 30
  got = ml.calc_implicit_diffusion(dy, dt)
  sd = ml.Ks * dt / dy**2
  want = np.linalg.solve(ml.calc_diffusion_matrix(sd), ml.calc_diffusion_matrix(-sd) @ ml.bs)
  assert np.allclose(got, want, rtol=1e-13, atol=0)
  assert got[0] == ml.bs[0] and got[-1] == ml.bs[-1]  # identity rows: the end points stay
