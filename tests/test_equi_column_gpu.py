"""Equi_Column.solve on the GPU: solve_bvp's Newton iteration and
residual control in `pm_equi_column_newton` against SciPy itself (the oracle restates the
problem and calls scipy.integrate.solve_bvp) and against the reference's outputs (G13)."""
import numpy as np
import pytest

from oracle import equi_column as EO
from conftest import load_golden, relerr
from pymoc_amd import configs

pytestmark = pytest.mark.gpu


def test_equi_column_golden_through_the_drop_in_class(gpu):
  """Every G13 problem through `Equi_Column(...).solve()`: the same mesh history as SciPy
  (node counts, iterations, status) and psi / b / H of the reference."""
  g = load_golden("equi_column")
  for name, kw in configs.equi_column_cases().items():
    m = gpu.Equi_Column(**kw)
    m.solve()
    r = EO.solve(EO.problem(**kw))
    eq = m._eq
    assert m.status == r["status"] == 0, name
    assert eq.x[0].size == r["x"].size and eq.niter[0] == r["niter"], (
        name, eq.x[0].size, r["x"].size, eq.niter[0], r["niter"])
    assert relerr(eq.x[0], r["x"]) <= 1e-15, name
    # measured 1e-17 ... 8e-12 (profiles/r01/probe_equi_column.txt); the bound leaves room for
    # the different elimination order of the linear solves (band LU here, SuperLU in SciPy)
    assert relerr(eq.y[0], r["y"]) <= 1e-9, (name, relerr(eq.y[0], r["y"]))
    assert abs(m.H - float(g[name + "_H"])) <= 1e-9 * abs(m.H), name
    assert relerr(m.z, g[name + "_z"]) <= 1e-9, name
    assert relerr(m.psi, g[name + "_psi"]) <= 1e-9, (name, relerr(m.psi, g[name + "_psi"]))
    assert relerr(m.b, g[name + "_b"]) <= 1e-9, (name, relerr(m.b, g[name + "_b"]))


def test_equi_column_batch_of_different_problems(gpu):
  """One batch, 24 members with different B_int / area / diffusivity, H unknown: every
  member against SciPy on that member alone (meshes refine independently)."""
  rng = np.random.default_rng(11)
  n = 24
  B = rng.uniform(2e3, 1.2e4, n)
  A = rng.uniform(6e13, 2e14, n)
  kap = rng.uniform(2e-5, 6e-5, n)
  eq = gpu.EquiColumnBatch(n, B_int=B, A=A, kappa=kap, nz=60).solve()
  nodes = set()
  for i in range(n):
    r = EO.solve(EO.problem(B_int=B[i], A=A[i], kappa=kap[i], nz=60))
    assert eq.status[i] == r["status"] == 0, i
    assert eq.x[i].size == r["x"].size and eq.niter[i] == r["niter"], i
    assert relerr(eq.y[i], r["y"]) <= 1e-9, (i, relerr(eq.y[i], r["y"]))
    assert abs(eq.H[i] - r["H"]) <= 1e-9 * r["H"], i
    nodes.add(eq.x[i].size)
  assert len(nodes) > 4


def test_equi_column_api_contract(gpu):
  """Constructor / helper behaviour the reference's tests pin
  (tests/modules/test_equi_column.py:113-368)."""
  z = np.asarray(np.linspace(-4000, 0, 80))
  with pytest.raises(Exception) as e:
    gpu.Equi_Column(z=z, A=2.0e14, kappa=3e-5, H=500.0, B_int=None, b_bot=None)
  assert str(e.value) == 'You need to specify either b_bot or B_int for bottom boundary condition'
  c = gpu.Equi_Column(z=z, B_int=3e3, A=2.0e14, kappa=3e-5)
  assert c.f == 1.2e-4 and c.A == 2.0e14 and c.H is None and c.H_guess == 1500.
  assert np.array_equal(c.zi, np.linspace(-1, 0, 100))
  assert c.bs == -0.025 / 1.2e-4**2 and c.B_int == 3e3
  assert c.kappa(-0.5, 1000.) == 3e-5 / (1000.**2 * 1.2e-4) and c.dkappa_dz(-0.5, 1000.) == 0
  assert c.psi_so(-0.5, 1000.) == 0
  assert c.alpha(-0.5, 1000.) == 1000.**2 / (2.0e14 * c.kappa(-0.5, 1000.))
  assert c.sol_init.shape == (4, 100) and (c.sol_init[0] == 1).all()
  assert (c.sol_init[3] == -c.bz(1500.)).all()
  ya, yb = np.array([1., 2., 3., 4.]), np.array([5., 6., 7., 8.])
  assert np.array_equal(c.bc(ya, yb, [1200.]),
                        np.array([1., 5., 2., 4. + c.bz(1200.), 7. - c.bs / 1200.]))
  with pytest.raises(TypeError) as e:
    c.bc(ya, yb)
  assert str(e.value) == 'Must provide a p array if column does not have an H value'
  with pytest.raises(TypeError):
    c.ode(c.zi, c.sol_init)
  y = np.vstack([np.linspace(0, 1, 100)] * 4)
  out = c.ode(c.zi, y, [1200.])
  assert np.array_equal(out[:3], y[1:]) and np.array_equal(
      out[3], c.alpha(c.zi, 1200.) * y[3] * (y[0] - 0 - 2.0e14 * 0 / 1200.**2))
  karr = np.linspace(3e-5, 1e-5, 80)
  c2 = gpu.Equi_Column(z=z, B_int=3e3, A=2.0e14, kappa=karr, psi_so=(z + 2000)**2, H=500.0,
                       b_bot=4e3)
  assert c2.b_bot == -4e3 / 1.2e-4**2 and (c2.sol_init[3] == -100.).all()
  assert c2.kappa(-0.5, 500.) == np.interp(-250., z, karr) / (500.**2 * 1.2e-4)
  assert c2.dkappa_dz(-0.5, 500.) == np.interp(-250., z, np.gradient(karr, z)) / (500. * 1.2e-4)
  assert c2.psi_so(-0.5, 500.) == np.interp(-250., z, (z + 2000)**2) / (1.2e-4 * 500.**3)
  assert np.array_equal(c2.bc(ya, yb), np.array([1., 5., 3. - c2.b_bot / 500., 7. - c2.bs / 500.]))


def test_equi_column_callable_profiles_are_tabulated(gpu):
  """examples/example_Equi_Bint.py with its own callables (kappa, dkappa_dz, psi_so): the
  drop-in class samples them on a 65537-level table; same meshes as the reference, depth and
  profiles to 1e-7 (measured: H 4e-10, psi / b 7e-9)."""
  g = load_golden("equi_column")
  for i in range(4):
    m = gpu.Equi_Column(**configs.equi_bint_callable_case(i))
    m.solve()
    name = "Bint_fn%d" % i
    assert m.status == 0
    Href = float(g[name + "_H"])
    assert abs(m.H - Href) <= 1e-7 * Href, (i, m.H, Href)
    zr, pr, br = g[name + "_z"], g[name + "_psi"], g[name + "_b"]
    assert m.z.size == zr.size and relerr(m.z, zr) <= 1e-7
    assert relerr(m.psi, pr) <= 1e-7 and relerr(m.b, br) <= 1e-7, i


def test_equi_column_failure_modes_match_solve_bvp(gpu):
  """A singular Jacobian ends with status 2 like SciPy; NaN or hopeless members of a batch end
  with a status of their own without disturbing their neighbours."""
  z = np.linspace(-4000, 0, 80)
  kw = dict(z=z, A=2e14, kappa=3e-5, H=500.0, B_int=None, b_bot=4e3)
  m = gpu.Equi_Column(**kw)
  m.solve()
  r = EO.solve(EO.problem(**kw))
  assert m.status == r["status"] == 2 and m._eq.x[0].size == r["x"].size
  eq = gpu.EquiColumnBatch(3, B_int=np.array([3e3, np.nan, 3e3]), A=2e14,
                           kappa=np.array([3e-5, 3e-5, 1e-9]), nz=40).solve()
  ref = EO.solve(EO.problem(B_int=3e3, A=2e14, kappa=3e-5, nz=40))
  assert eq.status[0] == 0 and eq.status[1] != 0 and eq.status[2] != 0
  assert abs(eq.H[0] - ref["H"]) <= 1e-9 * ref["H"] and eq.x[0].size == ref["x"].size
