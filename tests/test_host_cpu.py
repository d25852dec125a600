"""CPU-only checks: the C-ABI library loads and exports every symbol the header
declares, host-side coercion and error text match the reference's contract, config
generators are shard-invariant.  No compute call is made (there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _header_symbols():
  text = open(os.path.join(ROOT, "include", "pymoc_hip.h")).read()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return sorted(set(re.findall(r"\b(pm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
  from pymoc_amd import _lib
  names = _header_symbols()
  assert len(names) >= 20
  for n in names:
    assert hasattr(_lib.lib, n), "libpymoc_hip.so does not export %s" % n
    assert n in _lib.SIGNATURES, "no ctypes signature for %s" % n
  assert set(_lib.SIGNATURES) == set(names)
  assert b"gfx950" in _lib.lib.pm_version()


def test_struct_layout_matches_header(tmp_path):
  """sizeof of every struct of include/pymoc_hip.h as gcc sees it against its ctypes mirror."""
  import subprocess
  from conftest import ROOT
  from pymoc_amd import _lib
  names = ["pm_columns", "pm_thermwind", "pm_psi_so", "pm_so_ml", "pm_jn2018_bc", "pm_jn2018",
           "pm_run_schedule", "pm_twocol_loop", "pm_jn2018_loop", "pm_column_equi",
           "pm_equi_column"]
  src = tmp_path / "sizes.c"
  src.write_text('#include <stdio.h>\n#include "pymoc_hip.h"\nint main(void) {\n' +
                 "".join('  printf("%s %%zu\\n", sizeof(%s));\n' % (n, n) for n in names) +
                 "  return 0;\n}\n")
  exe = tmp_path / "sizes"
  subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
  out = dict(ln.split() for ln in subprocess.check_output([str(exe)], text=True).splitlines())
  for n in names:
    assert ctypes.sizeof(getattr(_lib, n)) == int(out[n]), n
  assert ctypes.sizeof(_lib.pm_columns) == 16 + 14 * 8  # 4 int32 + 14 pointers


def test_no_device_fails_loudly():
  from pymoc_amd import _lib
  n = ctypes.c_int(-1)
  rc = _lib.lib.pm_device_count(ctypes.byref(n))
  if rc == 0 and n.value > 0:
    pytest.skip("a GPU is visible here")
  with pytest.raises(_lib.PmError):
    _lib.require_device()
  import pymoc_amd
  col = pymoc_amd.Column(z=np.linspace(-4000., 0., 10), kappa=1e-5, Area=1e14, b=0.01)
  with pytest.raises(_lib.PmError):
    col.timestep(wA=0., dt=1.)


def test_cabi_rejects_bad_arguments_before_touching_the_device():
  """Every compute entry validates shapes / pointers first and reports through
  pm_last_error; none of these calls reaches a launch, so they run without a GPU."""
  from pymoc_amd import _lib
  L, C = _lib.lib, ctypes

  def err():
    return L.pm_last_error().decode()

  c = _lib.pm_columns()
  c.ncols, c.nz, c.nsel = 4, 1, 1
  assert L.pm_column_steps(C.byref(c), None, None, None, 1., 1, 7, 0, None) == _lib.PM_EINVAL
  assert "nz" in err()
  c.nz = 10
  assert L.pm_column_steps(C.byref(c), None, None, None, 1., 1, 7, 0, None) == _lib.PM_EINVAL
  assert "NULL" in err()
  assert L.pm_column_steps(None, None, None, None, 1., 1, 7, 0, None) == _lib.PM_EINVAL
  c.ncols = 0  # an empty batch is fine, pointers may be NULL
  assert L.pm_column_steps(C.byref(c), None, None, None, 1., 1, 7, 0, None) == _lib.PM_OK
  t = _lib.pm_thermwind()
  t.n, t.nz, t.nb = 2, 1, 10
  assert L.pm_thermwind_update(C.byref(t), 1, None) == _lib.PM_EINVAL
  so = _lib.pm_psi_so()
  so.n, so.nz, so.ny = 2, 1, 10
  assert L.pm_psi_so_update(C.byref(so), 1, None) == _lib.PM_EINVAL
  ml = _lib.pm_so_ml()
  ml.n, ml.nz, ml.ny = 2, 10, 2
  assert L.pm_so_ml_step(C.byref(ml), 1., None) == _lib.PM_EINVAL
  eq = _lib.pm_column_equi()
  eq.n, eq.nz, eq.mmax, eq.tol = 2, 10, 2000, 1e-3
  assert L.pm_column_equi_pass(C.byref(eq), None) == _lib.PM_EINVAL
  assert "mmax" in err()
  ec = _lib.pm_equi_column()
  ec.n, ec.mmax, ec.tol = 2, 64, 1e-3
  assert L.pm_equi_column_newton(C.byref(ec), None) == _lib.PM_EINVAL
  assert "NULL" in err()
  assert L.pm_axpby(4, 1., None, 1., None, None, None) == _lib.PM_EINVAL
  assert L.pm_equi_column_scratch_doubles(64) >= 64 * 100


def test_product_does_not_import_oracle():
  pkg = os.path.join(ROOT, "pymoc_amd")
  for dirpath, _, files in os.walk(pkg):
    for f in files:
      if f.endswith((".py", ".h", ".hip")):
        src = open(os.path.join(dirpath, f)).read()
        assert "import oracle" not in src and "from oracle" not in src, f
        assert "pymoc_oracle" not in src, f
        assert "import torch" not in src, f


# ---- utils contract (reference tests/utils/test_make_func.py, test_make_array.py)
def test_make_func_contract():
  from pymoc_amd.utils import make_func
  z = np.linspace(-10., 0., 5)
  f = lambda x: 2 * x  # noqa: E731
  assert make_func(f, z, 'f') is f
  arr = np.arange(5.)
  g = make_func(arr, z, 'g')
  assert np.array_equal(g(z), arr)
  arr[2] = 99.  # closure aliases the caller's array (make_func.py:32-37)
  assert g(z[2]) == 99.
  h = make_func(3.0, z, 'h')
  assert np.array_equal(h(z), 3.0 + 0 * z)
  with pytest.raises(TypeError) as e:
    make_func(1, z, 'myst')
  assert str(e.value) == "('myst', 'needs to be either function, numpy array, or float')"


def test_make_array_contract():
  from pymoc_amd.utils import make_array
  z = np.linspace(-10., 0., 5)
  arr = np.arange(5.)
  assert make_array(arr, z, 'a') is arr
  assert np.array_equal(make_array(lambda x: x**2, z, 'a'), z**2)
  assert np.array_equal(make_array(1.5, z, 'a'), 1.5 + 0 * z)
  with pytest.raises(TypeError) as e:
    make_array(1, z, 'myst')
  assert str(e.value) == "('myst', 'needs to be either function, numpy array, or float')"


def test_column_constructor_contract():
  from pymoc_amd import Column
  z = np.linspace(-4000., 0., 20)
  with pytest.raises(TypeError) as e:
    Column(z=1, kappa=1e-5, Area=1e14)
  assert str(e.value) == 'z needs to be numpy array providing grid levels'
  with pytest.raises(TypeError) as e:
    Column(z=np.array([]), kappa=1e-5, Area=1e14)
  assert str(e.value) == 'z needs to be numpy array providing grid levels'
  with pytest.raises(TypeError) as e:
    Column(z=z, kappa=1, Area=1e14)
  assert str(e.value) == "('kappa', 'needs to be either function, numpy array, or float')"
  with pytest.raises(TypeError) as e:
    Column(z=z, kappa=1e-5, Area=1e14, b=1)
  assert str(e.value) == "('b', 'needs to be either function, numpy array, or float')"
  b = 0.02 * np.exp(z / 300.)
  c = Column(z=z, kappa=lambda x: 1e-5 + 0 * x, Area=8e13, b=b, bs=0.02, bbot=0.001,
             bzbot=None, N2min=2e-7)
  assert c.z is z and c.b is b  # arrays are aliased, not copied (column.py:54,67)
  assert (c.bs, c.bbot, c.bzbot, c.N2min) == (0.02, 0.001, None, 2e-7)
  assert np.array_equal(c.bz, np.gradient(b, z))
  assert np.array_equal(c.Akappa(z), 8e13 * (1e-5 + 0 * z))
  assert np.array_equal(c.dAkappa_dz(z), np.gradient(8e13 * (1e-5 + 0 * z), z))


def test_configs_are_shard_invariant():
  from pymoc_amd import configs
  full = configs.config2(N=64)
  part = configs.config2(N=64, members=(16, 48))
  for k in ("kappa", "Area", "wA", "b0", "bs", "do_conv"):
    assert np.array_equal(full[k][16:48], part[k])
  full = configs.config3(N=32)
  part = configs.config3(N=32, members=(8, 9))
  assert np.array_equal(full["kappa"][8:9], part["kappa"])
  full = configs.config4(N=32)
  part = configs.config4(N=32, members=(30, 32))
  assert np.array_equal(full["tau"][30:32], part["tau"])
  full = configs.config5(N=8)
  part = configs.config5(N=8, members=(2, 5))
  assert np.array_equal(full["b_basin0"][2:5], part["b_basin0"])


def test_explicit_scheme_limits_hold_for_the_bench_configs():
  from pymoc_amd import configs
  for c in (configs.config2(N=256), configs.config3(N=256), configs.config4(N=256)):
    dz = np.diff(c["z"]).min()
    assert (c["kappa"].max() * c["dt"] / dz**2) < 0.5


def test_brentq_restatement_equals_scipy():
  """pymoc_amd.utils.brentq (host root-finding for CALLABLE bs in Psi_SO.ys) is SciPy's brentq
  decision for decision: bit-identical roots on random smooth functions."""
  from scipy import optimize
  from pymoc_amd.utils.brentq import brentq
  rng = np.random.default_rng(5)
  seen = 0
  for k in range(600):
    a, b, c = rng.uniform(0.5, 3), rng.uniform(-2, 2), rng.uniform(-1, 1)
    f = lambda y: np.tanh(a * y + b) + 0.3 * np.sin(3 * y) + 0.2 * c - 0.1 * k / 600
    if f(-4.) * f(4.) > 0:
      continue
    seen += 1
    assert optimize.brentq(f, -4., 4.) == brentq(f, -4., 4.)
  assert seen > 200
  with pytest.raises(ValueError):
    brentq(lambda y: 1. + y * y, -1., 1.)


def _div3_reference(d):
  """Pure-Python restatement of pm_div3_proven for one denominator (exact rationals): the
  candidate numerators (quotient within 6 / (2 D) ulp of a rounding midpoint) and whether the
  3-instruction quotient is correctly rounded on every one of them."""
  import math
  from fractions import Fraction as F

  def fma(a, b, c):
    return float(F(a) * F(b) + F(c))  # (int / int true division rounds correctly)

  m, _ = math.frexp(abs(d))
  D = int(m * 2**53)
  dm = float(D)
  y = 1.0 / dm
  v = (D & -D).bit_length() - 1
  Dp = D >> v
  cands = set()
  if v < 3 and Dp > 1:
    for t in (0, 1):
      sh = 53 + t
      for N in range(-6, 7):
        if N == 0 or N % (1 << v):
          continue
        A0 = ((N >> v) * pow((1 << (sh - v)) % Dp, -1, Dp)) % Dp
        lo, hi = (D, 1 << 53) if t == 0 else (1 << 52, D)
        A = A0 + ((lo - A0 + Dp - 1) // Dp) * Dp
        while A < hi:
          q, r = divmod(A * (1 << sh) - N, D)
          if r == 0 and q % 2 == 1:
            cands.add(A)
          A += Dp
  ok = True
  for A in cands:
    for a in (float(A), -float(A)):
      q0 = float(F(a) * F(y))
      r = fma(-dm, q0, a)
      if fma(r, y, q0) != float(F(a) / F(dm)):
        ok = False
  return ok, len(cands)


def test_div3_proof_matches_its_restatement():
  """pm_div3_proven (host function behind PM_COLS_DIV3_PROVEN): the same candidate numerators and
  the same verdict as an exact-rational restatement, for uniform mantissas, mantissas next to 1 and
  2, mantissas with trailing zeros and the grid spacings of BASELINE's grids; zero, subnormal and
  non-finite denominators are not proven."""
  import ctypes as C
  from pymoc_amd._lib import lib, check
  from pymoc_amd.columns import div3_proven
  from pymoc_amd import configs
  rng = np.random.default_rng(5)
  ds = list(rng.uniform(1, 2, 150) * 2.0**rng.integers(-30, 30, 150))
  ds += [1.0 + k * 2.0**-52 for k in range(1, 40)] + [2.0 - k * 2.0**-52 for k in range(1, 40)]
  ds += [float(np.float64(x).view(np.uint64) & ~np.uint64(3)) for x in []]
  ds += [float((np.float64(x).view(np.uint64) & ~np.uint64(1)).view(np.float64)) for x in rng.uniform(1, 2, 40)]
  ds += [float((np.float64(x).view(np.uint64) & ~np.uint64(3)).view(np.float64)) for x in rng.uniform(1, 2, 40)]
  ds += [3.0, 6e13, 1e-7, 40.0, 0.1, 86400.0 * 30]
  for mk in (configs.config2, configs.config5):
    z = mk(N=2)["z"]
    dz = np.diff(z)
    ds += list(np.unique(dz)) + list(np.unique(0.5 * (dz[1:] + dz[:-1])))
  for d in ds:
    d = float(d)
    ok, nc = C.c_int32(-1), C.c_int64(-1)
    arr = np.array([d])
    check(lib.pm_div3_proven(arr.ctypes.data, 1, C.byref(ok), C.byref(nc)))
    rok, rnc = _div3_reference(d)
    assert (bool(ok.value), nc.value) == (rok, rnc), (d, ok.value, nc.value, rok, rnc)
    assert ok.value == 1  # (no denominator is known to fail; the kernels still ask for the proof)
  for bad in (0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, 2.0**-1060):
    assert not div3_proven([1.5, bad])
  assert div3_proven([]) and div3_proven([1.5, -3.0, 2.0**-1000, 2.0**1000])
