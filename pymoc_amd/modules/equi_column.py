"""Drop-in `Equi_Column` with the reference's Python API, solved on the GPU.

Same constructor, attributes and method names as `pymoc.modules.Equi_Column`
(src/pymoc/modules/equi_column.py:6-435).  The helper methods (`alpha`, `bz`, `bc`, `ode`,
the non-dimensional profile closures) are host NumPy, as in the reference -- they are what
SciPy would call back into; `solve()` hands the problem to `pymoc_amd.EquiColumnBatch`, which
runs solve_bvp's Newton iteration and residual control on the device.

Deviation: `kappa` / `dkappa_dz` / `psi_so` given as CALLABLES are tabulated on a fine uniform
grid (TABLE_POINTS levels over [-TABLE_DEPTH_FACTOR * depth, 0]) and interpolated linearly on
the device: the reference calls them with the unknown depth H inside every Newton step, which
has no device counterpart.  Against the reference this costs ~1e-8 (measured on the callables
of examples/example_Equi_Bint.py: same meshes, H to 4e-10, psi / b to 7e-9; golden G13);
numbers and arrays on `z` are exact.
"""
import numpy as np

TABLE_POINTS = 65537        # levels of the table a callable profile is sampled on
TABLE_DEPTH_FACTOR = 4.0    # the table reaches this many times H (or H_guess) deep


class Equi_Column(object):
  def __init__(
      self,
      f=1.2e-4,
      b_s=0.025,
      b_bot=None,
      B_int=3e3,
      A=7e13,
      nz=100,
      sol_init=None,
      H_guess=1500.,
      kappa=6e-5,
      dkappa_dz=None,
      psi_so=None,
      z=None,
      H=None,
  ):
    self.f = f
    self.A = A
    self.H = H
    self.H_guess = H_guess
    self.z = z
    self.zi = np.asarray(np.linspace(-1, 0, nz))
    self._nz = nz
    self._kappa_in, self._dkappa_in, self._psi_in = kappa, dkappa_dz, psi_so
    self.kappa = self.init_kappa(kappa)
    self.dkappa_dz = self.init_dkappa_dz(kappa, dkappa_dz)
    self.init_psi_so(psi_so)
    self.init_b_boundaries(b_s, b_bot, B_int)
    self.sol_init = self.calc_sol_init(sol_init, nz, b_bot)

  # ---- non-dimensional profiles, functions of (z*, H)  (equi_column.py:101-185)
  # One sampler serves the three profile kinds the constructor accepts: `_dimensional(v)` gives
  # v at physical depths (callable: called; array: np.interp on self.z; number: itself), and
  # each init_* divides by its own power of H and f.
  def _dimensional(self, v):
    if callable(v):
      return v
    if isinstance(v, np.ndarray):
      return lambda zp: np.interp(zp, self.z, v)
    return lambda zp: v

  def init_kappa(self, kappa):
    dim = self._dimensional(kappa)
    return lambda z, H: dim(z * H) / (H**2 * self.f)

  def init_dkappa_dz(self, kappa, dkappa_dz=None):
    if not callable(kappa) and not isinstance(kappa, np.ndarray):
      return lambda z, H: 0  # a constant diffusivity has no gradient
    if callable(kappa):
      if callable(dkappa_dz):
        slope = dkappa_dz
      else:  # differentiate the callable on whatever depths it is asked for
        slope = lambda zp: np.gradient(kappa(zp), zp)
    else:
      slope = self._dimensional(np.gradient(kappa, self.z))
    return lambda z, H: slope(z * H) / (H * self.f)

  def init_psi_so(self, psi_so=None):
    if not (callable(psi_so) or isinstance(psi_so, np.ndarray)):
      self.psi_so = lambda z, H: 0
      return
    dim = self._dimensional(psi_so)
    self.psi_so = lambda z, H: dim(z * H) / (self.f * H**3)

  def calc_sol_init(self, sol_init, nz=None, b_bot=None):
    if sol_init is not None:
      return sol_init
    n = len(self.z) if nz is None else nz
    # rows: psi (1 everywhere), psi', psi'' (0), psi''' (-100 with a bottom buoyancy, else
    # minus the bottom stratification of a 1500 m deep cell)
    third = -100.0 if b_bot is not None else -self.bz(1500.)
    return np.vstack([np.ones(n), np.zeros(n), np.zeros(n), np.full(n, third)])

  def init_b_boundaries(self, b_s, b_bot=None, B_int=None):
    if b_bot is None and B_int is None:
      raise Exception(
          'You need to specify either b_bot or B_int for bottom boundary condition'
      )
    self.bs = -b_s / self.f**2  # buoyancies are carried as -b / f^2
    if b_bot is None:
      self.B_int = B_int
    else:
      self.b_bot = -b_bot / self.f**2

  def alpha(self, z, H):
    return H**2 / (self.A * self.kappa(z, H))

  def bz(self, H):
    return self.B_int / (self.f**3 * H**2 * self.A * self.kappa(-1, H))

  # ---- what SciPy would call back into (equi_column.py:286-406)
  def _depth(self, p):
    """The cell depth: the fixed H, or the free parameter p[0]; the reference's TypeError
    when neither exists."""
    if self.H is not None:
      return self.H
    if p is None or len(p) == 0:
      raise TypeError('Must provide a p array if column does not have an H value')
    return p[0]

  def bc(self, ya, yb, p=None):
    depth = self._depth(p)
    bottom = (ya[2] - self.b_bot / depth if getattr(self, 'b_bot', None) is not None
              else ya[3] + self.bz(depth))
    free_depth = [ya[1]] if self.H is None else []
    return np.array([ya[0], yb[0]] + free_depth + [bottom, yb[2] - self.bs / depth])

  def ode(self, z, y, p=None):
    H = self._depth(p)
    drive = y[0] - self.psi_so(z, H) - self.A * self.dkappa_dz(z, H) / (H**2)
    return np.vstack((y[1], y[2], y[3], self.alpha(z, H) * y[3] * drive))

  # ---- equi_column.py:408-435
  def solve(self):
    from ..equi_column import EquiColumnBatch
    kappa_in, psi_in, zg, dk_in = self._kappa_in, self._psi_in, self.z, None
    if callable(kappa_in) or callable(psi_in):
      # tabulate: callables -> samples; array / scalar companions are re-sampled the way the
      # reference's closures would evaluate them (np.interp on self.z, or constant)
      depth = TABLE_DEPTH_FACTOR * (self.H if self.H is not None else self.H_guess)
      if self.z is not None:
        depth = max(depth, -float(np.min(self.z)))
      zg = np.linspace(-depth, 0., TABLE_POINTS)
      sample = lambda v: (v(zg) if callable(v) else
                          (np.interp(zg, self.z, v) if isinstance(v, np.ndarray) else None))
      if callable(kappa_in):
        dk_in = (self._dkappa_in(zg) if callable(self._dkappa_in)
                 else np.gradient(kappa_in(zg), zg))
        kappa_in = kappa_in(zg)
      elif isinstance(kappa_in, np.ndarray):
        dk_in = np.interp(zg, self.z, np.gradient(kappa_in, self.z))
        kappa_in = sample(kappa_in)
      if callable(psi_in) or isinstance(psi_in, np.ndarray):
        psi_in = np.asarray(sample(psi_in), dtype=np.float64)
    bbot_set = getattr(self, 'b_bot', None) is not None
    f2 = self.f**2
    eq = EquiColumnBatch(
        1, f=self.f, b_s=-self.bs * f2, b_bot=-self.b_bot * f2 if bbot_set else None,
        B_int=None if bbot_set else self.B_int, A=self.A, nz=np.shape(self.sol_init)[1],
        sol_init=np.asarray(self.sol_init, dtype=np.float64)[None], H_guess=self.H_guess,
        kappa=kappa_in, psi_so=psi_in, z=zg, H=self.H, dkappa_dz=dk_in)
    eq.zi = self.zi
    eq.bs[:] = self.bs  # exactly the reference's non-dimensional values
    if bbot_set:
      eq.bb[:] = self.b_bot
    eq.solve()
    self._eq = eq
    x, y = eq.x[0], eq.y[0]
    if self.H is None:
      self.H = eq.H[0]
    self.status = int(eq.status[0])
    if self.z is None:
      self.z = x * self.H
      self.psi = y[0, :] * self.f * self.H**3 / 1e6
      self.b = -y[2, :] * self.f**2 * self.H
    else:
      sol = eq.sol(0, self.z / self.H)
      self.psi = sol[0, :] * self.f * self.H**3 / 1e6
      self.psi[self.z < -self.H] = np.nan
      self.b = -sol[2, :] * self.f**2 * self.H
      self.b[self.z < -self.H] = np.nan
