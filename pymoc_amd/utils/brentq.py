"""scipy.optimize.brentq(f, a, b) with SciPy's defaults (xtol = 2e-12, rtol = 4 eps,
maxiter = 100), restated from scipy/optimize/Zeros/brentq.c (SciPy 1.15.3) decision for
decision, for the one place where the root has to be found ON THE HOST: `Psi_SO.ys` with a
CALLABLE surface buoyancy, which only Python can evaluate (psi_SO.py:106-140).  Array profiles
never come here: the kernel inverts them (pymoc_amd/csrc/psi_so.hip.h, `brentq_interp` is the
device twin of this function).  tests/test_host_cpu.py checks it against SciPy's own brentq."""
import math


def brentq(f, xa, xb, xtol=2e-12, rtol=8.881784197001252e-16, maxiter=100):
  xpre, xcur = float(xa), float(xb)
  xblk = fblk = spre = scur = 0.0
  fpre, fcur = float(f(xpre)), float(f(xcur))
  if fpre == 0:
    return xpre
  if fcur == 0:
    return xcur
  if math.copysign(1.0, fpre) == math.copysign(1.0, fcur):
    raise ValueError("f(a) and f(b) must have different signs")
  for _ in range(maxiter):
    if fpre != 0 and fcur != 0 and math.copysign(1.0, fpre) != math.copysign(1.0, fcur):
      xblk, fblk = xpre, fpre
      spre = scur = xcur - xpre
    if abs(fblk) < abs(fcur):
      xpre, xcur, xblk = xcur, xblk, xcur
      fpre, fcur, fblk = fcur, fblk, fcur
    delta = (xtol + rtol * abs(xcur)) / 2
    sbis = (xblk - xcur) / 2
    if fcur == 0 or abs(sbis) < delta:
      return xcur
    if abs(spre) > delta and abs(fcur) < abs(fpre):
      if xpre == xblk:  # secant
        stry = -fcur * (xcur - xpre) / (fcur - fpre)
      else:  # inverse quadratic
        dpre = (fpre - fcur) / (xpre - xcur)
        dblk = (fblk - fcur) / (xblk - xcur)
        stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre))
      if 2 * abs(stry) < min(abs(spre), 3 * abs(sbis) - delta):
        spre, scur = scur, stry
      else:
        spre = scur = sbis
    else:
      spre = scur = sbis
    xpre, fpre = xcur, fcur
    if abs(scur) > delta:
      xcur += scur
    else:
      xcur += delta if sbis > 0 else -delta
    fcur = float(f(xcur))
  raise RuntimeError("Failed to converge after %d iterations." % maxiter)
