"""Equi_Column.solve: parity against SciPy on the same problem and wall time per solve."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pymoc_amd as gpu
from pymoc_amd import configs
from oracle import equi_column as EO

def rel(a, b):
  return np.abs(a - b).max() / np.abs(b).max()

for name, kw in configs.equi_column_cases().items():
  m = gpu.Equi_Column(**kw)
  t0 = time.perf_counter(); m.solve(); t1 = time.perf_counter()
  q = EO.problem(**kw)
  t2 = time.perf_counter(); r = EO.solve(q); t3 = time.perf_counter()
  eq = m._eq
  print("%-16s nodes %4d mesh-iters %d newton %2d  y %.1e  H %.1e  gpu %.1f ms  scipy %.1f ms" % (
      name, eq.x[0].size, eq.niter[0], eq.newton_iters[0], rel(eq.y[0], r["y"]),
      abs(m.H - r["H"]) / r["H"], 1e3 * (t1 - t0), 1e3 * (t3 - t2)))
rng = np.random.default_rng(11)
for n in (64, 1024):
  B = rng.uniform(2e3, 1.2e4, n); A = rng.uniform(6e13, 2e14, n); kap = rng.uniform(2e-5, 6e-5, n)
  t0 = time.perf_counter()
  eq = gpu.EquiColumnBatch(n, B_int=B, A=A, kappa=kap, nz=60).solve()
  t1 = time.perf_counter()
  t2 = time.perf_counter()
  for i in range(16):
    EO.solve(EO.problem(B_int=B[i], A=A[i], kappa=kap[i], nz=60))
  t3 = time.perf_counter()
  print("batch n=%d: %.1f ms total = %.2f ms/member (status ok %d, max nodes %d); scipy %.1f ms/member" % (
      n, 1e3 * (t1 - t0), 1e3 * (t1 - t0) / n, (eq.status == 0).sum(), max(x.size for x in eq.x),
      1e3 * (t3 - t2) / 16))
