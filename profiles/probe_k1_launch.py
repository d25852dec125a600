import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np
import pymoc_amd as gpu
from pymoc_amd import configs
for N in (1024, 8192):
  c = configs.config2(N=N)
  batch = gpu.ColumnBatch(c["z"], c["kappa"], c["Area"], c["b0"], bs=c["bs"], bbot=c["bbot"], N2min=c["N2min"], do_conv=c["do_conv"])
  from pymoc_amd.device import DeviceArray
  wA = DeviceArray.from_host(np.ascontiguousarray(np.broadcast_to(c["wA"], (N, c["z"].size))))
  batch.steps(wA, c["dt"], 100); gpu.synchronize()
  for n in (1, 12, 24, 48, 96, 1000):
    K = 50
    batch.steps(wA, c["dt"], n); gpu.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): batch.steps(wA, c["dt"], n)
    gpu.synchronize()
    el = (time.perf_counter() - t0) / K
    print("N=%5d steps/launch %4d: %.1f us per launch, %.3f us per step, %.2e col-steps/s" % (N, n, el*1e6, el*1e6/n, N*n/el))
