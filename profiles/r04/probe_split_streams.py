"""A coupled ensemble split into K sub-ensembles on K HIP streams, every stream's whole run
queued at once (members are independent; no host pacing): does one GPU get more done?
usage: CONFIG=3|4|5 python profiles/r04/probe_split_streams.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pymoc_amd as gpu
from pymoc_amd import configs
from pymoc_amd.device import Stream

CONFIG = int(os.environ.get("CONFIG", 4))
N = {3: 4096, 4: 8192, 5: 4096}[CONFIG]
STEPS = {3: 2400, 4: 2400, 5: 3600}[CONFIG]

def make(k, K, stream, stag):
  sl = (k * N // K, (k + 1) * N // K)
  if CONFIG == 5:
    cfg = configs.config5(N=N, members=sl)
    cfg["rest_mask"] = np.repeat(cfg["rest_mask"][None], sl[1] - sl[0], axis=0)
    return gpu.JN2018Ensemble(cfg, stream=stream)
  cfg = configs.config3(N=N, members=sl) if CONFIG == 3 else configs.config4(N=N, members=sl)
  return gpu.TwoColEnsemble(cfg, stream=stream)

for K in (1, 2, 3, 4, 2, 1):
  streams = [Stream() for _ in range(K)]
  ens = [make(k, K, streams[k], 0) for k in range(K)]
  M = ens[0].M
  for k, e in enumerate(ens):
    e.run(10 * M)
  for s in streams:
    s.sync()
  t0 = time.perf_counter()
  for e in ens:
    e.run(STEPS)
  t_issue = time.perf_counter() - t0
  for s in streams:
    s.sync()
  dt = time.perf_counter() - t0
  print("config %d, %d stream(s): %.4g coupled steps/s (%.1f us per interval; host issued in %.1f ms of %.1f)"
        % (CONFIG, K, N * STEPS / dt, dt / (STEPS / M) * 1e6, t_issue * 1e3, dt * 1e3), flush=True)
  del ens, streams
