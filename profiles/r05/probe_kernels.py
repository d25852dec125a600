"""Run-average duration of every kernel of a coupled config (LaunchTimer: events around every
launch of a full-length run) and the wall-clock rate of an untimed full-length run.
usage: python profiles/r05/probe_kernels.py 3 4 5 6 [--arith contracted] [--nz81]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pymoc_amd
from pymoc_amd import configs
from pymoc_amd.device import Stream, LaunchTimer
st = Stream()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
arith = "contracted" if "--contracted" in sys.argv else "exact"
def make(c):
  if c == 3:
    return pymoc_amd.TwoColEnsemble(configs.config3(), stream=st, arith=arith), 2400
  if c == 4:
    return pymoc_amd.TwoColEnsemble(configs.config4(), stream=st, overlap_updates="--overlap" in sys.argv, arith=arith), 2400
  if c == 6:
    return pymoc_amd.TwoBasinEnsemble(configs.config_twobasin(), stream=st, overlap_updates="--overlap" in sys.argv, arith=arith), 2400
  if "--nz81" in sys.argv:
    cfg = configs.config5(nz=81, dt_days=30.)
  else:
    cfg = configs.config5()
  cfg["rest_mask"] = np.repeat(cfg["rest_mask"][None], 4096, axis=0)
  return pymoc_amd.JN2018Ensemble(cfg, stream=st, arith=arith), 3600
for c in (args or ["5"]):
  c = int(c)
  for rep in range(2):
    e, steps = make(c)
    e.run(10 * e.M); st.sync()
    t0 = time.perf_counter(); e.run(steps); st.sync(); dt = time.perf_counter() - t0
    e.timer = LaunchTimer(); e.run(steps)
    print("config", c, arith, "%.4g coupled steps/s" % (e.n * steps / dt),
          {k: (n, round(1e3 * t / n, 2)) for k, (n, t) in e.timer.summary(st).items()}, "null span %.2f us" % (1e3 * e.timer.null_ms), flush=True)
    del e
