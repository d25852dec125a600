"""K1 inside the two-column loops: launch time against the number of fused steps (config 3's
8192 columns = 4096 members x 2, nz = 100, forcing from a thermal-wind update)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pymoc_amd as gpu
from pymoc_amd import configs
from pymoc_amd.device import Stream, Event
st = Stream()
cfgno = int(sys.argv[1]) if len(sys.argv) > 1 else 3
e = gpu.TwoColEnsemble(configs.config3() if cfgno == 3 else configs.config4(), stream=st)
e.run(241)
st.sync()
for k in (3, 6, 12, 24, 48, 96):
  for _ in range(3):
    e._steps(k)
  st.sync()
  e0, e1 = Event(), Event()
  R = 20
  e0.record(st)
  for _ in range(R):
    e._steps(k)
  e1.record(st)
  st.sync()
  us = e0.elapsed_ms(e1) * 1e3 / R
  print("config %d: %3d steps per launch: %7.1f us per launch, %6.3f us per step" % (cfgno, k, us, us / k), flush=True)
