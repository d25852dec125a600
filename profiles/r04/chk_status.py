import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, pymoc_amd
from pymoc_amd import configs
c = configs.config5(N=64)
c["rest_mask"] = np.repeat(c["rest_mask"][None], 64, axis=0)
e = pymoc_amd.JN2018Ensemble(c, fused_run=False)
e.run(72)
print("status nz=200:", np.unique(e.ml.status.download(), return_counts=True))
c = configs.config5(N=64, nz=81, ny=51, dt_days=30.)
c["rest_mask"] = np.repeat(c["rest_mask"][None], 64, axis=0)
for fr in (False, True):
  e = pymoc_amd.JN2018Ensemble(c, fused_run=fr)
  e.run(72)
  print("status nz=81 fused_run", fr, np.unique(e.ml.status.download(), return_counts=True))
c3 = configs.config3(N=64)
e = pymoc_amd.TwoColEnsemble(c3, fused_run=True); e.run(100)
print("twocol run status", np.unique(e.run_status.download(), return_counts=True))
