"""Config 4: one MOC interval (24 column steps, then Psi_SO.solve || thermal wind on two streams)
captured into a hipGraph and replayed, against the eager launches."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pymoc_amd
from pymoc_amd import configs
from pymoc_amd.device import Stream, Graph
st = Stream()
cfg = configs.config4()
for rep in range(2):
  for mode in ("eager", "graph1", "graph10"):
    e = pymoc_amd.TwoColEnsemble(cfg, stream=st)
    e.run(241); st.sync()
    if mode == "eager":
      t0 = time.perf_counter(); e.run(2400); st.sync(); t = time.perf_counter() - t0
    else:
      k = 1 if mode == "graph1" else 10
      with Graph.capture(st) as cap:
        for _ in range(k):
          e._steps(24)
          e._update()
      g = cap.graph
      st.sync()
      t0 = time.perf_counter()
      for _ in range(100 // k):
        g.launch(st)
      st.sync(); t = time.perf_counter() - t0
      e.ii += 2400
    s = e.state()
    print("%s: %.4g coupled steps/s, %.1f us per interval, checksum %.17g" % (mode, 8192 * 2400 / t, t / 100 * 1e6, float(np.sum(s["b_basin"]))), flush=True)
    del e
