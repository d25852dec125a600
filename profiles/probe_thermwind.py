"""Probe: where k_thermwind spends its time (4096 members, nz=100, nb=500)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pymoc_amd
from pymoc_amd import configs, _lib
from pymoc_amd.device import DeviceArray, Event
c = configs.config3(N=4096)
b1, b2 = DeviceArray.from_host(c["b_basin0"]), DeviceArray.from_host(c["b_north0"])
for nb in (500, 64):
  t = pymoc_amd.ThermwindBatch(c["z"], 4096, f=c["f"], nb=nb)
  t.update(b1, b2)
  for label, ops in (("solve", 1), ("psib", 2), ("psib+psibz", 6), ("all", 7)):
    t.update(b1, b2, ops=ops)
    e0, e1 = Event(), Event()
    e0.record()
    for _ in range(20): t.update(b1, b2, ops=ops)
    e1.record(); pymoc_amd.synchronize()
    print("nb=%3d %-11s %.1f us per update" % (nb, label, e0.elapsed_ms(e1) / 20 * 1e3))
