"""Probe: the memory-bound regime of the column step (one step per launch, 262144 columns x
nz=100 = 1.26 GB per launch).  PYMOC_STREAM_CPW=n forces n columns per wave (1 = the
k_column_steps path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pymoc_amd
from pymoc_amd import configs
from pymoc_amd.device import DeviceArray, Event
N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 100
c = configs.config2(N=N, nz=nz)
b = pymoc_amd.ColumnBatch(c["z"], c["kappa"], c["Area"], c["b0"], bs=c["bs"], bbot=c["bbot"],
                          N2min=c["N2min"], do_conv=c["do_conv"])
wA = DeviceArray.from_host(c["wA"])
for _ in range(3): b.steps(wA, c["dt"], 1)
e0, e1 = Event(), Event()
e0.record()
for _ in range(20): b.steps(wA, c["dt"], 1)
e1.record(); pymoc_amd.synchronize()
ms = e0.elapsed_ms(e1) / 20
print("cpw=%s N=%d nz=%d: %.1f us per step, %.2f TB/s (48 nz B per column-step)" % (
    os.environ.get("PYMOC_STREAM_CPW", "auto"), N, nz, ms * 1e3, 48.0 * nz * N / (ms * 1e-3) / 1e12))
