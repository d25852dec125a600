"""K1 (k_column_steps<64,2,2,true,true>, 96 registers = 5 waves per SIMD) inside configs 3 / 4 with
the occupancy capped by unused LDS (PYMOC_K1_LDS bytes per 4-wave block): 8 / 16 waves per SIMD
run as rounds of 5+3 / 5+5+5+1 at full occupancy, 4+4 / 4x4 at four.  Child process per setting."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for rep in range(2):
  for lds in ("0", "40000", "54000"):
    env = dict(os.environ, PYMOC_K1_LDS=lds)
    print("== PYMOC_K1_LDS=%s" % lds, flush=True)
    subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "r04", "probe_kernels.py")] + (sys.argv[1:] or ["3", "4"]), env=env, check=True)
