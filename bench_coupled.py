#!/usr/bin/env python3
"""bench_coupled.py -- throughput of the coupled drivers (BASELINE configs 3, 4, 5; 6 = the
two-basin topology of SURVEY 8f row N1) on one GPU.  Secondary to bench.py (which measures the headline metric on config 2); prints one
JSON line per config with column-timesteps/s and coupled steps/s.

  python bench_coupled.py [--configs 3 4 5] [--members N] [--steps K]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--configs", type=int, nargs="*", default=[3, 4, 5, 6])
  ap.add_argument("--members", type=int, default=0, help="0 = the per-GPU size of SURVEY 8d")
  ap.add_argument("--steps", type=int, default=0)
  ap.add_argument("--no-graph", action="store_true")
  ap.add_argument("--bvp-refine", type=int, default=0,
                  help="config 4: 0 = solve_bvp's adaptive GM mesh (reference parity, as bench.py); "
                       "R > 0 = fixed R-fold mesh")
  ap.add_argument("--unfused", action="store_true",
                  help="config 5: one launch per component and step instead of the fused loop")
  args = ap.parse_args()
  import numpy as np
  import pymoc_amd
  from pymoc_amd import configs
  pymoc_amd._lib.require_device()
  for c in args.configs:
    if c == 3:
      n = args.members or 4096
      cfg = configs.config3(N=n)
      ens = pymoc_amd.TwoColEnsemble(cfg)
      steps, warm, ncol = args.steps or 2400, 241, 2
    elif c == 4:
      n = args.members or 8192
      cfg = dict(configs.config4(N=n), bvp_refine=args.bvp_refine)
      ens = pymoc_amd.TwoColEnsemble(cfg)
      steps, warm, ncol = args.steps or 2400, 241, 2
    elif c == 6:  # two-basin topology, 3 columns per member
      n = args.members or 2048
      cfg = configs.config_twobasin(N=n)
      ens = pymoc_amd.TwoBasinEnsemble(cfg)
      steps, warm, ncol = args.steps or 2400, 241, 3
    else:
      n = args.members or 4096
      cfg = configs.config5(N=n)
      cfg["rest_mask"] = np.repeat(cfg["rest_mask"][None], n, axis=0)
      ens = pymoc_amd.JN2018Ensemble(cfg, use_graph=not args.no_graph, fused=not args.unfused)
      steps, warm, ncol = args.steps or 3600, 2 * cfg["MOC_up_iters"], 2
    ens.run(warm)
    pymoc_amd.synchronize()
    t0 = time.perf_counter()
    ens.run(steps)
    pymoc_amd.synchronize()
    el = time.perf_counter() - t0
    bad = int(ens.nonfinite_members().size)
    print(json.dumps({
        "config": c, "members": n, "nz": int(cfg["z"].size), "steps": steps,
        "seconds": el, "coupled_steps_per_s": n * steps / el,
        "column_timesteps_per_s": ncol * n * steps / el,
        "ms_per_step": el * 1e3 / steps, "nonfinite_members": bad}), flush=True)


if __name__ == "__main__":
  main()
