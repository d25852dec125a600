#!/usr/bin/env python3
"""A Jansen & Nadeau (2018) parameter sweep on all GPUs of a node: the ensemble is sharded by
member (one process per GPU), time-stepped with no communication, and the per-member
diagnostics are gathered with RCCL and written in the reference's wire format
(run_JansenNadeau_2018.py:266-272), readable by examples/Plot_overturning.py of PyMOC.

    python examples/jn2018_sweep.py --members 256 --years 20 --out /tmp/sweep
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        examples/jn2018_sweep.py --members 32768 --years 100 --out /tmp/sweep
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pymoc_amd
from pymoc_amd import configs, sharding
from pymoc_amd.diagnostics import JN2018Diagnostics, save_pickup


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--members", type=int, default=256, help="ensemble size over all GPUs")
  ap.add_argument("--years", type=float, default=20.)
  ap.add_argument("--nz", type=int, default=200)
  ap.add_argument("--dt-days", type=float, default=10.,
                  help="time step; stay inside the reference's own stable range (nz=200: 10 d, "
                       "nz=81: 30 d -- DESIGN.md section 4)")
  ap.add_argument("--out", default=None, help="directory for diagnostics / pickups (rank 0)")
  args = ap.parse_args()
  rank, world, local_rank = sharding.world_info()
  pymoc_amd._lib.require_device(local_rank)
  comm = sharding.make_communicator()
  lo, hi = sharding.member_range(args.members, world, rank)
  cfg = configs.config5(N=args.members, nz=args.nz, dt_days=args.dt_days, members=(lo, hi))
  cfg["rest_mask"] = np.repeat(cfg["rest_mask"][None], hi - lo, axis=0)
  ens = pymoc_amd.JN2018Ensemble(cfg)
  total = int(np.ceil(args.years * 360 * 86400. / cfg["dt"]))
  total -= total % cfg["MOC_up_iters"]
  diag_iters = 10 * cfg["MOC_up_iters"]
  rec = JN2018Diagnostics(ens, diag_iters, total)
  ens.recorder = rec
  ens.run(total)
  st = ens.state()
  # the one exchange of the run: per-member output to rank 0's view of the whole ensemble
  psi_max = sharding.gather_members(comm, st["Psi"].max(axis=1, keepdims=True), args.members)
  b_mid = sharding.gather_members(comm, st["b_basin"][:, [cfg["z"].size // 2]], args.members)
  bad = sharding.gather_members(
      comm, np.isin(np.arange(hi - lo), ens.nonfinite_members())[:, None].astype(float),
      args.members)
  if rank == 0:
    # members the explicit scheme loses (the reference loses the same ones)
    # are reported, not folded into the ranges
    ok = bad[:, 0] == 0
    print("%d members on %d GPU(s), %d steps: max AMOC %.2f .. %.2f Sv, mid-depth b %.2e .. "
          "%.2e over the %d finite members, %d non-finite members"
          % (args.members, world, total, psi_max[ok].min(), psi_max[ok].max(), b_mid[ok].min(),
             b_mid[ok].max(), int(ok.sum()), int(bad.sum())))
  if args.out and rec.nd > 0:
    os.makedirs(args.out, exist_ok=True)
    rec.save_ensemble(os.path.join(args.out, "diags_rank%d.npz" % rank), cfg["tau"], cfg["KGM"])
    save_pickup(ens, os.path.join(args.out, "pickup_rank%d.npz" % rank))
    if rank == 0:  # member 0 in the reference's own single-run layout
      rec.save_member(os.path.join(args.out, "diags_member0.npz"), 0, cfg["tau"][0],
                      cfg["KGM"][0])
  comm.barrier()
  comm.close()


if __name__ == "__main__":
  main()
