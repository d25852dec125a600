#!/usr/bin/env python3
"""bench.py -- column-timesteps/s of the HIP column engine on BASELINE config 2.

Workload (configs[1] of BASELINE.json): an ensemble of 1024
independent advective-diffusive Columns per GPU, nz=100, fp64, prescribed static upwelling
wA, convective adjustment on odd members, dt=30 d.  A bench "step" is one pass of the hot
path over the rank's batch: ONE launch that advances every column by `--steps-per-launch`
(default 1000) model time steps -- the config's whole job, wA being static (the coupled
drivers fuse a MOC-update interval the same way).  `value` counts Column.timestep calls:
columns x steps x steps-per-launch / time.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; the launcher only provides RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*.
Ranks own disjoint member blocks (weak scaling: 1024 columns per GPU), synchronise with
an RCCL barrier on both sides of the timed region, and all-gather the final buoyancy
(the diagnostic output) with RCCL inside it.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def run_steps(batch, wA, dt, nsteps, per_launch, lanes):
  """Exactly nsteps steps as ceil(nsteps/per_launch) launches; returns launch count."""
  done, launches = 0, 0
  while done < nsteps:
    n = min(per_launch, nsteps - done)
    batch.steps(wA, dt, n, lanes_per_col=lanes)
    done += n
    launches += 1
  return launches


def cpu_baseline(cfg, budget_s=12.0):
  """The oracle (plain C port of the reference algorithm, 1 thread) on the same
  workload: whole 1024-column x 1000-step jobs until `budget_s` of CPU time is spent."""
  import oracle as O
  ncols, nz = cfg["b0"].shape
  O.column_ensemble_steps(cfg["z"], cfg["kappa"][:8], cfg["Area"][:8], cfg["b0"][:8],
                          cfg["wA"][:8], cfg["dt"], cfg["do_conv"][:8], cfg["bs"][:8],
                          cfg["bbot"][:8], cfg["N2min"][:8], 10)  # page in
  chunk = 250
  t0 = time.perf_counter()
  done = 0
  b = cfg["b0"]
  while time.perf_counter() - t0 < budget_s:
    b = O.column_ensemble_steps(cfg["z"], cfg["kappa"], cfg["Area"], b, cfg["wA"],
                                cfg["dt"], cfg["do_conv"], cfg["bs"], cfg["bbot"],
                                cfg["N2min"], chunk)
    done += chunk
  el = time.perf_counter() - t0
  return {"value": ncols * done / el, "unit": "column-timesteps/s", "cores": 1,
          "kind": "port",
          "sample": "%d columns x nz=%d x %d steps of the same config-2 workload in %.1f s "
                    "(oracle/pymoc_oracle.c, gcc -O2, 1 thread of %d host cores)" %
                    (ncols, nz, done, el, os.cpu_count())}


def load_traffic(path, key):
  try:
    with open(path) as f:
      return json.load(f).get(key)
  except Exception:
    return None


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=50, help="timed launches (bench steps)")
  ap.add_argument("--warmup", type=int, default=5, help="untimed launches")
  ap.add_argument("--steps-per-launch", type=int, default=1000,
                  help="model time steps fused in one launch (= one bench step)")
  ap.add_argument("--columns", type=int, default=1024, help="columns per GPU")
  ap.add_argument("--nz", type=int, default=100)
  ap.add_argument("--lanes", type=int, default=0, help="lanes per column (0 = auto)")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-single-step", action="store_true")
  ap.add_argument("--force-rccl", action="store_true",
                  help="use the RCCL communicator even with one rank (plumbing check)")
  args = ap.parse_args()

  import pymoc_amd
  from pymoc_amd import configs, sharding
  from pymoc_amd.device import DeviceArray, Event, Stream

  rank, world, local_rank = sharding.world_info()
  if world != args.gpus:
    raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch N>1 with torch.distributed.run "
                     "(one process per GPU)" % (args.gpus, world))
  pymoc_amd._lib.require_device(local_rank)
  stream = Stream()
  comm = (sharding.RcclCommunicator(stream=stream) if args.force_rccl
          else sharding.make_communicator(stream=stream))

  C, nz, F, K, W = args.columns, args.nz, args.steps_per_launch, args.steps, args.warmup
  lo, hi = rank * C, (rank + 1) * C
  cfg = configs.config2(N=world * C, nz=nz, members=(lo, hi))
  batch = pymoc_amd.ColumnBatch(cfg["z"], cfg["kappa"], cfg["Area"], cfg["b0"],
                                bs=cfg["bs"], bbot=cfg["bbot"], N2min=cfg["N2min"],
                                do_conv=cfg["do_conv"], stream=stream)
  wA = DeviceArray.from_host(cfg["wA"], stream=stream)
  use_gather = world > 1 or args.force_rccl
  gathered = DeviceArray((world, C, nz)) if use_gather else None
  dt = cfg["dt"]

  run_steps(batch, wA, dt, W * F, F, args.lanes)
  if use_gather:  # RCCL sets its channels up at the first call of each collective: not timed
    comm.allgather_device(batch.b, gathered, stream)
    comm.max_host(0.0)
  stream.sync()

  ev0, ev1 = Event(), Event()
  comm.barrier(stream)
  pymoc_amd.synchronize()
  t0 = time.perf_counter()
  ev0.record(stream)
  launches = run_steps(batch, wA, dt, K * F, F, args.lanes)
  ev1.record(stream)
  if use_gather:  # diagnostic output: RCCL all-gather of the final buoyancy
    comm.allgather_device(batch.b, gathered, stream)
  stream.sync()
  comm.barrier(stream)
  pymoc_amd.synchronize()
  elapsed = time.perf_counter() - t0
  elapsed = comm.max_host(elapsed)
  kernel_ms = ev0.elapsed_ms(ev1)

  nonfinite = int(batch.get_nonfinite().sum())
  b_final = batch.get_b()

  if rank == 0:
    lanes = args.lanes or 64
    value = world * C * K * F / elapsed
    launch_s = kernel_ms * 1e-3 / launches
    alg_bytes = 24.0 * nz * C * F  # read b, read wA, write b per model step
    achieved = alg_bytes / launch_s / 1e9
    traffic = load_traffic(os.path.join(ROOT, "profiles", "traffic_r01.json"),
                           "column_steps_F%d_C%d_nz%d" % (F, C, nz))
    out = {
        "metric": "column-timesteps/sec (ensemble) at nz=%d" % nz,
        "value": value, "unit": "column-timesteps/s", "n_gpus": world, "steps": K,
        "warmup": W, "ms_per_step": elapsed * 1e3 / K, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[1]: ensemble of %d independent Columns nz=%d "
                        "fp64 per GPU, static wA, do_conv on odd members, dt=30 d "
                        "(pymoc_amd.configs.config2, seed 20240)" % (C, nz),
            "columns_per_gpu": C, "nz": nz, "model_steps_per_step": F,
            "step": "one launch = %d Column.timestep calls of every column" % F,
            "lanes_per_column": lanes,
            "parallelism": "ensemble sharded over %d GPU(s), RCCL all-gather of the "
                           "final state" % world},
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
            "traffic_unit": "bytes per launch (FETCH_SIZE+WRITE_SIZE, calibrated; profiles/)",
            "kernel": "k_column_steps<%d,%d>" % (lanes, -(-nz // lanes)),
            "kernel_ms_per_launch": launch_s * 1e3, "launches": launches,
            "algorithmic_bytes_per_launch": alg_bytes,
            "note": "algorithmic bytes = 24*nz per column-step (SURVEY 8d); with %d steps "
                    "fused per launch the state stays in registers, so real HBM traffic "
                    "(`traffic`) is far below it" % F},
        "nonfinite_columns": nonfinite,
        "checksum": float(np.sum(b_final)),
    }
    if not args.no_single_step:
      # streaming mode for comparison: one step per launch (b, wA and the static
      # coefficients cross HBM/L2 every step); not part of `value`
      n1 = 2000
      run_steps(batch, wA, dt, 200, 1, args.lanes)
      stream.sync()
      e0, e1 = Event(), Event()
      e0.record(stream)
      run_steps(batch, wA, dt, n1, 1, args.lanes)
      e1.record(stream)
      stream.sync()
      ms1 = e0.elapsed_ms(e1) / n1
      out["single_step_launches"] = {
          "ms_per_step": ms1, "column_timesteps_per_s": C / (ms1 * 1e-3),
          "achieved_GBps_algorithmic": 24.0 * nz * C / (ms1 * 1e-3) / 1e9,
          "achieved_GBps_with_static_coefficients": 48.0 * nz * C / (ms1 * 1e-3) / 1e9}
    if not args.no_single_step and world == 1 and C == 1024 and nz == 100:
      # the SAME kernel where it is memory-bound: one step per launch on an ensemble far
      # beyond the caches (262144 columns: 1.26 GB cross HBM per launch); not part of `value`
      Cb = 262144
      cb = configs.config2(N=Cb, nz=nz)
      big = pymoc_amd.ColumnBatch(cb["z"], cb["kappa"], cb["Area"], cb["b0"], bs=cb["bs"],
                                  bbot=cb["bbot"], N2min=cb["N2min"], do_conv=cb["do_conv"],
                                  stream=stream)
      wAb = DeviceArray.from_host(cb["wA"], stream=stream)
      run_steps(big, wAb, dt, 3, 1, args.lanes)
      stream.sync()
      e0, e1 = Event(), Event()
      e0.record(stream)
      run_steps(big, wAb, dt, 20, 1, args.lanes)
      e1.record(stream)
      stream.sync()
      msb = e0.elapsed_ms(e1) / 20
      real = 48.0 * nz * Cb  # b, wA, kappa, Area, dAkappa in; b out
      out["streaming_262144_columns"] = {
          "ms_per_step": msb, "column_timesteps_per_s": Cb / (msb * 1e-3),
          "hbm_GBps": real / (msb * 1e-3) / 1e9,
          "frac_of_hbm_peak": real / (msb * 1e-3) / 1e9 / HBM_PEAK_GBPS,
          "achieved_GBps_algorithmic": 24.0 * nz * Cb / (msb * 1e-3) / 1e9}
      del big, wAb
    if world == 1 and not args.no_cpu_baseline:
      out["cpu_baseline"] = cpu_baseline(cfg)
    print(json.dumps(out), flush=True)
  comm.barrier(stream)
  comm.close()


if __name__ == "__main__":
  main()
