#!/usr/bin/env python3
"""bench.py -- throughput of the HIP engine on BASELINE.json's configs, one JSON line.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5]

Default (`--config 2`, BASELINE configs[1], the configuration the metric is quoted on): an
ensemble of 1024 independent advective-diffusive Columns per GPU, nz=100, fp64.  A bench
"step" is one pass of the hot path over the rank's batch: ONE launch that advances every
column by `--steps-per-launch` (default 1000 = the config's whole job; wA is static) model
time steps.  `value` counts Column.timestep calls: columns x K x steps-per-launch / time.
At N=1 the same line also carries, under "coupled", the driver-run numbers of the coupled
configs 3, 4 and 5 at their SURVEY 8d sizes and full run lengths (2400 / 2400 / 3600 model
steps), each with its dominant kernel's roofline and a CPU baseline.

`--config 3|4|5` makes a coupled config the headline instead: a bench step is then one
MOC-update interval (MOC_up_iters model steps of both columns [+ the mixed layer] and one
refresh of the diagnostics); with N>1 every rank owns a fixed number of members (weak
scaling) and {b_basin, b_north, Psi_AMOC, Psi_SO} are all-gathered with RCCL on device
buffers every Diag_iters model steps and at the end, inside the timed region.

N>1: one process per GPU.  Either the driver's launcher
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N ...
or plain `python bench.py --gpus N ...`, which starts the N ranks itself through
pymoc_amd/launch.py (no torch anywhere in the product).  RCCL barrier + device sync on
both sides of the timed region, max over ranks, rank 0 prints the line.

The line is compact (numbers only; what every key means, the flop / byte models and where each
figure can be recomputed from profiles/ is DESIGN.md section 6) and ENDS with the coupled
configs' blocks "c3", "c4", "c5", so that a record keeping only the tail of the line still holds
them.  Every coupled roofline is computed from the RUN-AVERAGE duration of the dominant kernel:
HIP events around every launch of a full-length run (pymoc_amd.device.LaunchTimer).

The CPU baselines (oracle/, the plain-C restatement of the reference's algorithm driven
member by member like the reference's loops) are timed FIRST, before this process touches
the GPU: on 1 core and on all cores this process may use (`multiprocessing`, fork).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)
# what pymoc_amd.launch.child_env gives its ranks, for every other launcher
# (torch.distributed.run) too; pymoc_amd._lib repeats it right before it loads the engine
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBPS = 8000.0    # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_PEAK_TFLOPS = 78.6   # fp64 vector: 256 CU x 4 SIMD x 16 lanes/clk x 2 flop x 2.4 GHz

SIZES = {  # per-GPU members and run length of each config
    2: dict(members=1024, nsteps=1000, ncol=1),
    3: dict(members=4096, nsteps=2400, ncol=2),
    4: dict(members=8192, nsteps=2400, ncol=2),
    5: dict(members=4096, nsteps=3600, ncol=2),
    6: dict(members=2048, nsteps=2400, ncol=3),  # SURVEY 8f row N1 (two-basin), not a BASELINE config
}

# ------------------------------------------------------------------------- flop models
def flops_column_step(nz):
  """K1: ~14 flop per interior level incl. the 3 divisions (SURVEY 8d 'Flops')."""
  return 14.0 * (nz - 2)


def flops_thermwind_update(nz, nb):
  """K2 + K3 per member and update: solve 12 nz, Psib 6 nb (nz-1), Psibz 2 x 10 nz."""
  return 12.0 * nz + 6.0 * nb * (nz - 1) + 20.0 * nz


def flops_so_ml_step(ny):
  return 30.0 * ny


# ------------------------------------------------------------------------ CPU baselines
def usable_cores():
  """Cores this process may really use: affinity mask, capped by a cgroup CPU quota."""
  n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
  try:
    with open("/sys/fs/cgroup/cpu.max") as f:
      quota, period = f.read().split()
    if quota != "max":
      n = max(1, min(n, int(math.ceil(float(quota) / float(period)))))
  except Exception:
    pass
  return n


def _cpu_worker(job):
  """One worker = one core: repeat whole member runs of its slice until the budget is
  spent.  Returns (column-timesteps done, seconds)."""
  config, lo, hi, n_total, nz, budget = job
  import oracle as O
  from oracle import drivers as D
  from pymoc_amd import configs
  size = SIZES[config]
  if config == 2:
    c = configs.config2(N=n_total, nz=nz, members=(lo, hi))
    a = (c["z"], c["kappa"], c["Area"])
    chunk, done, b = 250, 0, c["b0"]
    O.column_ensemble_steps(*a, b[:1], c["wA"][:1], c["dt"], c["do_conv"][:1], c["bs"][:1],
                            c["bbot"][:1], c["N2min"][:1], 2)  # page in
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget:
      b = O.column_ensemble_steps(*a, b, c["wA"], c["dt"], c["do_conv"], c["bs"], c["bbot"],
                                  c["N2min"], chunk)
      done += chunk
    return (hi - lo) * done, time.perf_counter() - t0
  cfg = {3: configs.config3, 4: configs.config4, 5: configs.config5,
         6: configs.config_twobasin}[config](N=n_total, members=(lo, hi))
  nsteps = size["nsteps"]
  units, j = 0, 0
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < budget:
    m = configs.member(cfg, j % (hi - lo), config)
    if config == 3:
      D.run_twocol(m, nsteps, ())
    elif config == 4:
      D.run_twocol(m, nsteps, (), so=True)
    elif config == 6:
      D.run_twobasin(m, nsteps, ())
    else:
      D.run_jn2018(m, nsteps, ())
    units += size["ncol"] * nsteps
    j += 1
  return units, time.perf_counter() - t0


def cpu_baseline(config, nz, budget_1=6.0, budget_all=6.0, max_workers=0):
  """The oracle on the same workload: (i) 1 process = 1 core, (ii) one process per usable
  core.  Bounded samples: every worker repeats full-length runs of members of its slice of
  the ensemble for `budget` seconds."""
  import multiprocessing as mp
  size = SIZES[config]
  n = size["members"]
  unit = "column-timesteps/s"
  u1, t1 = _cpu_worker((config, 0, min(n, 64), n, nz, budget_1))
  cores = usable_cores()
  workers = min(cores, max_workers) if max_workers else cores
  workers = max(1, min(workers, n))
  per = n // workers
  jobs = [(config, w * per, (w + 1) * per, n, nz, budget_all) for w in range(workers)]
  ctx = mp.get_context("fork")
  t0 = time.perf_counter()
  with ctx.Pool(workers) as pool:
    res = pool.map(_cpu_worker, jobs, chunksize=1)
  wall = time.perf_counter() - t0
  rate_all = sum(u / t for u, t in res)
  what = {2: "oracle C port, 250-step launches of the column batch",
          3: "oracle, full 2400-step member runs",
          4: "oracle, full 2400-step member runs (adaptive GM mesh)",
          5: "oracle, full 3600-step member runs",
          6: "oracle, full 2400-step member runs"}[config]
  base = {"value": sig(u1 / t1), "unit": unit, "cores": 1, "kind": "port",
          "sample": "%s, members [0,%d), %.1f s" % (what, min(n, 64), t1),
          "all_cores": {"value": sig(rate_all), "cores": workers,
                        "sample": "%d workers x %.1f s" % (workers, budget_all)}}
  return base


# ------------------------------------------------------------------------ GPU helpers
def time_calls(fn, reps, stream, Event, warm=2):
  """ms per call of `fn` (HIP events on the stream the kernels are launched on)."""
  for _ in range(warm):
    fn()
  e0, e1 = Event(), Event()
  e0.record(stream)
  for _ in range(reps):
    fn()
  e1.record(stream)
  stream.sync()
  return e0.elapsed_ms(e1) / reps


def sig(x, n=5):
  """n significant digits (keeps the JSON line short)."""
  if x is None:
    return None
  x = float(x)
  return x if not math.isfinite(x) else float("%.*g" % (n, x))


def load_json(path):
  try:
    with open(path) as f:
      return json.load(f)
  except Exception:
    return {}


def run_steps(batch, wA, dt, nsteps, per_launch, lanes, arith="exact"):
  """Exactly nsteps steps as ceil(nsteps/per_launch) launches; returns launch count."""
  done, launches = 0, 0
  while done < nsteps:
    n = min(per_launch, nsteps - done)
    batch.steps(wA, dt, n, lanes_per_col=lanes, arith=arith)
    done += n
    launches += 1
  return launches


# ------------------------------------------------------------------------- config 2
WORKLOAD_SHORT = {
    2: "BASELINE configs[1]: %d independent Columns x nz=%d fp64 per GPU, static wA, do_conv on odd "
       "members, dt=30 d (pymoc_amd.configs.config2)",
    3: "BASELINE configs[2]: %d example_twocol.py members per GPU, nz=%d, MOC_up_iters=24, nb=500 "
       "(configs.config3; kappa_4k <= 2.5e-4: fixture G18)",
    4: "BASELINE configs[3]: %d example_twocol_plusSO.py members per GPU, nz=%d, ny=40, c=0.1 GM BVP "
       "on solve_bvp's adaptive mesh (configs.config4; A_basin >= 4.5e13: G18)",
    5: "BASELINE configs[4]: %d run_JansenNadeau_2018.py members per GPU, nz=%d, dt=10 d, ny=51, "
       "MOC_up_iters=36, nb=500 (configs.config5; db <= 8e-4: G18)",
    6: "SURVEY 8f row N1: %d twobasin_NadeauJansen.py members per GPU (3 columns, 2 thermal winds, "
       "2 SO sectors), nz=%d, MOC_up_iters=24, nb=500 (configs.config_twobasin)",
}


def profile_counters():
  """SQ / TCC counters of separate rocprofv3 --pmc passes, replayed (never measured in a bench
  run): the newest profiles/rNN/counters.json."""
  for r in ("r05", "r04", "r03"):
    d = load_json(os.path.join(ROOT, "profiles", r, "counters.json"))
    if d:
      return d, r
  return {}, None


def bench_config2(args, env):
  pymoc_amd, configs, DeviceArray, Event = (env["pymoc_amd"], env["configs"],
                                            env["DeviceArray"], env["Event"])
  stream, comm, rank, world = env["stream"], env["comm"], env["rank"], env["world"]
  C, nz, F, K, W = args.members or 1024, args.nz, args.steps_per_launch, args.steps, args.warmup
  cfg = configs.config2(N=world * C, nz=nz, members=(rank * C, (rank + 1) * C))
  batch = pymoc_amd.ColumnBatch(cfg["z"], cfg["kappa"], cfg["Area"], cfg["b0"],
                                bs=cfg["bs"], bbot=cfg["bbot"], N2min=cfg["N2min"],
                                do_conv=cfg["do_conv"], stream=stream)
  wA = DeviceArray.from_host(cfg["wA"], stream=stream)
  # diagnostic output of the job: the final buoyancy of every member gathered into one buffer.
  # EVERY N pays it inside the timed region the same way -- pack on the compute stream, the
  # exchange on the communication stream (N = 1 without a communicator: the pack alone) -- and
  # every rank's clock stops when ITS device has finished (all launches done, gather landed);
  # the closing barrier comes after, so that its own latency is not read as scaling loss.
  use_gather = world > 1 or args.force_rccl
  diag = env["make_gather"](comm if use_gather else None, C, world * C, [("b", nz)], stream,
                            getattr(args, "gather", "all"),
                            not getattr(args, "gather_inline", False))
  dt = cfg["dt"]

  run_steps(batch, wA, dt, W * F, F, args.lanes)
  diag.gather(dict(b=batch.b))  # RCCL sets its channels up at the first call: not timed
  diag.wait()
  comm.max_host(0.0)
  stream.sync()

  ev0, ev1 = Event(), Event()
  comm.barrier(stream)
  pymoc_amd.synchronize()
  t0 = time.perf_counter()
  ev0.record(stream)
  launches = run_steps(batch, wA, dt, K * F, F, args.lanes)
  ev1.record(stream)
  diag.gather(dict(b=batch.b))
  stream.sync()
  diag.wait()
  t_local = time.perf_counter() - t0
  comm.barrier(stream)
  pymoc_amd.synchronize()
  t_closed = time.perf_counter() - t0
  elapsed = comm.max_host(t_local)
  elapsed_closed = comm.max_host(t_closed)
  kernel_ms = ev0.elapsed_ms(ev1)
  nonfinite = int(batch.get_nonfinite().sum())
  b_final = batch.get_b()
  # the opt-in tolerance mode (PM_OP_CONTRACTED) on a copy of the same ensemble, same K launches
  # (every rank runs it: no collective inside)
  tol = pymoc_amd.ColumnBatch(cfg["z"], cfg["kappa"], cfg["Area"], cfg["b0"], bs=cfg["bs"],
                              bbot=cfg["bbot"], N2min=cfg["N2min"], do_conv=cfg["do_conv"],
                              stream=stream)
  run_steps(tol, wA, dt, W * F, F, args.lanes, arith="contracted")
  stream.sync()
  ev2, ev3 = Event(), Event()
  ev2.record(stream)
  run_steps(tol, wA, dt, K * F, F, args.lanes, arith="contracted")
  ev3.record(stream)
  stream.sync()
  tol_ms = ev2.elapsed_ms(ev3) / launches
  tol_b = tol.get_b()
  tol_err = float(np.max(np.abs(tol_b - b_final)) / np.max(np.abs(b_final)))
  del tol
  if rank != 0:
    return None

  G, P = batch.kernel_shape(args.lanes)
  value = world * C * K * F / elapsed
  launch_s = kernel_ms * 1e-3 / launches
  flop = flops_column_step(nz) * C * F
  alg_bytes = 24.0 * nz * C * F  # read b, read wA, write b per model step
  tf = flop / launch_s / 1e12
  prof, prof_round = profile_counters()
  kname = batch.kernel_name(F, args.lanes)
  cnt = prof.get("c2/" + kname) or {}
  out = {
      "metric": "column-timesteps/sec (ensemble) at nz=%d" % nz,
      "value": sig(value, 6), "unit": "column-timesteps/s", "n_gpus": world, "steps": K,
      "warmup": W, "ms_per_step": sig(elapsed * 1e3 / K), "higher_is_better": True,
      "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
      "config": {"workload": WORKLOAD_SHORT[2] % (C, nz), "columns_per_gpu": C, "nz": nz,
                 "model_steps_per_step": F, "kernel": kname,
                 "parallelism": "ensemble sharded over %d GPU(s), RCCL gather (%s) of the final "
                                "state" % (world, getattr(args, "gather", "all"))},
      "roofline": {"bound": "fp64-valu", "achieved": sig(tf), "peak": FP64_PEAK_TFLOPS,
                   "unit": "TFLOP/s", "frac": sig(tf / FP64_PEAK_TFLOPS, 4),
                   "traffic": cnt.get("hbm_bytes_per_launch"), "kernel_us": sig(launch_s * 1e6),
                   "issue_frac": cnt.get("issue_frac"), "counters": prof_round,
                   "alg_bytes": alg_bytes, "alg_hbm_GBps": sig(alg_bytes / launch_s / 1e9)},
      "timing": {"on_stream_ms_per_step": sig(kernel_ms / K), "ms_per_step_incl_closing_barrier":
                 sig(elapsed_closed * 1e3 / K), "gather": getattr(args, "gather", "all"),
                 "gather_overlap": not getattr(args, "gather_inline", False),
                 "collectives": getattr(diag, "ncollectives", 0)},
      "nonfinite": nonfinite, "checksum": float(np.sum(b_final)),
      "contracted": {"value": sig(C * F / (tol_ms * 1e-3)), "kernel_us": sig(tol_ms * 1e3),
                     "frac": sig(flop / (tol_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4),
                     "max_rel_diff": sig(tol_err, 3)},
  }
  if not args.no_single_step:
    # one step per launch (b, wA and the static coefficients cross HBM/L2 every step): the
    # regime of BASELINE config 1 and of user loops that refresh Psi every step
    ms1 = time_calls(lambda: batch.steps(wA, dt, 1, lanes_per_col=args.lanes), 2000, stream,
                     Event, warm=200)
    out["single_step_us"] = sig(ms1 * 1e3)
    # the same one-step launches replayed from a captured hipGraph of 64 nodes (one host call per
    # 64 steps): the device-side cost of a kernel boundary, without the host's launch path
    from pymoc_amd.device import Graph
    with Graph.capture(stream) as cap:
      for _ in range(64):
        batch.steps(wA, dt, 1, lanes_per_col=args.lanes)
    msg = time_calls(lambda: cap.graph.launch(stream), 40, stream, Event, warm=5)
    out["single_step_graph_us"] = sig(msg * 1e3 / 64)
  if not args.no_single_step and world == 1 and C == 1024 and nz == 100:
    # the SAME kernel where HBM does bind: one step per launch on an ensemble far beyond the
    # caches (262144 columns: 0.84 ... 1.26 GB cross HBM per launch)
    Cb = 262144
    cb = configs.config2(N=Cb, nz=nz)
    big = pymoc_amd.ColumnBatch(cb["z"], cb["kappa"], cb["Area"], cb["b0"], bs=cb["bs"],
                                bbot=cb["bbot"], N2min=cb["N2min"], do_conv=cb["do_conv"],
                                stream=stream)
    wAb = DeviceArray.from_host(cb["wA"], stream=stream)
    # the C-ABI default first: no hints, every array streamed (48 nz B per column-step)
    big.use_hints(uniform_area=False)
    msb = time_calls(lambda: big.steps(wAb, dt, 1, lanes_per_col=args.lanes), 20, stream,
                     Event, warm=3)
    big.use_hints()
    # forcing precombined once per overturning update (PM_OP_WEFF) + uniform Area:
    # b, weff, kappa in, b out = 32 nz B per column-step
    weffb = big.combine_forcing(wAb)
    msw = time_calls(lambda: big.steps(weffb, dt, 1, lanes_per_col=args.lanes, precombined=True),
                     20, stream, Event, warm=3)
    # ... and with the sweep's structure handed over: kappa = kappa_back[member] + profile[level]
    # (how config 2 builds it) is FORMED on the device and every column's Area is one number:
    # b and weff in, b out = 24 nz B per column-step, SURVEY 8d's algorithmic bytes; bit-identical
    # (tests/test_column_gpu.py::test_affine_kappa_hint_bitwise)
    aff = pymoc_amd.ColumnBatch(cb["z"], cb["kappa"], cb["Area"], cb["b0"], bs=cb["bs"],
                                bbot=cb["bbot"], N2min=cb["N2min"], do_conv=cb["do_conv"],
                                stream=stream, kappa_affine=(cb["kappa_back"], cb["kappa_profile"]))
    weffa = aff.combine_forcing(wAb)
    msa = time_calls(lambda: aff.steps(weffa, dt, 1, lanes_per_col=args.lanes, precombined=True),
                     20, stream, Event, warm=3)
    # (PM_COLS_DIV3_PROVEN: 3-instruction quotients; 8 columns per wave as straight-line code when
    # the batch divides: column.hip.h, launch_column_steps)
    sname = "k_column_stream<2,5,true,true,true,%s,%d,-1,-1>" % (
        "true" if aff.div3_proven else "false", 8 if Cb % 8 == 0 and Cb >= 8 * 4096 else 0)
    del aff, weffa
    gbps = 24.0 * nz * Cb / (msa * 1e-3) / 1e9
    cs = (prof.get("c2s/" + sname) or prof.get("c2s/k_column_stream<2,5,true,true,true>")
          or prof.get("c2s/k_column_stream<2,5,true,true>") or {})
    out["hbm_regime"] = {
        "bound": "hbm", "achieved": sig(gbps), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": sig(gbps / HBM_PEAK_GBPS, 4), "traffic": cs.get("hbm_bytes_per_launch"),
        "columns": Cb, "bytes_per_column_step": 24 * nz, "kernel": sname,
        "kernel_us": sig(msa * 1e3), "value": sig(Cb / (msa * 1e-3)),
        "kappa_streamed_32nz": {"kernel_us": sig(msw * 1e3), "value": sig(Cb / (msw * 1e-3)),
                                "frac": sig(32.0 * nz * Cb / (msw * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
        "all_streamed_48nz": {"kernel_us": sig(msb * 1e3),
                              "frac": sig(48.0 * nz * Cb / (msb * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}}
    del big, wAb, weffb
  return out


# ---------------------------------------------------------------------- configs 3, 4, 5
def make_ensemble(config, env, members, comm=None, n_total=None, **kw):
  pymoc_amd, configs = env["pymoc_amd"], env["configs"]
  rank, world, stream = env["rank"], env["world"], env["stream"]
  n_total = n_total or members
  lo = rank * members if comm is not None else 0
  sl = (lo, lo + members)
  kw = dict(kw, stream=stream)
  if comm is not None:
    kw.update(comm=comm, n_total=n_total)
    kw.update(env.get("gather_kw", {}))
  if config == 3:
    cfg = configs.config3(N=n_total, members=sl)
    ens = pymoc_amd.TwoColEnsemble(cfg, diag_iters=240, arith=env.get("arith", "exact"), **kw)
  elif config == 4:
    cfg = dict(configs.config4(N=n_total, members=sl), bvp_refine=env.get("bvp_refine", 0))
    ens = pymoc_amd.TwoColEnsemble(cfg, arith=env.get("arith", "exact"), **kw)
  elif config == 6:
    cfg = configs.config_twobasin(N=n_total, members=sl)
    ens = pymoc_amd.TwoBasinEnsemble(cfg, diag_iters=240, arith=env.get("arith", "exact"), **kw)
  else:
    cfg = configs.config5(N=n_total, members=sl)
    cfg["rest_mask"] = np.repeat(cfg["rest_mask"][None], members, axis=0)
    ens = pymoc_amd.JN2018Ensemble(cfg, arith=env.get("arith", "exact"), **kw)
  return cfg, ens


KERNEL_MODELS = {
    # per LAUNCH of the kernel: (flop, what one launch is)
    "k_thermwind": lambda n, nz, ny, nb, M: n * flops_thermwind_update(nz, nb),
    # Psi_SO.solve + thermal wind of a member in one launch: the thermal wind's count
    "k_so_tw_update": lambda n, nz, ny, nb, M: n * flops_thermwind_update(nz, nb),
    "k_column_steps": lambda n, nz, ny, nb, M, ncol=2: ncol * n * M * flops_column_step(nz),
    "k_jn2018_steps": lambda n, nz, ny, nb, M: n * M * (2 * flops_column_step(nz) +
                                                        flops_so_ml_step(ny)),
}


def kernel_breakdown(config, env, members, nsteps, warm_blocks, overlap=False, **kw):
  """Run-average duration of every kernel of a coupled config: a fresh ensemble, `warm_blocks`
  untimed MOC intervals, then the full run with HIP events around EVERY launch (LaunchTimer).  Returns
  ({kernel: [launches, avg us]}, roofline of the kernel with the largest total)."""
  from pymoc_amd.device import LaunchTimer
  # (config 4: Psi_SO.solve and the thermal wind one after the other here -- side by side, as the
  # driver runs them, each is stretched by the other and neither duration is the kernel's own;
  # likewise the two-basin update's two pairs)
  if config in (4, 6):
    kw = dict(kw, overlap_updates=bool(overlap))
  cfg, ens = make_ensemble(config, env, members, **kw)
  ens.run(warm_blocks * ens.M)
  env["stream"].sync()
  ens.timer = LaunchTimer()
  ens.run(nsteps)
  summ = ens.timer.summary(env["stream"])
  null_us = sig(1e3 * ens.timer.null_ms, 3)
  ens.timer = None
  kern = {k: [n, sig(1e3 * t / n, 4)] for k, (n, t) in summ.items()}
  dom = max((k for k in summ if k != "k_column_steps_short"), key=lambda k: summ[k][1])
  us = 1e3 * summ[dom][1] / summ[dom][0]
  n, nz, M, nb, ny = ens.n, ens.nz, ens.M, ens.nb, getattr(ens, "ny", 0)
  roof = {"bound": "fp64-valu", "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "kernel": dom,
          "kernel_us": sig(us, 4), "traffic": None, "event_null_us": null_us}
  prof, prof_round = profile_counters()
  cnt = None
  for kk, vv in prof.items():
    if kk.startswith("c%d/" % config) and kk.split("/", 1)[1].startswith(
        "k_jn2018" if dom == "k_jn2018_steps" else dom):
      if cnt is None or vv.get("avg_us", 0) * vv.get("launches", 0) > cnt.get("avg_us", 0) * cnt.get("launches", 0):
        cnt = vv
  issued = cnt["fp64_flop_issued_per_launch"] / (us * 1e-6) / 1e12 if cnt else None
  if dom in KERNEL_MODELS:
    extra = {"ncol": SIZES[config]["ncol"]} if dom == "k_column_steps" else {}
    tf = KERNEL_MODELS[dom](n, nz, ny, nb, M, **extra) / (us * 1e-6) / 1e12
  else:
    # the adaptive GM solve has no closed-form count: its figure is the fp64 flop the SQ counted
    # as issued (replayed counters), over this run's average kernel time
    tf = issued
  roof["achieved"] = sig(tf)
  roof["frac"] = sig(tf / FP64_PEAK_TFLOPS, 4) if tf is not None else None
  if cnt:
    roof["issued_frac"] = sig(issued / FP64_PEAK_TFLOPS, 4)
    roof["traffic"] = cnt.get("hbm_bytes_per_launch")
    roof["counters"] = prof_round
  del ens
  return kern, roof


def bench_coupled(config, args, env, members, nsteps=None, warm_blocks=None, sharded=False,
                  record=None, **kw):
  """Time a coupled config.  nsteps=None: K MOC intervals (headline mode).
  `record` (config 5): attach the diagnostics recorder -- the seven time series of
  run_JansenNadeau_2018.py:192-226 sampled every Diag_iters steps for `record` members ("all" or
  a count, gathered on the device) -- and end the timed region when the series are in host
  memory."""
  pymoc_amd = env["pymoc_amd"]
  stream, comm, rank, world = env["stream"], env["comm"], env["rank"], env["world"]
  use_comm = comm if sharded else None
  cfg, ens = make_ensemble(config, env, members, use_comm,
                           world * members if sharded else None, **kw)
  M = ens.M
  steps = nsteps if nsteps is not None else args.steps * M
  warm = (warm_blocks if warm_blocks is not None else args.warmup) * M
  if record is not None:
    from pymoc_amd.diagnostics import JN2018Diagnostics
    sel = None if record == "all" else np.linspace(0, members - 1, int(record)).astype(int)
    ens.recorder = JN2018Diagnostics(ens, ens.diag_iters, warm + steps + ens.diag_iters,
                                     members=sel, flush_every=1)
  ens.run(warm)
  if use_comm is not None:
    ens.gather_diagnostics()  # RCCL channel set-up of this collective: not timed
    ens.diag.wait()
    comm.max_host(0.0)
  stream.sync()
  comm.barrier(stream)
  pymoc_amd.synchronize()
  g0 = ens.diag.ngathers if ens.diag is not None else 0
  c0 = getattr(ens.diag, "ncollectives", 0) if ens.diag is not None else 0
  t0 = time.perf_counter()
  ens.run(steps)
  if use_comm is not None:
    ens.gather_diagnostics()  # the final output gather
  stream.sync()
  if use_comm is not None:
    ens.diag.wait()           # ... landed (it runs on the communication stream)
  if record is not None:
    ens.recorder.ts.wait()    # the time series are in (page-locked) host memory
  t_local = time.perf_counter() - t0
  comm.barrier(stream)
  pymoc_amd.synchronize()
  el = comm.max_host(t_local)
  bad = ens.nonfinite_members()
  tot = world * members if sharded else members
  res = {"members_per_gpu": members, "nz": int(ens.nz), "model_steps": steps, "M": M,
         "seconds": sig(el), "steps_per_s": sig(tot * steps / el, 6),
         "nonfinite": [int(cfg["members"][i]) for i in bad[:8]]}
  if use_comm is not None:
    res["gathers_in_timed_region"] = ens.diag.ngathers - g0
    res["rccl_collectives_in_timed_region"] = getattr(ens.diag, "ncollectives", 0) - c0
    res["gather_bytes_per_rank"] = ens.diag.bytes_per_rank
  return res, ens


def headline_coupled(config, args, env):
  members = args.members or SIZES[config]["members"]
  res, ens = bench_coupled(config, args, env, members, sharded=True)
  M, nz = ens.M, int(ens.nz)
  diag_iters = ens.diag_iters
  del ens
  roof = None
  if env["rank"] == 0:  # (no collective inside: the other ranks go straight to the final barrier)
    _, roof = kernel_breakdown(config, env, members, args.steps * M, args.warmup)
  if env["rank"] != 0:
    return None
  world, K, W = env["world"], args.steps, args.warmup
  ncol = SIZES[config]["ncol"]
  out = {
      "metric": "column-timesteps/sec (ensemble) at nz=%d" % nz,
      "value": sig(ncol * res["steps_per_s"], 6), "unit": "column-timesteps/s", "n_gpus": world,
      "steps": K, "warmup": W, "ms_per_step": sig(res["seconds"] * 1e3 / K),
      "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
      "data": "synthetic",
      "config": {"workload": WORKLOAD_SHORT[config] % (members, nz), "members_per_gpu": members,
                 "nz": nz, "model_steps_per_step": M,
                 "parallelism": "ensemble sharded over %d GPU(s); RCCL gather (%s, %s) of "
                                "{b_basin,b_north,Psi_AMOC,Psi_SO} every %s model steps and at "
                                "the end" % (world, args.gather, "in line" if args.gather_inline
                                             else "on a communication stream", diag_iters)},
      "roofline": roof,
  }
  out.update(res)
  return out


def coupled_block(c, args, env, cpu):
  """The compact block of a coupled config inside the default line: full-length run at the SURVEY
  8d size; exact (default path), contracted columns, and the variants the config has."""
  n, steps = SIZES[c]["members"], SIZES[c]["nsteps"]
  res, ens = bench_coupled(c, args, env, n, nsteps=steps, warm_blocks=10)
  st = ens.state()
  blk = {"members": n, "steps": steps, "steps_per_s": res["steps_per_s"],
         "nonfinite": res["nonfinite"],
         "checksum": float(np.nansum(st["b_Atl" if c == 6 else "b_basin"]))}
  del ens, st
  kern, roof = kernel_breakdown(c, env, n, steps, 10)
  blk.update(dom=roof["kernel"], dom_us=roof["kernel_us"], frac=roof["frac"],
             issued_frac=roof.get("issued_frac"), kernels=kern)
  rc, e2 = bench_coupled(c, args, dict(env, arith="contracted"), n, nsteps=steps, warm_blocks=10)
  blk["contracted"] = rc["steps_per_s"]
  del e2
  if c == 4:  # the fixed 8-fold GM mesh of round 1 (~1e-6 from the reference) for comparison
    r8, e2 = bench_coupled(4, args, dict(env, bvp_refine=8), n, nsteps=steps, warm_blocks=10)
    blk["fixed_mesh_R8"] = r8["steps_per_s"]
    del e2
  if c in (3, 5):  # the persistent per-member run kernel (one launch per Diag_iters stretch)
    try:
      rf, e2 = bench_coupled(c, args, env, n, nsteps=steps, warm_blocks=10, fused_run=True)
      blk["fused_run"] = rf["steps_per_s"]
      del e2
    except ValueError:
      blk["fused_run"] = None  # the phases' LDS does not fit at this shape
  if c == 5:  # the diagnostics recorder on: device-resident time series, asynchronous copies
    blk["recorder"] = {}
    for name, rec in (("sel64", 64), ("all", "all")):
      rr, e2 = bench_coupled(5, args, env, n, nsteps=steps, warm_blocks=10, record=rec)
      blk["recorder"][name] = rr["steps_per_s"]
      del e2
  if c in cpu:
    ncol = SIZES[c]["ncol"]
    blk["cpu1"] = sig(cpu[c]["value"] / ncol)
    blk["cpuN"] = sig(cpu[c]["all_cores"]["value"] / ncol)
    blk["cores"] = cpu[c]["all_cores"]["cores"]
  return blk


# ------------------------------------------------------------------------------ main
def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=20, help="timed bench steps")
  ap.add_argument("--warmup", type=int, default=5, help="untimed bench steps")
  ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5, 6))
  ap.add_argument("--steps-per-launch", type=int, default=1000,
                  help="config 2: model time steps fused in one launch (= one bench step)")
  ap.add_argument("--members", "--columns", type=int, default=0,
                  help="members (columns) per GPU; 0 = the SURVEY 8d size of the config")
  ap.add_argument("--nz", type=int, default=100, help="config 2 only")
  ap.add_argument("--lanes", type=int, default=0, help="config 2: lanes per column (0 = auto)")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-single-step", action="store_true")
  ap.add_argument("--no-coupled", action="store_true",
                  help="config 2 at N=1: skip the configs 3-5 entries")
  ap.add_argument("--cpu-workers", type=int, default=0, help="cap of the all-core baseline")
  ap.add_argument("--cpu-seconds", type=float, default=6.0, help="budget per baseline leg")
  ap.add_argument("--gather", choices=("all", "root"), default="root",
                  help="diagnostic exchange: ncclAllGather, or point-to-point gather to rank 0 "
                       "(the writer; what the reference's output cadence needs)")
  ap.add_argument("--gather-inline", action="store_true",
                  help="run the exchange on the compute stream (round 4's behaviour; A/B)")
  ap.add_argument("--breakdown-only", action="store_true",
                  help="coupled config: only the full-length run with HIP events around every "
                       "launch that the default line's c<N> block takes its kernel averages and "
                       "roofline from (what profiles/collect_r05.sh traces)")
  ap.add_argument("--overlap", action="store_true",
                  help="with --breakdown-only, configs 4 / 6: the update's launches side by side "
                       "on two streams, as the drivers run them (default there: one after the other)")
  ap.add_argument("--force-rccl", action="store_true",
                  help="use the RCCL communicator even with one rank (plumbing check)")
  args = ap.parse_args()

  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    # plain `python bench.py --gpus N`: start the N ranks ourselves, before anything here
    # touches the GPU (pymoc_amd/launch.py loaded by path: importing the package would
    # dlopen the engine)
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "pymoc_launch", os.path.join(ROOT, "pymoc_amd", "launch.py"))
    launch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(launch)
    sys.exit(launch.spawn([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                          args.gpus))

  rank = int(os.environ.get("RANK", "0"))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  if world != args.gpus:
    raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

  # CPU baselines first: nothing below this block has touched the GPU yet
  cpu = {}
  want_coupled = args.config == 2 and world == 1 and not args.no_coupled
  if world == 1 and rank == 0 and not args.no_cpu_baseline:
    for c in ([args.config] + ([3, 4, 5, 6] if want_coupled else [])):
      cpu[c] = cpu_baseline(c, args.nz if c == 2 else 100, args.cpu_seconds, args.cpu_seconds,
                            args.cpu_workers)

  import pymoc_amd
  from pymoc_amd import configs, sharding
  from pymoc_amd.device import DeviceArray, Event, Stream
  _, _, local_rank = sharding.world_info()
  pymoc_amd._lib.require_device(local_rank)
  stream = Stream()
  comm = (sharding.RcclCommunicator(stream=stream) if (args.force_rccl and world == 1)
          else sharding.make_communicator(stream=stream))
  env = dict(pymoc_amd=pymoc_amd, configs=configs, DeviceArray=DeviceArray, Event=Event,
             stream=stream, comm=comm, rank=rank, world=world,
             gather_kw=dict(gather=args.gather, gather_overlap=not args.gather_inline),
             make_gather=lambda cm, n, ntot, fields, st, mode, overlap: sharding.DiagnosticGather(
                 cm, n, ntot, fields, stream=st, mode=mode, overlap=overlap))

  if args.breakdown_only and args.config != 2:
    c = args.config
    n = args.members or SIZES[c]["members"]
    kern, roof = kernel_breakdown(c, env, n, SIZES[c]["nsteps"], 10, overlap=args.overlap)
    print(json.dumps({"config": c, "members": n, "steps": SIZES[c]["nsteps"],
                      "overlap": bool(args.overlap), "kernels": kern, "roofline": roof},
                     separators=(",", ":")), flush=True)
    comm.close()
    return
  if args.config == 2:
    out = bench_config2(args, env)
    blocks = {}
    if want_coupled and out is not None:
      for c in (6, 3, 4, 5):  # (c3 / c4 / c5 stay at the very end of the line)
        blocks["c%d" % c] = coupled_block(c, args, env, cpu)
  else:
    out = headline_coupled(args.config, args, env)
    blocks = {}
  if out is not None:
    if args.config in cpu:
      out["cpu_baseline"] = cpu[args.config]
    out.update(blocks)  # last: they land in the tail of the line
    print(json.dumps(out, separators=(",", ":")), flush=True)
  comm.barrier(stream)
  comm.close()


if __name__ == "__main__":
  main()
