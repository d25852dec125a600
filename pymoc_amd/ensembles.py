"""Batched drivers: the coupled time loops of the reference's example scripts, run for a
whole parameter-sweep ensemble with all state resident in HBM.

The reference has no driver class: the loops live in examples/*.py.  The
classes here reproduce those loops' order of operations and cadence exactly:
  JN2018Ensemble           examples/run_JansenNadeau_2018.py:201-261 (config 5)
  TwoBasinEnsemble         examples/twobasin_NadeauJansen.py:99-122
  ColumnThermwindEnsemble  examples/example_timestepping.py:73-80   (BASELINE config 1)
  EquiIterationEnsemble    examples/example_iteration.py:59-68      
  TwoColEnsemble           examples/example_twocol.py:85-96         (config 3)
                           examples/example_twocol_plusSO.py:99-115 (config 4, with_so)
Each member is independent; `cfg` is a dict from `pymoc_amd.configs`.
"""
import numpy as np

from . import _lib
from .columns import ColumnBatch
from .device import DeviceArray, _sh, launch_span
from .equilibrium import ColumnEquiBatch
from .psi_so import PsiSOBatch
from .sharding import DiagnosticGather
from .so_ml import SOMLBatch
from .thermwind import ThermwindBatch

_TW_ALL = _lib.PM_TW_SOLVE | _lib.PM_TW_PSIB | _lib.PM_TW_PSIBZ


def _rows(v, n, nz):
  a = np.asarray(v, dtype=np.float64)
  if a.ndim == 0:
    return np.full((n, nz), a)
  if a.ndim == 1 and a.shape[0] == n and n != nz:
    return np.repeat(a[:, None], nz, axis=1)
  if a.ndim == 1:
    return np.broadcast_to(a, (n, nz)).copy()
  return a


def _run_fits(kind, nz, nb, ny):
  """Do the phases of a persistent run kernel (kind 0: pm_twocol_run, 1: pm_jn2018_run) fit the
  160 KB of LDS of a CU at 16 members per block?"""
  import ctypes as C
  nbytes = C.c_size_t(0)
  _lib.check(_lib.lib.pm_run_lds_bytes(int(kind), int(nz), int(nb), int(ny), C.byref(nbytes)))
  return 0 < nbytes.value <= 160 * 1024


def _vec(v, n):
  a = np.asarray(v, dtype=np.float64)
  return np.full(n, a) if a.ndim == 0 else a


class ColumnThermwindEnsemble(object):
  """One column per member, thermal wind against b2 = 0 re-solved after EVERY step and
  applied in z-space: wA = Psi * 1e6 (example_timestepping.py:73-80).

  Every model step is two launches (the column step, the thermal-wind solve), so a small
  ensemble is bound by the host's launch rate: `use_graph` (default) captures GRAPH_STEPS steps
  once into a hipGraph and replays it -- one host call per GRAPH_STEPS model steps; the kernels
  and their order are the same, so are the results."""
  GRAPH_STEPS = 32

  def __init__(self, cfg, n=None, stream=None, lanes_per_col=0, use_graph=True):
    z = cfg['z']
    nz = z.size
    b0 = np.atleast_2d(cfg['b0'])
    n = b0.shape[0] if n is None else n
    self.n, self.nz, self.dt = n, nz, float(cfg['dt'])
    self.lanes, self.stream = lanes_per_col, stream
    self.cols = ColumnBatch(z, _rows(cfg['kappa'], n, nz), _rows(cfg['Area'], n, nz),
                            _rows(b0, n, nz), bs=_vec(cfg['bs'], n), bbot=_vec(cfg['bbot'], n),
                            stream=stream)
    self.tw = ThermwindBatch(z, n, f=cfg['f'], nb=1, stream=stream, z_dev=self.cols.z)
    self.b2 = DeviceArray.zeros((n, nz), stream=stream)
    self.wA = DeviceArray.zeros((n, nz), stream=stream)
    self._use_graph, self._graph = bool(use_graph), None
    self._solve()

  def _solve(self):
    self.tw.update(self.cols.b, self.b2, ops=_lib.PM_TW_SOLVE | _lib.PM_TW_WA_PSI,
                   wA1=self.wA, nb=1)

  def _step(self):
    self.cols.steps(self.wA, self.dt, 1, lanes_per_col=self.lanes)
    self._solve()

  def run(self, nsteps):
    from .device import Graph
    remaining = int(nsteps)
    while self._use_graph and remaining >= self.GRAPH_STEPS:
      if self._graph is None:
        with Graph.capture(self.stream) as cap:
          for _ in range(self.GRAPH_STEPS):
            self._step()
        self._graph = cap.graph
      self._graph.launch(self.stream)
      remaining -= self.GRAPH_STEPS
    for _ in range(remaining):
      self._step()

  def state(self):
    return dict(b=self.cols.get_b(), Psi=self.tw.Psi.download(stream=self.stream))


class EquiIterationEnsemble(object):
  """One basin column per member: its equilibrium profile for the current overturning
  (`Column.solve_equi`) and the thermal-wind overturning against b_N = 0 for the relaxed
  profile, iterated (example_iteration.py:59-68):
      wA = AMOC.Psi*1e6;  basin.solve_equi(wA);
      AMOC.update(b1 = keep*AMOC.b1(z) + relax*basin.b);  AMOC.solve()"""

  def __init__(self, cfg, stream=None):
    z = cfg['z']
    nz = z.size
    b0 = np.atleast_2d(cfg['b_basin0'])
    self.n, self.nz = n, _ = b0.shape[0], nz
    self.keep, self.relax = float(cfg['keep']), float(cfg['relax'])
    self.stream = stream
    self.tw = ThermwindBatch(z, n, f=cfg['f'], nb=1, stream=stream)
    self.b1 = DeviceArray.from_host(_rows(b0, n, nz), stream=stream)
    self.b2 = DeviceArray.zeros((n, nz), stream=stream)
    self.wA = DeviceArray.zeros((n, nz), stream=stream)
    self.eq = ColumnEquiBatch.from_profiles(
        z, _rows(cfg['kappa'], n, nz), _rows(cfg['A_basin'], n, nz), _vec(cfg['bs'], n),
        _vec(cfg['bbot'], n), n=n, stream=stream, z_dev=self.tw.z)
    self._solve()

  def _solve(self):
    self.tw.update(self.b1, self.b2, ops=_lib.PM_TW_SOLVE | _lib.PM_TW_WA_PSI, wA1=self.wA,
                   nb=1)

  def iterate(self, niter=1):
    for _ in range(int(niter)):
      self.eq.solve(self.wA)
      _lib.check(_lib.lib.pm_axpby(self.n * self.nz, self.keep, self.b1.ptr, self.relax,
                                   self.eq.b.ptr, self.b1.ptr, _sh(self.stream)))
      self._solve()

  def state(self):
    return dict(b=self.eq.get_b(), bz=self.eq.get_bz(), b1=self.b1.download(stream=self.stream),
                Psi=self.tw.Psi.download(stream=self.stream))


class TwoColEnsemble(object):
  """Basin + northern sinking column per member, coupled by the thermal-wind overturning
  mapped to isopycnal space every MOC_up_iters steps."""

  def __init__(self, cfg, stream=None, lanes_per_col=0, comm=None, n_total=None,
               diag_iters=None, keep_history=False, arith="exact", overlap_updates=False,
               fused_run=None, gather="all", gather_overlap=True):
    """`fused_run`: carry the members through whole stretches of the loop -- many [refresh the
    overturning, MOC_up_iters steps] intervals -- in ONE launch of the persistent per-member
    kernel (pm_twocol_run), ending a launch only where the diagnostics are gathered.  Same device
    functions as the launch sequence, bit-identical results; needs: no SO channel, exact
    arithmetic, Area constant in z, the phases' LDS within 160 KB.  None = off (measured slower
    than the launch sequence on config 3: DESIGN.md section 6).
    `overlap_updates=True`: with an SO channel, Psi_SO.solve and the thermal wind of an update run
    side by side on two streams (bit-identical results; see `_update`).  Round 3's default; since
    the round-5 thermal wind (37 us beside a Psi_SO.solve of 113) the fork / join events cost what
    the overlap gives: config 4 203-204 us per interval side by side, 194-204 one after the other
    (profiles/r05/c4_modes.log).
    `comm` (a pymoc_amd.sharding communicator) makes this rank's members one shard of an
    `n_total`-member ensemble: stepping is unchanged (members never interact) and
    {b_basin, b_north, Psi, Psi_SO} are all-gathered on device buffers every `diag_iters`
    steps (default cfg['Diag_iters']) and by `gather_diagnostics()` at the end of a run;
    `gather="root"` sends them to rank 0 only, `gather_overlap` runs the exchange on a
    communication stream of its own beside the stepping (sharding.DiagnosticGather).
    `arith="contracted"`: the columns step in the opt-in tolerance mode (ColumnBatch.steps)."""
    z = cfg['z']
    nz = z.size
    n = np.atleast_2d(cfg['b_basin0']).shape[0]
    self.n, self.nz = n, nz
    self.dt, self.M, self.nb = float(cfg['dt']), int(cfg['MOC_up_iters']), int(cfg['nb'])
    self.lanes = lanes_per_col
    self.arith = arith
    self.stream = stream
    self.diag_iters = cfg.get('Diag_iters') if diag_iters is None else diag_iters
    self.diag = None
    self.timer = None  # optional device.LaunchTimer: events around every launch of run()
    if comm is not None or keep_history:
      self.diag = DiagnosticGather(comm, n, n if n_total is None else n_total,
                                   [(k, nz) for k in ('b_basin', 'b_north', 'Psi', 'Psi_SO')],
                                   stream=stream, keep_history=keep_history, mode=gather,
                                   overlap=gather_overlap)
    kap = _rows(cfg['kappa'], n, nz)
    # rows [0, n): basin columns, rows [n, 2n): northern columns
    self.cols = ColumnBatch(
        z, np.concatenate([kap, kap]),
        np.concatenate([_rows(cfg['A_basin'], n, nz), _rows(cfg['A_north'], n, nz)]),
        np.concatenate([_rows(cfg['b_basin0'], n, nz), _rows(cfg['b_north0'], n, nz)]),
        bs=np.concatenate([_vec(cfg['bs'], n), _vec(cfg['bs_north'], n)]),
        bbot=np.concatenate([_vec(cfg['bbot'], n), _vec(cfg['bbot'], n)]),
        do_conv=np.concatenate([np.zeros(n, bool), np.ones(n, bool)]), stream=stream)
    self.tw = ThermwindBatch(z, n, f=cfg['f'], nb=self.nb, stream=stream, z_dev=self.cols.z)
    self.wA = DeviceArray.zeros((2 * n, nz), stream=stream)
    self._off = n * nz * 8
    self.ii = 0
    self.so = None
    if 'y' in cfg and 'bs_SO' in cfg:  # example_twocol_plusSO.py:69-81
      ny = cfg['y'].size
      self.so = PsiSOBatch(z, cfg['y'], n, tau=cfg['tau'], KGM=cfg['KGM'], f=cfg['f'],
                           L=cfg['L'], c=cfg.get('c'), bvp_with_Ek=cfg.get('bvp_with_Ek', False),
                           bvp_refine=cfg.get('bvp_refine', 0), stream=stream,
                           z_dev=self.cols.z)
      self.bs_SO = DeviceArray.from_host(_rows(cfg['bs_SO'], n, ny) if np.ndim(cfg['bs_SO']) == 1
                                         else cfg['bs_SO'], stream=stream)
    self._zero_so = (DeviceArray.zeros((n, nz), stream=stream)
                     if self.so is None and self.diag is not None else None)
    self._overlap = self.so is not None and bool(overlap_updates)
    if self._overlap:
      from .device import Stream, Event
      import os
      self._side = Stream(high_priority=os.environ.get("PYMOC_SIDE_PRIORITY", "0") == "1")
      self._ev_fork, self._ev_join = Event(), Event()
    can_fuse = (self.so is None and arith == "exact" and self.cols.uniform_area and
                not self.cols.has_bzbot and
                _run_fits(0, nz, self.nb, 0))
    if fused_run and not can_fuse:
      raise ValueError("fused_run needs: no SO channel, exact arithmetic, Area constant in z, no "
                       "bzbot, 4 <= nz <= 256 and the phases' LDS within 160 KB")
    self._fused_run = False if fused_run is None else bool(fused_run)
    self.run_status = (DeviceArray.zeros((n,), np.int32, stream=stream) if self._fused_run
                       else None)
    self._update()  # AMOC.solve(); AMOC.Psibz() [; SO.solve()] on the initial profiles

  # device views
  @property
  def _b_basin(self):
    return self.cols.b.ptr

  @property
  def _b_north(self):
    return self.cols.b.ptr + self._off

  def _psi_so(self):
    return self.so.Psi if self.so is not None else None

  def _update(self):
    # SO.solve() and AMOC.solve() both read basin.b only.  With the SO channel the two launches
    # run SIDE BY SIDE on two streams (each leaves the machine partly idle: together 174 us
    # instead of 200 us per update of 8192 members) and the columns form
    # wAb = (Psi_iso_b - SO.Psi)*1e6, wAN = -Psi_iso_n*1e6 themselves (PM_OP_WA_PSI) once both
    # are done -- the same operations as the thermal-wind launch's wA1 / wA2 epilogue.
    if self.so is not None and self._overlap:
      self._ev_fork.record(self.stream)     # the columns' steps before this update
      self._side.wait(self._ev_fork)
      self.so.stream = self._side
      with launch_span(self.timer, "k_psi_so", self._side):
        self.so.update(self._b_basin, self.bs_SO)
      self.so.stream = self.stream
      self._ev_join.record(self._side)
      with launch_span(self.timer, "k_thermwind", self.stream):
        self.tw.update(self._b_basin, self._b_north, ops=_TW_ALL, store_psib=False)
      if self.stream is not None:
        self.stream.wait(self._ev_join)
      else:
        from ._lib import check, lib
        check(lib.pm_stream_wait_event(None, self._ev_join.handle))
      return
    if self.so is not None:
      with launch_span(self.timer, "k_psi_so", self.stream):
        self.so.update(self._b_basin, self.bs_SO)
    with launch_span(self.timer, "k_thermwind", self.stream):
      self.tw.update(self._b_basin, self._b_north, ops=_TW_ALL, store_psib=False,
                     Psi_SO=self._psi_so(),
                     wA1=self.wA.ptr, wA2=self.wA.ptr + self._off)

  def _steps(self, n):
    with launch_span(self.timer, "k_column_steps" if n >= 3 else "k_column_steps_short",
                     self.stream):
      self._steps_launch(n)

  def _steps_launch(self, n):
    if self.so is not None and self._overlap and n >= 3:
      self.cols.steps(None, self.dt, n, lanes_per_col=self.lanes, arith=self.arith,
                      psi_forcing=(self.tw.psibz, self.so.Psi))
      return
    if self.so is not None and self._overlap:  # a launch of 1-2 steps: the forcing as an array
      from ._lib import check, lib
      check(lib.pm_twocol_forcing(self.n, self.nz, self.tw.psibz.ptr, self.so.Psi.ptr,
                                  self.wA.ptr, _sh(self.stream)))
    self.cols.steps(self.wA, self.dt, n, lanes_per_col=self.lanes, arith=self.arith)

  def _run_fused(self, nsteps):
    """The same loop through pm_twocol_run: one launch per stretch that ends at a diagnostic
    gather (or at the end of the run)."""
    import ctypes as C
    M, remaining = self.M, int(nsteps)
    while remaining > 0:
      ii = self.ii
      k0 = ii if ii % M == 0 else (ii // M + 1) * M  # the next step followed by an update
      last = ii + remaining - 1                       # last step of this run
      sch = _lib.pm_run_schedule()
      sch.m_steps = M
      gather_at = None
      if k0 > last:
        sch.n_first, sch.n_updates, sch.n_last = remaining, 0, 0
        end = last + 1
      else:
        sch.n_first = k0 - ii + 1
        nu, k = 0, k0
        while True:
          nu += 1
          if self.diag is not None and self.diag.due(k, self.diag_iters):
            gather_at, tail = k, 0
            break
          if k + M > last:  # no further update inside this run
            tail = last - k
            break
          k += M
        sch.n_updates = nu
        sch.n_last = tail
        end = k + 1 + sch.n_last
      d = _lib.pm_twocol_loop()
      d.cols = self.cols.descriptor()
      d.tw = self.tw.descriptor(self._b_basin, self._b_north, wA1=self.wA.ptr,
                                wA2=self.wA.ptr + self._off, store_psib=False)
      d.wA, d.dt, d.sched, d.status = self.wA.ptr, self.dt, sch, self.run_status.ptr
      with launch_span(self.timer, "k_twocol_run", self.stream):
        _lib.check(_lib.lib.pm_twocol_run(C.byref(d), _sh(self.stream)))
      remaining -= end - ii
      self.ii = end
      if gather_at is not None:
        self.gather_diagnostics(gather_at)

  def run(self, nsteps):
    """`for ii in range(nsteps): step both columns; if ii % MOC_up_iters == 0: update`,
    with the steps between two updates fused into one launch (wA is constant there)."""
    if self._fused_run:
      return self._run_fused(nsteps)
    remaining = int(nsteps)
    while remaining > 0:
      nxt = self.ii if self.ii % self.M == 0 else (self.ii // self.M + 1) * self.M
      n = min(nxt - self.ii + 1, remaining)
      self._steps(n)
      self.ii += n
      remaining -= n
      if (self.ii - 1) % self.M == 0:
        self._update()
        if self.diag is not None and self.diag.due(self.ii - 1, self.diag_iters):
          self.gather_diagnostics(self.ii - 1)

  def gather_diagnostics(self, step=None):
    """All-gather {b_basin, b_north, Psi_AMOC, Psi_SO} of every rank's members (device
    buffers, one collective); `self.diag.last()` returns the assembled host arrays."""
    self.diag.gather(dict(b_basin=self._b_basin, b_north=self._b_north, Psi=self.tw.Psi,
                          Psi_SO=self.so.Psi if self.so is not None else self._zero_so),
                     step=self.ii if step is None else step)

  def state(self):
    b = self.cols.get_b()
    out = dict(b_basin=b[:self.n], b_north=b[self.n:], Psi=self.tw.Psi.download(stream=self.stream),
               Psi_iso_b=self.tw.psibz1.download(stream=self.stream), Psi_iso_n=self.tw.psibz2.download(stream=self.stream))
    if self.so is not None:
      out.update(Psi_SO=self.so.Psi.download(stream=self.stream), Psi_Ek=self.so.Psi_Ek.download(stream=self.stream),
                 Psi_GM=self.so.Psi_GM.download(stream=self.stream))
    return out

  def nonfinite_members(self):
    nf = self.cols.get_nonfinite()
    return np.nonzero(nf[:self.n] | nf[self.n:])[0]


class JN2018Ensemble(object):
  """run_JansenNadeau_2018.py with default flags: basin + north columns, thermal wind,
  SO channel overturning (no BVP smoother) and the SO mixed layer, with the script's
  per-step bottom-BC / bottom-boundary-layer diffusivity switching.

  Per step: BC switch -> both columns (convective adjustment on) -> mixed layer; every
  MOC_up_iters steps (before the step) the three diagnostics are refreshed.  With
  `use_graph` a whole MOC block is captured once into a hipGraph and replayed."""

  def __init__(self, cfg, stream=None, lanes_per_col=0, use_graph=False, fused=None,
               comm=None, n_total=None, diag_iters=None, keep_history=False, arith="exact",
               shared_coef=True, fused_run=None, gather="all", gather_overlap=True,
               split_lanes=False):
    """`fused_run`: whole stretches of the loop -- many [PsiSO.solve, AMOC.solve / Psibz,
    MOC_up_iters steps] intervals -- in ONE launch of the persistent per-member kernel
    (pm_jn2018_run), ending a launch only where diagnostics are sampled or gathered;
    bit-identical to the launch sequence; needs the fused step loop's conditions and the phases'
    LDS within 160 KB.  None = off (measured slower than the launch sequence: DESIGN.md section 6).
    `comm`, `n_total`, `diag_iters`, `gather`, `gather_overlap`: as for TwoColEnsemble; the gather happens where the
    script samples its diagnostics (`if ii % Diag_iters == 0`, right after the MOC update,
    run_JansenNadeau_2018.py:218-226; default Diag_iters = 10 MOC_up_iters, :99).
    `arith="contracted"`: the columns of the fused loop step in the opt-in tolerance mode
    (PM_JN_CONTRACTED; uniform-Area ensembles with ny <= 64 -- others stay exact).
    `shared_coef`: let the fused loop read ONE copy of kappa / d(A kappa)/dz / Area per column
    kind when all members' profiles are identical (checked here on the host arrays;
    PM_JN_SHARED_COEF); results are bit-identical either way.
    `split_lanes`: the fused loop steps both columns of a member together, one per half of the
    wavefront (PM_JN_SPLIT_LANES; bit-identical; measured a tie on config 5, hence opt-in)."""
    if arith not in ("exact", "contracted"):
      raise ValueError("arith must be 'exact' or 'contracted'")
    self.arith = arith
    self.shared_coef = bool(shared_coef)
    self.split_lanes = bool(split_lanes)
    import ctypes as C
    from ._lib import pm_jn2018_bc
    z, y = cfg['z'], cfg['y']
    nz, ny = z.size, y.size
    n = np.atleast_2d(cfg['b_basin0']).shape[0]
    self.n, self.nz, self.ny = n, nz, ny
    self.dt, self.M, self.nb = float(cfg['dt']), int(cfg['MOC_up_iters']), int(cfg['nb'])
    self.lanes, self.stream = lanes_per_col, stream
    bb0, bn0 = _rows(cfg['b_basin0'], n, nz), _rows(cfg['b_north0'], n, nz)
    kap = np.broadcast_to(np.asarray(cfg['kappa'], dtype=np.float64), (n, nz))
    kapeff = np.broadcast_to(np.asarray(cfg['kappaeff'], dtype=np.float64), (n, nz))
    # Column(kappa=kappaeff, bbot=b[0]) (:159-171); coefficient set 0 = kappa, 1 = kappaeff
    self.cols = ColumnBatch(
        z, np.concatenate([kap, kap]),
        np.concatenate([_rows(cfg['A_basin'], n, nz), _rows(cfg['A_north'], n, nz)]),
        np.concatenate([bb0, bn0]),
        bs=np.concatenate([_vec(cfg['bs'], n), _vec(cfg['bs_north'], n)]),
        bbot=np.concatenate([bb0[:, 0], bn0[:, 0]]), do_conv=True,
        kappa_alt=np.concatenate([kapeff, kapeff]), stream=stream)
    self.cols.set_ksel(np.ones(2 * n, dtype=np.int32))
    self.tw = ThermwindBatch(z, n, f=cfg['f'], nb=self.nb, stream=stream, z_dev=self.cols.z)
    self.so = PsiSOBatch(z, y, n, tau=cfg['tau'], KGM=cfg['KGM'], f=cfg['f'], L=cfg['L'],
                         stream=stream, z_dev=self.cols.z)
    bs0 = cfg['bs_SO0']
    self.ml = SOMLBatch(y, nz, _rows(bs0, n, ny) if np.ndim(bs0) == 1 else bs0,
                        surflux=cfg['surflux'], rest_mask=cfg['rest_mask'],
                        b_rest=cfg['b_rest'], Ks=cfg['Ks'], h=cfg['h'], L=cfg['L'],
                        v_pist=cfg['v_pist'], stream=stream)
    self.wA = DeviceArray.zeros((2 * n, nz), stream=stream)
    self._off = n * nz * 8
    self._bc = pm_jn2018_bc()
    d = self._bc
    d.n, d.nz, d.ny, d.reserved = n, nz, ny, 0
    d.Psi_SO, d.Psi_res_b, d.Psi_res_n = self.so.Psi.ptr, self.tw.psibz1.ptr, self.tw.psibz2.ptr
    d.b_basin, d.b_north = self.cols.b.ptr, self.cols.b.ptr + self._off
    d.bs_SO, d.bbot, d.ksel = self.ml.bs.ptr, self.cols.bbot.ptr, self.cols.ksel.ptr
    self._C = C
    self.ii = 0
    self._graph = None
    self._use_graph = use_graph
    # fused: one launch per MOC block for the whole [BC switch, 2 columns, mixed layer] loop
    self._fused = (nz <= 256) if fused is None else bool(fused)
    self.recorder = None  # optional diagnostics.JN2018Diagnostics
    self.timer = None     # optional device.LaunchTimer
    can_fuse = (self._fused and self.cols.uniform_area and ny <= 64 and
                _run_fits(1, nz, self.nb, ny))
    if fused_run and not can_fuse:
      raise ValueError("fused_run needs the fused step loop (Area constant in z, ny <= 64, "
                       "4 <= nz <= 256) and the phases' LDS within 160 KB")
    self._fused_run = False if fused_run is None else bool(fused_run)
    self._updated_at = -1  # iteration whose MOC update has been done already (fused_run)
    # the two diagnostic launches of an update as one (pm_so_tw_update): scalar tau only? no --
    # any Psi_SO without the boundary-value smoother on nz <= 256
    self._one_update_launch = bool(cfg.get('one_update_launch', True)) and nz <= 256
    self.diag_iters = (cfg.get('Diag_iters', 10 * self.M) if diag_iters is None
                       else diag_iters)
    self.diag = None
    if comm is not None or keep_history:
      self.diag = DiagnosticGather(comm, n, n if n_total is None else n_total,
                                   [(k, nz) for k in ('b_basin', 'b_north', 'Psi', 'Psi_SO')],
                                   stream=stream, keep_history=keep_history, mode=gather,
                                   overlap=gather_overlap)

  def gather_diagnostics(self, step=None):
    self.diag.gather(dict(b_basin=self.cols.b.ptr, b_north=self.cols.b.ptr + self._off,
                          Psi=self.tw.Psi, Psi_SO=self.so.Psi),
                     step=self.ii if step is None else step)

  def _after_update(self):
    if self.recorder is not None:
      self.recorder.maybe_record(self.ii)
    if self.diag is not None and self.diag.due(self.ii, self.diag_iters):
      self.gather_diagnostics(self.ii)

  def _update(self):
    b_basin, b_north = self.cols.b.ptr, self.cols.b.ptr + self._off
    # psib / bgrid go to HBM only in the updates a diagnostics recorder samples right after (every
    # Diag_iters steps, run_JansenNadeau_2018.py:218-226); all the other updates sum only the
    # classes Psibz reads (thermwind.hip.h)
    rec = self.recorder
    store = rec is not None and self.ii % rec.Diag_iters == 0
    if self._one_update_launch:
      # PsiSO.solve + AMOC.solve / Psibz of a member by one wave, ONE launch (pm_so_tw_update)
      ds = self.so.descriptor(b_basin, self.ml.bs)
      dw = self.tw.descriptor(b_basin, b_north, Psi_SO=self.so.Psi, wA1=self.wA.ptr,
                              wA2=self.wA.ptr + self._off, store_psib=store)
      with launch_span(self.timer, "k_so_tw_update", self.stream):
        _lib.check(_lib.lib.pm_so_tw_update(self._C.byref(ds), self._C.byref(dw), _TW_ALL,
                                            _sh(self.stream)))
      return
    with launch_span(self.timer, "k_psi_so", self.stream):
      self.so.update(b_basin, self.ml.bs)
    with launch_span(self.timer, "k_thermwind", self.stream):
      self.tw.update(b_basin, b_north, ops=_TW_ALL, store_psib=store,
                     Psi_SO=self.so.Psi, wA1=self.wA.ptr,
                     wA2=self.wA.ptr + self._off)

  def _step(self):
    from ._lib import check, lib
    from .device import _sh
    check(lib.pm_jn2018_bc_switch(self._C.byref(self._bc), _sh(self.stream)))
    self.cols.steps(self.wA, self.dt, 1, lanes_per_col=self.lanes)
    self.ml.step(self.cols.b.ptr, self.so.Psi, self.dt)

  def _block(self):
    self._update()
    for _ in range(self.M):
      self._step()

  def _fused_steps(self, nsteps):
    from ._lib import check, lib
    from .device import _sh
    d = self._jn_descriptor()
    with launch_span(self.timer, "k_jn2018_steps", self.stream):
      check(lib.pm_jn2018_steps(self._C.byref(d), self.dt, int(nsteps), _sh(self.stream)))

  def _div3(self):
    """PM_JN_DIV3_PROVEN: the columns' static denominators (ColumnBatch.div3_proven) and the
    mixed layer's h, L and y[1] - y[0] admit the 3-instruction exact quotient."""
    if not hasattr(self, "_div3_ok"):
      from .columns import div3_proven, device_reciprocals_exact, _HINT_DIV3_OFF
      t = self.ml
      den = np.array([t.h, t.L, t.y_host[1] - t.y_host[0]], dtype=np.float64)
      a = np.abs(den)
      self._div3_ok = bool(self.cols.uniform_area and self.cols.div3_proven and
                           not (self.cols.__dict__.get("_hints_off", 0) & _HINT_DIV3_OFF) and
                           ((a >= 2.0**-200) & (a <= 2.0**200)).all() and
                           div3_proven(den) and device_reciprocals_exact(den))
    return self._div3_ok

  def _jn_descriptor(self):
    from ._lib import pm_jn2018, pm_so_ml
    d = pm_jn2018()
    d.n = self.n
    d.hints = _lib.PM_JN_UNIFORM_AREA if self.cols.uniform_area else 0
    if self.arith == "contracted":
      d.hints |= _lib.PM_JN_CONTRACTED
    if self.shared_coef and self.cols.uniform_area and self.cols.shared_halves:
      d.hints |= _lib.PM_JN_SHARED_COEF
    if self.split_lanes:
      d.hints |= _lib.PM_JN_SPLIT_LANES
    if self._div3():
      d.hints |= _lib.PM_JN_DIV3_PROVEN
    d.cols = self.cols.descriptor()
    d.wA, d.Psi_SO = self.wA.ptr, self.so.Psi.ptr
    d.Psi_res_b, d.Psi_res_n = self.tw.psibz1.ptr, self.tw.psibz2.ptr
    ml, t = pm_so_ml(), self.ml
    ml.n, ml.nz, ml.ny, ml.reserved = t.n, t.nz, t.ny, 0
    ml.y, ml.bs, ml.Psi_s = t.y.ptr, t.bs.ptr, t.Psi_s.ptr
    ml.b_basin, ml.Psi_b = None, None
    ml.surflux, ml.rest_mask, ml.b_rest = t.surflux.ptr, t.rest_mask.ptr, t.b_rest.ptr
    ml.Ks, ml.h, ml.L, ml.v_pist = t.Ks, t.h, t.L, t.v_pist
    ml.status = t.status.ptr
    d.ml = ml
    return d

  def _stops_after_update(self, ii):
    return ((self.recorder is not None and ii % self.recorder.Diag_iters == 0) or
            (self.diag is not None and self.diag.due(ii, self.diag_iters)))

  def _run_fused(self, nsteps):
    """The loop through pm_jn2018_run: one launch per stretch that ends where the script samples
    its diagnostics (right after a MOC update) or at the end of the run."""
    from ._lib import check, lib
    from .device import _sh
    M, remaining = self.M, int(nsteps)
    b_basin, b_north = self.cols.b.ptr, self.cols.b.ptr + self._off
    while remaining > 0:
      ii = self.ii
      sch = _lib.pm_run_schedule()
      sch.m_steps = M
      fresh = self._updated_at == ii  # this iteration's update is behind us
      sch.n_first = 0 if (ii % M == 0 and not fresh) else min(M - ii % M, remaining)
      pos, rem = ii + sch.n_first, remaining - sch.n_first
      blocks, stopped = [], False
      while rem > 0:
        if self._stops_after_update(pos):
          blocks.append(0)
          stopped = True
          break
        ns = min(M, rem)
        blocks.append(ns)
        pos += ns
        rem -= ns
      sch.n_updates, sch.n_last = len(blocks), (blocks[-1] if blocks else 0)
      d = _lib.pm_jn2018_loop()
      d.jn = self._jn_descriptor()
      d.so = self.so.descriptor(b_basin, self.ml.bs)
      d.tw = self.tw.descriptor(b_basin, b_north, Psi_SO=self.so.Psi, wA1=self.wA.ptr,
                                wA2=self.wA.ptr + self._off,
                                store_psib=self.recorder is not None)
      d.dt, d.sched = self.dt, sch
      check(lib.pm_memset(self.ml.status.ptr, 0, self.ml.status.nbytes, _sh(self.stream)))
      with launch_span(self.timer, "k_jn2018_run", self.stream):
        check(lib.pm_jn2018_run(self._C.byref(d), _sh(self.stream)))
      remaining -= pos - ii
      self.ii = pos
      if stopped:
        self._updated_at = pos
        self._after_update()

  def run(self, nsteps):
    from .device import Graph
    if self._fused_run:
      return self._run_fused(nsteps)
    remaining = int(nsteps)
    while remaining > 0 and self._fused:
      if self.ii % self.M == 0 and self._updated_at != self.ii:
        self._update()
        self._after_update()
      n = min(self.M - self.ii % self.M, remaining)
      self._fused_steps(n)
      self.ii += n
      remaining -= n
    while remaining > 0:
      if (self._use_graph and self.recorder is None and self.diag is None and
          self.ii % self.M == 0 and remaining >= self.M):
        if self._graph is None:
          with Graph.capture(self.stream) as cap:
            self._block()
          self._graph = cap.graph
        self._graph.launch(self.stream)
        self.ii += self.M
        remaining -= self.M
        continue
      if self.ii % self.M == 0 and self._updated_at != self.ii:
        self._update()
        self._after_update()
      self._step()
      self.ii += 1
      remaining -= 1

  def state(self):
    b = self.cols.get_b()
    return dict(b_basin=b[:self.n], b_north=b[self.n:], bs_SO=self.ml.bs.download(stream=self.stream),
                Psi=self.tw.Psi.download(stream=self.stream), Psi_SO=self.so.Psi.download(stream=self.stream),
                Psi_iso_b=self.tw.psibz1.download(stream=self.stream), Psi_iso_n=self.tw.psibz2.download(stream=self.stream),
                Psi_s=self.ml.Psi_s.download(stream=self.stream))

  def nonfinite_members(self):
    nf = self.cols.get_nonfinite()
    return np.nonzero(nf[:self.n] | nf[self.n:])[0]


class TwoBasinEnsemble(object):
  """twobasin_NadeauJansen.py: Atlantic, northern-sinking and Pacific columns; AMOC
  (Atl vs north) and zonal (Atl vs Pac) thermal-wind overturnings mapped to isopycnal space;
  one Southern-Ocean overturning per basin sector.  Columns are stored Atl rows [0,n),
  north rows [n,2n), Pac rows [2n,3n).

  An update (:111-122) is four independent solves of the columns' current profiles:
  {SO_Atl.solve, AMOC.solve / Psibz} and {SO_Pac.solve, ZOC.solve / Psibz}.  Each pair is ONE
  launch (pm_so_tw_update: Psi_SO.solve and the thermal wind of a member by one wavefront) and
  the two pairs run one after the other, or SIDE BY SIDE on two streams (`overlap_updates=True`),
  joined by an event; the forcing of the columns (:103-105) is formed by the column kernel itself
  from the overturnings (PM_OP_WA_TWOBASIN) -- the same device functions and operations as four
  separate launches and a forcing kernel, bit-identical results.
  `comm`, `n_total`, `diag_iters`, `keep_history`, `gather`, `gather_overlap`: as for
  TwoColEnsemble; the exchanged fields are what the script samples every `plot_iters` steps
  (:124-133): the three columns' b and the four overturnings.  `arith="contracted"`: the columns
  step in the opt-in tolerance mode."""

  FIELDS = ("b_Atl", "b_north", "b_Pac", "Psi_AMOC", "Psi_ZOC", "Psi_SO_Atl", "Psi_SO_Pac")

  def __init__(self, cfg, stream=None, lanes_per_col=0, comm=None, n_total=None,
               diag_iters=None, keep_history=False, arith="exact", overlap_updates=False,
               gather="all", gather_overlap=True, use_graph=True):
    if arith not in ("exact", "contracted"):
      raise ValueError("arith must be 'exact' or 'contracted'")
    # Measured at 2048 members (profiles/r05/probe_c6_modes.py, us per interval of 24 steps):
    #   the two update pairs one after the other on ONE stream, the forcing formed by the column
    #   kernel (PM_OP_WA_TWOBASIN): 58.2 -- the default;  ... with pm_twobasin_forcing: 63.8;
    #   the pairs side by side on two streams (`overlap_updates=True`; fork / join events):
    #   66-78, 64-70 replayed from a hipGraph (`use_graph`: a whole interval captured once; only
    #   with overlap_updates, off while a LaunchTimer is attached).  With the round-5 thermal wind
    #   a pair is 17 us: the events cost more than the overlap gives.  (Both pairs as the halves
    #   of ONE launch: 33 us against 2 x 17 -- not kept.)
    self._use_graph, self._graph = bool(use_graph), None
    z, y = cfg['z'], cfg['y']
    nz, ny = z.size, y.size
    n = np.size(cfg['tau']) if np.ndim(cfg['tau']) else 1
    self.n, self.nz, self.ny = n, nz, ny
    self.dt, self.M, self.nb = float(cfg['dt']), int(cfg['MOC_up_iters']), int(cfg['nb'])
    self.lanes, self.stream, self.arith = lanes_per_col, stream, arith
    self.timer = None  # optional device.LaunchTimer
    kap = _rows(cfg['kappa'], n, nz)
    rows = lambda v: _rows(v, n, nz)  # noqa: E731
    self.cols = ColumnBatch(
        z, np.concatenate([kap, kap, kap]),
        np.concatenate([rows(cfg['A_Atl']), rows(cfg['A_north']), rows(cfg['A_Pac'])]),
        np.concatenate([rows(cfg['b_Atl0']), rows(cfg['b_north0']), rows(cfg['b_Pac0'])]),
        bs=np.concatenate([_vec(cfg['bs'], n), _vec(cfg['bs_north'], n), _vec(cfg['bs'], n)]),
        bbot=np.full(3 * n, float(cfg['bbot'])), N2min=float(cfg['N2min']),
        do_conv=np.concatenate([np.zeros(n, bool), np.ones(n, bool), np.zeros(n, bool)]),
        stream=stream)
    zd = self.cols.z
    self.amoc = ThermwindBatch(z, n, f=cfg['f_AMOC'], nb=self.nb, stream=stream, z_dev=zd)
    self.zoc = ThermwindBatch(z, n, f=cfg['f_ZOC'], nb=self.nb, stream=stream, z_dev=zd)
    so = dict(tau=cfg['tau'], KGM=cfg['K'], f=cfg['f_SO'], stream=stream, z_dev=zd)
    self.so_atl = PsiSOBatch(z, y, n, L=cfg['L_Atl'], **so)
    self.so_pac = PsiSOBatch(z, y, n, L=cfg['L_Pac'], **so)
    # the two sectors' Psi_SO in ONE array (rows [0, n) Atlantic, [n, 2n) Pacific): with the two
    # thermal winds' psibz arrays that is what the column kernel forms its forcing from
    # (PM_OP_WA_TWOBASIN) -- no forcing launch between an update and the steps that follow it
    self._so_psi = DeviceArray.zeros((2 * n, nz), stream=stream)
    self.so_atl.Psi, self.so_pac.Psi = self._so_psi.view(0, n), self._so_psi.view(n, n)
    self._forcing_in_k1 = True
    self._wA_fresh = False  # does self.wA hold the forcing of the last update?
    self.bs_SO = DeviceArray.from_host(_rows(cfg['bs_SO'], n, ny) if np.ndim(cfg['bs_SO']) == 1
                                       else cfg['bs_SO'], stream=stream)
    self.wA = DeviceArray.zeros((3 * n, nz), stream=stream)
    self._off = n * nz * 8
    self.ii = 0
    self.diag_iters = (cfg.get('plot_iters', cfg.get('Diag_iters', 10 * self.M))
                       if diag_iters is None else diag_iters)
    self.diag = None
    if comm is not None or keep_history:
      self.diag = DiagnosticGather(comm, n, n if n_total is None else n_total,
                                   [(k, nz) for k in self.FIELDS], stream=stream,
                                   keep_history=keep_history, mode=gather,
                                   overlap=gather_overlap)
    self._pairs = nz <= 256   # pm_so_tw_update covers the shape
    self._overlap = bool(overlap_updates)
    if self._overlap:
      from .device import Event, Stream
      self._side = Stream()
      self._ev_fork, self._ev_join = Event(), Event()
    # initial diagnostics (:58-79): AMOC against b2 = 0.01*b_Atl, the rest on the initial columns
    b2 = DeviceArray.from_host(rows(cfg['b2_init']), stream=stream)
    self._update(b_north=b2.ptr)
    if stream is not None:
      stream.sync()  # b2 is released on return
    else:
      _lib.check(_lib.lib.pm_stream_sync(None))

  def _solve_pair(self, so, tw, b_so, b1, b2, stream, names):
    """{Psi_SO.solve on b_so, thermal wind of (b1, b2)} on `stream`: one launch when the shape
    allows, else two."""
    import ctypes as C
    if self._pairs:
      ds = so.descriptor(b_so, self.bs_SO)
      dw = tw.descriptor(b1, b2, store_psib=False)
      with launch_span(self.timer, "k_so_tw_update", stream):
        _lib.check(_lib.lib.pm_so_tw_update(C.byref(ds), C.byref(dw), _TW_ALL, _sh(stream)))
      return
    keep_so, keep_tw = so.stream, tw.stream
    so.stream = tw.stream = stream
    try:
      with launch_span(self.timer, "k_psi_so", stream):
        so.update(b_so, self.bs_SO)
      with launch_span(self.timer, "k_thermwind", stream):
        tw.update(b1, b2, ops=_TW_ALL, store_psib=False)
    finally:
      so.stream, tw.stream = keep_so, keep_tw

  def _update(self, b_north=None):
    from ._lib import check, lib
    bA = self.cols.b.ptr
    bN = self.cols.b.ptr + self._off if b_north is None else b_north
    bP = self.cols.b.ptr + 2 * self._off
    if self._overlap:
      self._ev_fork.record(self.stream)     # the columns' steps before this update
      self._side.wait(self._ev_fork)
      self._solve_pair(self.so_pac, self.zoc, bP, bA, bP, self._side, "pac")
      self._ev_join.record(self._side)
      self._solve_pair(self.so_atl, self.amoc, bA, bA, bN, self.stream, "atl")
      check(lib.pm_stream_wait_event(_sh(self.stream), self._ev_join.handle))
    else:
      self._solve_pair(self.so_atl, self.amoc, bA, bA, bN, self.stream, "atl")
      self._solve_pair(self.so_pac, self.zoc, bP, bA, bP, self.stream, "pac")
    self._wA_fresh = False
    if not self._forcing_in_k1:
      self._form_forcing()

  def _form_forcing(self):
    """wA of the three columns as an array (:103-105; launches of 1-2 steps, and
    `_forcing_in_k1 = False`): pm_twobasin_forcing."""
    from ._lib import check, lib
    w = self.wA.ptr
    check(lib.pm_twobasin_forcing(self.n, self.nz, self.amoc.psibz1.ptr, self.zoc.psibz1.ptr,
                                  self.so_atl.Psi.ptr, self.amoc.psibz2.ptr,
                                  self.zoc.psibz2.ptr, self.so_pac.Psi.ptr, w, w + self._off,
                                  w + 2 * self._off, _sh(self.stream)))
    self._wA_fresh = True

  def _steps(self, n):
    """n column steps under the forcing of the last update: formed by the column kernel itself
    (>= 3 steps per launch) or taken from the array."""
    if self._forcing_in_k1 and n >= 3:
      self.cols.steps(None, self.dt, n, lanes_per_col=self.lanes, arith=self.arith,
                      twobasin_forcing=(self.amoc.psibz, self.zoc.psibz, self._so_psi))
      return
    if not self._wA_fresh:
      self._form_forcing()
    self.cols.steps(self.wA, self.dt, n, lanes_per_col=self.lanes, arith=self.arith)

  def run(self, nsteps):
    remaining = int(nsteps)
    while remaining > 0:
      nxt = self.ii if self.ii % self.M == 0 else (self.ii // self.M + 1) * self.M
      n = min(nxt - self.ii + 1, remaining)
      if (self._use_graph and self.timer is None and self._overlap and n == self.M and
          self.ii % self.M == 1 and self.M >= 3):
        # a full interval: M steps, then the update they end on
        from .device import Graph
        if self._graph is None:
          with Graph.capture(self.stream) as cap:
            self._steps(n)
            self._update()
          self._graph = cap.graph
        self._graph.launch(self.stream)
        self.ii += n
        remaining -= n
        if self.diag is not None and self.diag.due(self.ii - 1, self.diag_iters):
          self.gather_diagnostics(self.ii - 1)
        continue
      with launch_span(self.timer, "k_column_steps" if n >= 3 else "k_column_steps_short",
                       self.stream):
        self._steps(n)
      self.ii += n
      remaining -= n
      if (self.ii - 1) % self.M == 0:
        self._update()
        if self.diag is not None and self.diag.due(self.ii - 1, self.diag_iters):
          self.gather_diagnostics(self.ii - 1)

  def gather_diagnostics(self, step=None):
    """Gather the seven fields the script samples every plot_iters steps (:124-133)."""
    b = self.cols.b.ptr
    self.diag.gather(dict(b_Atl=b, b_north=b + self._off, b_Pac=b + 2 * self._off,
                          Psi_AMOC=self.amoc.Psi, Psi_ZOC=self.zoc.Psi,
                          Psi_SO_Atl=self.so_atl.Psi, Psi_SO_Pac=self.so_pac.Psi),
                     step=self.ii if step is None else step)

  def state(self):
    b, n = self.cols.get_b(), self.n
    return dict(b_Atl=b[:n], b_north=b[n:2 * n], b_Pac=b[2 * n:], Psi_AMOC=self.amoc.Psi.download(stream=self.stream),
                Psi_ZOC=self.zoc.Psi.download(stream=self.stream), Psi_SO_Atl=self.so_atl.Psi.download(stream=self.stream),
                Psi_SO_Pac=self.so_pac.Psi.download(stream=self.stream))

  def nonfinite_members(self):
    nf, n = self.cols.get_nonfinite(), self.n
    return np.nonzero(nf[:n] | nf[n:2 * n] | nf[2 * n:])[0]
