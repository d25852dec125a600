"""The drop-in module classes: the reference's own unit tests
(tests/modules/test_column.py:247-336) re-expressed against pymoc_amd, and user-style
loops written exactly like the reference's example scripts."""
import numpy as np
import pytest

import oracle as O
from oracle import drivers
from conftest import load_golden, relerr
from pymoc_amd import configs

pytestmark = pytest.mark.gpu


def test_vertadvdiff_reference_unit_test(gpu):
  dt = 60 * 86400
  Area = 6e13
  z = np.asarray(np.linspace(-4000, 0, 80))
  kappa = 2e-5
  b = np.linspace(0.03, -0.002, 80)
  wA = Area * np.sin(z)
  db_dt1 = (0.0004 / 50.0 / Area) * (-wA)
  column = gpu.Column(z=z, Area=Area, kappa=kappa, b=b.copy(), bbot=-0.002, bs=0.03)
  column.vertadvdiff(wA, dt, do_conv=False)
  assert all(np.around(column.b[2:-2], 3) == np.around(b[2:-2] - dt * db_dt1[2:-2], 3))
  assert np.array_equal(column.b, O.column_vertadvdiff(z, kappa + 0 * z, Area + 0 * z, b, wA, dt,
                                                       bs=0.03, bbot=-0.002))


def test_convect_reference_unit_test(gpu):
  N2min = 1.5e-7
  z = np.asarray([-4000.0, -1000.0, -100.0, 0.0])
  b = np.asarray([-0.03, -0.02, 0.01, 0.01])
  column = gpu.Column(z=z, b=b.copy(), bs=0.0, N2min=N2min, kappa=2e-5, Area=6e13)
  b[2:] = 0.0 + N2min * (z[2:] - z[1])
  column.convect()
  assert all(column.b == b)


def test_horadv_reference_unit_test(gpu):
  z = np.asarray([-4000.0, -1000.0, -100.0, 0.0])
  b = np.asarray([-0.03, 0.01, -0.0025, -0.002])
  column = gpu.Column(z=z, b=b.copy(), kappa=2e-5, Area=6e13)
  vdx_in = np.asarray([2e8, 2.5e8, 0.0, 0.0])
  b_in = np.asarray([-0.02, 0.01, -0.001, 0.001])
  dt = 60 * 86400
  b[0] = -0.03 + dt * 2e6 / 6e13
  column.horadv(vdx_in, b_in, dt)
  assert all(np.around(column.b, 4) == np.around(b, 4))


def test_timestep_composition_reference_unit_test(gpu):
  Area = 6e13
  z = np.asarray(np.linspace(-4000, 0, 80))
  b = np.linspace(-np.sqrt(0.04), 0.0, 80)**2.
  vdx_in = np.asarray([2e4 for n in z])
  b_in = np.asarray([-0.02 for n in z])
  wA = np.sin(z) / Area
  dt = 30 * 86400
  mk = lambda: gpu.Column(z=z, b=b.copy(), bs=-0.0, bbot=-0.04, kappa=2e-5, Area=Area)  # noqa
  c1, c2 = mk(), mk()
  c1.timestep(wA=wA, dt=dt)
  c2.vertadvdiff(wA=wA, dt=dt)
  assert all(c1.b == c2.b)
  c2.horadv(vdx_in=vdx_in, b_in=b_in, dt=dt)
  c2.convect()
  assert any(c1.b != c2.b)
  c1, c2 = mk(), mk()
  c1.timestep(wA=wA, dt=dt, b_in=b_in, vdx_in=vdx_in)
  c2.vertadvdiff(wA=wA, dt=dt)
  c2.horadv(vdx_in=vdx_in, b_in=b_in, dt=dt)
  assert all(c1.b == c2.b)
  c2.convect()
  assert any(c1.b != c2.b)
  c1, c2 = mk(), mk()
  c1.timestep(wA=wA, dt=dt, b_in=b_in, vdx_in=vdx_in, do_conv=True)
  c2.convect()
  c2.vertadvdiff(wA=wA, dt=dt)
  c2.horadv(vdx_in=vdx_in, b_in=b_in, dt=dt)
  assert all(c1.b == c2.b)
  c = mk()
  with pytest.raises(TypeError) as e:
    c.timestep(wA=wA, dt=dt, vdx_in=vdx_in)
  assert str(e.value) == "b_in is needed if vdx_in is provided"
  with pytest.raises(TypeError) as e:
    c.timestep(wA=1, dt=dt)
  assert str(e.value) == "('wA', 'needs to be either function, numpy array, or float')"


def test_column_updates_users_array_in_place_and_tracks_attribute_pokes(gpu):
  z = np.linspace(-4000., 0., 50)
  b = 0.03 * np.exp(z / 300.)
  col = gpu.Column(z=z, kappa=lambda zz: 1e-5 + 0 * zz, Area=8e13, b=b, bs=0.03, bbot=0.0)
  assert col.b is b
  col.timestep(wA=0., dt=86400.)
  assert col.b is b and b[0] == 0.0          # updated in place (column.py:249)
  col.bbot = -0.001                          # user pokes between steps
  col.kappa = lambda zz: 5e-5 + 0 * zz       # (run_JansenNadeau_2018.py:233-254)
  ref = O.column_timestep(z, 5e-5 + 0 * z, 8e13 + 0 * z, b, 0 * z, 86400., bs=0.03, bbot=-0.001)
  col.timestep(wA=0., dt=86400.)
  assert np.array_equal(col.b, ref) and col.b is b  # reading col.b runs the queued step
  assert np.array_equal(b, ref)                     # ... into the caller's array


def test_column_lazy_queue_is_bitwise_the_eager_sequence(gpu):
  """Lazy stepping: identical consecutive timestep() calls are queued and
  run as ONE fused launch when the state is needed; every way of needing it must give exactly
  the step-by-step result -- reading .b, a changed wA / dt / do_conv, attribute pokes
  (run_JansenNadeau_2018.py:233-254 style), an in-place edit of an aliased kappa array,
  another module object reading through an alias, horadv inputs, and the eager mode."""
  from pymoc_amd.modules import column as colmod
  z = np.linspace(-4000., 0., 80)
  kap0 = 2e-5 + 2e-4 * np.exp(-z / 1000 - 4)
  kapeff = lambda zz: 6e-5 + 0 * zz  # noqa: E731
  b0 = 0.03 * np.exp(z / 300.) - 0.001
  wA1 = 8e13 * 2e-8 * np.sin(np.pi * z / 4000.)
  wA2 = -0.5 * wA1

  def script(col, tw):
    """A user loop with every kind of event; returns what the user would have seen."""
    seen = []
    for ii in range(40):
      wA = wA1 if (ii // 7) % 2 == 0 else wA2   # forcing changes every 7 steps
      if ii == 11:
        col.bbot = -0.002                      # attribute poke
      if ii == 17:
        col.kappa = kapeff                     # coefficient set switch
      if ii == 23:
        col.kappa = colmod.make_func(kap0, z, 'kappa')
      if ii == 29:
        kap0[:10] *= 1.5                       # in-place edit of the aliased kappa array
      col.timestep(wA=wA, dt=86400. * 30, do_conv=(ii >= 20))
      if ii == 5:
        seen.append(col.b.copy())              # plain read
      if ii == 14:
        tw.update(b1=col.b)                    # alias handed to another module object ...
      if ii in (15, 16, 33):
        tw.solve()                             # ... which reads it later
        seen.append(tw.Psi.copy())
      if ii == 36:
        col.timestep(wA=wA, dt=86400. * 30, vdx_in=1e3 + 0 * z, b_in=b0)  # horadv: immediate
    seen.append(col.b.copy())
    return seen

  def run(lazy):
    colmod.LAZY = lazy
    kap0[:] = 2e-5 + 2e-4 * np.exp(-z / 1000 - 4)
    col = gpu.Column(z=z, kappa=kap0, Area=8e13, b=b0.copy(), bs=0.03, bbot=-0.001)
    tw = gpu.Psi_Thermwind(z=z, b1=b0.copy(), b2=0.)
    try:
      return script(col, tw)
    finally:
      colmod.LAZY = True

  eager, lazy = run(False), run(True)
  assert len(eager) == len(lazy) == 5
  for a, b in zip(eager, lazy):
    assert np.array_equal(a, b)
  # and the queue really fuses: 24 identical steps = 1 launch's worth of work, flushed by .b
  col = gpu.Column(z=z, kappa=2e-5, Area=8e13, b=b0.copy(), bs=0.03, bbot=-0.001)
  for _ in range(24):
    col.timestep(wA=wA1, dt=86400. * 30)
  assert col._q is not None and col._q[0] == 24
  ref = b0.copy()
  for _ in range(24):
    ref = O.column_timestep(z, 2e-5 + 0 * z, 8e13 + 0 * z, ref, wA1, 86400. * 30, bs=0.03,
                            bbot=-0.001)
  assert np.array_equal(col.b, ref) and col._q is None


def test_column_callable_depending_on_mutable_state_vs_oracle(gpu):
  """A user callable whose OUTPUT changes while its identity does not (a closure over a scale
  factor the loop updates): the reference evaluates kappa(z) and Area(z) in every step
  (column.py:122, :241-248), so every step must see the current values -- lazy and eager."""
  from pymoc_amd.modules import column as colmod
  z = np.linspace(-4000., 0., 80)
  b0 = 0.03 * np.exp(z / 300.) - 0.001
  wA = 8e13 * 2e-8 * np.sin(np.pi * z / 4000.)
  state = {"scale": 1.0, "area": 8e13}
  kappa = lambda zz: state["scale"] * (2e-5 + 2e-4 * np.exp(-zz / 1000 - 4))  # noqa: E731
  area = lambda zz: state["area"] + 0 * zz  # noqa: E731
  for lazy in (True, False):
    colmod.LAZY = lazy
    try:
      state.update(scale=1.0, area=8e13)
      col = gpu.Column(z=z, kappa=kappa, Area=area, b=b0.copy(), bs=0.03, bbot=-0.001)
      ref = b0.copy()
      for ii in range(30):
        if ii == 10:
          state["scale"] = 2.5   # same function object, different values
        if ii == 20:
          state["area"] = 5e13
        col.timestep(wA=wA, dt=86400. * 30)
        ref = O.column_timestep(z, kappa(z), area(z), ref, wA, 86400. * 30, bs=0.03, bbot=-0.001)
      assert np.array_equal(col.b, ref), lazy
    finally:
      colmod.LAZY = True


def test_column_alias_write_between_queued_steps(gpu):
  """`arr = col.b` once, then `arr[k] = ...` between timestep() calls: the write is applied
  after the steps queued before it, as in the reference (where b is updated in place)."""
  z = np.linspace(-4000., 0., 80)
  b0 = 0.03 * np.exp(z / 300.) - 0.001
  wA = 8e13 * 2e-8 * np.sin(np.pi * z / 4000.)
  col = gpu.Column(z=z, kappa=2e-5, Area=8e13, b=b0.copy(), bs=0.03, bbot=-0.001)
  arr = col.b
  ref = b0.copy()
  for ii in range(20):
    if ii in (7, 13):
      arr[40] = 0.011 + 1e-4 * ii   # through the alias, not through col.b
      ref[40] = 0.011 + 1e-4 * ii
    col.timestep(wA=wA, dt=86400. * 30)
    ref = O.column_timestep(z, 2e-5 + 0 * z, 8e13 + 0 * z, ref, wA, 86400. * 30, bs=0.03,
                            bbot=-0.001)
  assert np.array_equal(col.b, ref) and col.b is arr


def test_user_loop_like_example_twocol(gpu):
  """A loop written exactly like examples/example_twocol.py:85-96, with the drop-in classes."""
  m = configs.twocol_member(nz=80)
  z = m["z"]
  AMOC = gpu.Psi_Thermwind(z=z, b1=m["b_basin0"].copy(), b2=m["b_north0"].copy())
  AMOC.solve()
  [Psi_iso_b, Psi_iso_n] = AMOC.Psibz()
  basin = gpu.Column(z=z, kappa=m["kappa"].copy(), Area=m["A_basin"], b=m["b_basin0"].copy(),
                     bs=m["bs"], bbot=m["bbot"])
  north = gpu.Column(z=z, kappa=m["kappa"].copy(), Area=m["A_north"], b=m["b_north0"].copy(),
                     bs=m["bs_north"], bbot=m["bbot"])
  for ii in range(0, 60):
    wAb = Psi_iso_b * 1e6
    wAN = -Psi_iso_n * 1e6
    basin.timestep(wA=wAb, dt=m["dt"])
    north.timestep(wA=wAN, dt=m["dt"], do_conv=True)
    if ii % m["MOC_up_iters"] == 0:
      AMOC.update(b1=basin.b, b2=north.b)
      AMOC.solve()
      [Psi_iso_b, Psi_iso_n] = AMOC.Psibz()
  ref = drivers.run_twocol(m, 60, {60})[60]
  assert np.array_equal(basin.b, ref["b_basin"])
  assert np.array_equal(north.b, ref["b_north"])
  assert np.array_equal(AMOC.Psi, ref["Psi"])


def test_rccl_communicator_single_rank(gpu):
  """RCCL through the C-ABI on the one GPU of this box: init, barrier, all-gather, max."""
  from pymoc_amd import sharding
  from pymoc_amd.device import DeviceArray
  comm = sharding.RcclCommunicator(rank=0, world=1)
  a = np.arange(600.).reshape(6, 100)
  send, recv = DeviceArray.from_host(a), DeviceArray((1, 6, 100))
  comm.allgather_device(send, recv)
  comm.barrier()
  assert np.array_equal(recv.download()[0], a)
  assert np.array_equal(sharding.gather_members(comm, a, 6), a)
  assert comm.max_host(3.5) == 3.5
  # gather to the root through ncclSend / ncclRecv: with one rank the root sends to itself
  # (always_collective), the only way the point-to-point path can run on a one-GPU box
  recv2 = DeviceArray((1, 6, 100))
  comm.gather_root_device(send, recv2, root=0)
  comm.barrier()
  assert np.array_equal(recv2.download()[0], a)
  # ... and the coupled drivers' exchange object on top of it, both modes, on its own stream
  for mode in ("all", "root"):
    g = sharding.DiagnosticGather(comm, 6, 6, [("a", 100), ("b", 40)], mode=mode,
                                  keep_history=True)
    bsrc = DeviceArray.from_host(np.arange(240.).reshape(6, 40))
    for k in range(3):  # three gathers back to back: both send buffers get reused
      send.upload(a + k)
      g.gather(dict(a=send, b=bsrc), step=k)
    assert g.ncollectives == 3
    last = g.last()
    assert np.array_equal(last["a"], a + 2) and np.array_equal(last["b"], bsrc.download())
    assert [s for s, _ in g.history] == [0, 1, 2]
    assert all(np.array_equal(d["a"], a + k) for k, (_, d) in enumerate(g.history))
  comm.close()


def test_rows_pack_gathers_selected_rows(gpu):
  """pm_rows_pack: dst[r] = src[sel[r]] for several fields in one launch (the diagnostic
  recorder's append and the exchange's send-buffer pack)."""
  from pymoc_amd.device import DeviceArray, rows_pack
  rng = np.random.default_rng(5)
  a, b = rng.standard_normal((37, 100)), rng.standard_normal((37, 51))
  da, db = DeviceArray.from_host(a), DeviceArray.from_host(b)
  sel = np.array([36, 0, 5, 5, 17], dtype=np.int32)
  dsel = DeviceArray.from_host(sel)
  out = DeviceArray.zeros((5 * 100 + 5 * 51 + 37 * 60,))
  rows_pack([(da.ptr, out.ptr, 100, 100), (db.ptr, out.ptr + 8 * 500, 51, 51)], 5, sel=dsel)
  # no selection, a row prefix of a wider array (src_stride > nlev)
  rows_pack([(da.ptr, out.ptr + 8 * (500 + 255), 60, 100)], 37)
  h = out.download()
  assert np.array_equal(h[:500].reshape(5, 100), a[sel])
  assert np.array_equal(h[500:755].reshape(5, 51), b[sel])
  assert np.array_equal(h[755:].reshape(37, 60), a[:, :60])
  with pytest.raises(gpu._lib.PmError):
    rows_pack([(da.ptr, out.ptr, 100, 50)], 5)  # stride shorter than the row
