"""Device-side self tests of the arithmetic building blocks."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("emax", [40, 200, 400])
def test_division_by_precomputed_reciprocal_is_correctly_rounded(gpu, emax):
  """div_by_recip (common.hip.h) == IEEE `/`, bit for bit, on ~4e9 random operand pairs
  per exponent range (3/8 of the mantissas at the all-ones / all-zeros / half edges)."""
  from pymoc_amd._lib import lib, check
  tested, bad = C.c_uint64(0), C.c_uint64(0)
  check(lib.pm_selftest_fastdiv(20240 + emax, 2048, 2000, emax, C.byref(tested), C.byref(bad)))
  assert tested.value == 4 * 2000 * 256 * 2048
  assert bad.value == 0, "%d of %d quotients differ" % (bad.value, tested.value)


def test_three_instruction_quotient_on_its_candidate_numerators(gpu):
  """PM_COLS_DIV3_PROVEN: for 2^18 random denominators (uniform mantissas, mantissas next to 1 and
  2, trailing zeros) the host proof passes, and the DEVICE's q0 = a y, r = fma(-d, q0, a), q =
  fma(r, y, q0) equals the HOST's IEEE quotient bit for bit on every candidate numerator (the only
  ones whose quotient lies close enough to a rounding boundary to fail), both signs, rescaled, plus
  arbitrary ones.  (The device's own `/` does NOT: ~1e-4 of these near-midpoint quotients are an
  ulp off -- reported, and bounded here so that a change of that rate is noticed.)"""
  from pymoc_amd._lib import lib, check
  tested, bad, unproven, off = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
  pair = (C.c_double * 2)(0., 0.)
  check(lib.pm_selftest_div3(20245, 1 << 18, C.byref(tested), C.byref(bad), C.byref(unproven),
                             C.byref(off), pair))
  print("3-instruction quotient: %d pairs, %d mismatches, %d unproven denominators; the device's own "
        "a / d is off on %d of them" % (tested.value, bad.value, unproven.value, off.value))
  assert tested.value > (1 << 20)
  assert bad.value == 0 and unproven.value == 0, (bad.value, unproven.value, tested.value,
                                                  float(pair[0]).hex(), float(pair[1]).hex())
  assert off.value < tested.value // 1000


def test_device_reciprocals_of_the_bench_grids_are_correctly_rounded(gpu):
  """pm_recip_check: the device's 1.0 / d equals the host's IEEE quotient for BASELINE's grid
  spacings, Areas and for 2^16 random denominators (what ColumnBatch asks before it sets
  PM_COLS_DIV3_PROVEN)."""
  import numpy as np
  from pymoc_amd import configs
  from pymoc_amd._lib import lib, check
  c = configs.config2()
  dz = np.diff(c["z"])
  rnd = np.random.default_rng(3)
  d = np.concatenate([dz, 0.5 * (dz[1:] + dz[:-1]), np.asarray(c["Area"])[:, 0],
                      rnd.uniform(1, 2, 1 << 16) * 2.0**rnd.integers(-60, 60, 1 << 16)])
  d = np.ascontiguousarray(d)
  ok = C.c_int32(0)
  check(lib.pm_recip_check(d.ctypes.data, d.size, C.byref(ok)))
  assert ok.value == 1


@pytest.mark.parametrize("nhas", [1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64])
def test_dpp_wave_scans_match_serial_composition(gpu, nhas):
  """The GM boundary-value solve's wave scans (psi_so.hip.h: rows of 16 by DPP row shifts, the
  rows joined by row_bcast / v_readlane) for every way the occupied lanes can end inside or at
  the edge of a row: the element prefix and suffix scans and the affine suffix scan against the
  same compositions done serially (the association differs, so 1e-12 rather than bits), the
  integer prefix sum exactly."""
  from pymoc_amd._lib import lib, check
  for seed in range(5):
    dev, bad = (C.c_double * 3)(), C.c_int32(0)
    check(lib.pm_selftest_so_scans(nhas, 1000 + seed, dev, C.byref(bad)))
    assert bad.value == 0
    assert max(dev) < 1e-12, (nhas, seed, list(dev))


def test_high_priority_stream_runs_the_same_work(gpu):
  """pm_stream_create_priority: a batch stepped on a stream of the device's highest priority gives
  the bits it gives on an ordinary stream (the entry point only changes dispatch order)."""
  import numpy as np
  from pymoc_amd import configs
  from pymoc_amd.device import Stream
  c = configs.config2(N=96)
  out = []
  for st in (Stream(), Stream(high_priority=True)):
    b = gpu.ColumnBatch(c["z"], c["kappa"], c["Area"], c["b0"], bs=c["bs"], bbot=c["bbot"],
                        N2min=c["N2min"], do_conv=c["do_conv"], stream=st)
    b.steps(gpu.DeviceArray.from_host(c["wA"], stream=st), c["dt"], 50)
    st.sync()
    out.append(b.get_b())
  assert np.array_equal(out[0], out[1])
